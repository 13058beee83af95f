#!/bin/bash
# usage: scratch/variants.sh name "sed-expression"  -> scratch/libs/name.so built from a patched copy of kernels.hip
set -e
cd /root/repo
name=$1; expr=$2
cp nutpie_amd/csrc/kernels.hip /tmp/kernels_backup.hip
sed -i "$expr" nutpie_amd/csrc/kernels.hip
make -j8 -C nutpie_amd/csrc 2>&1 | grep -E "rror" || true
cp nutpie_amd/libnutpie_hip.so scratch/libs/$name.so
cp /tmp/kernels_backup.hip nutpie_amd/csrc/kernels.hip

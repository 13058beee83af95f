#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
NPHIP_DEBUG=1 timeout 900 python scratch/r6_dense_resident.py > gpurun_out/r6_dense_resident.txt 2>&1
echo "rc $?" >> gpurun_out/r6_dense_resident.txt
cat gpurun_out/r6_dense_resident.txt

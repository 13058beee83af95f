#!/bin/bash
# usage: scratch/devlib.sh <name> "<extra hipcc flags>" ["sed expression applied to a copy of kernels.hip"]
#   -> scratch/libs/<name>.so with ONE kernel instantiation (seconds to build):
#   -DNPHIP_DEV_W1NV=8 (the 1000-dim kernel) or -DNPHIP_DEV_LEAN -DNPHIP_DEV_W=4 -DNPHIP_DEV_NC=20; add -DNPHIP_PROFILE for the cycle attribution
set -e
cd /root/repo/nutpie_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $2"
SRC=kernels.hip
if [ -n "$3" ]; then SRC=kernels_variant_$1.hip; sed "$3" kernels.hip > $SRC; cmp -s kernels.hip $SRC && { echo "sed expression changed nothing"; rm -f $SRC; exit 1; }; fi
/opt/rocm/bin/hipcc $F -c $SRC -o /tmp/dev_$1_k.o
[ -n "$3" ] && rm -f $SRC
[ -f /tmp/dev_host.o ] && [ /tmp/dev_host.o -nt host.hip ] && [ /tmp/dev_host.o -nt engine_types.h ] || /opt/rocm/bin/hipcc $F -c host.hip -o /tmp/dev_host.o
mkdir -p ../../scratch/libs
[ -f linalg.o ] || /opt/rocm/bin/hipcc $F -c linalg.hip -o linalg.o
[ -f lowrank_est.o ] || /opt/rocm/bin/hipcc $F -c lowrank_est.hip -o lowrank_est.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libs/$1.so /tmp/dev_$1_k.o /tmp/dev_host.o linalg.o lowrank_est.o -pthread
ls -la ../../scratch/libs/$1.so

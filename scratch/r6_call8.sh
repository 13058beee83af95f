#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for v in 0 128 64; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg2.so timeout 120 python scratch/r6_dg_variants.py 129 check 2>&1 | grep -v amdgpu.ids
done
for v in 32 160; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so timeout 120 python scratch/r6_dg_variants.py 1000 2>&1 | grep -v amdgpu.ids
done
NPHIP_DG_VARIANT=0 NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so timeout 300 python scratch/r6_dg_variants.py 1000 check 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_dg_variants6.txt 2>&1
cat gpurun_out/r6_dg_variants6.txt

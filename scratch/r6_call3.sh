#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for v in 0 1 3 2 4 7; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so timeout 120 python scratch/r6_dg_variants.py 1000 2>&1 | grep -v amdgpu.ids
done
NPHIP_DG_VARIANT=3 NUTPIE_HIP_LIB=$R/scratch/libs/dg2.so timeout 120 python scratch/r6_dg_variants.py 129 check 2>&1 | grep -v amdgpu.ids
for v in 0 3 7; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg2.so timeout 120 python scratch/r6_dg_variants.py 129 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r6_dg_variants.txt 2>&1
cat gpurun_out/r6_dg_variants.txt

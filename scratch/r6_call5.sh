#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
for v in 33 32 41; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so timeout 120 python scratch/r6_dg_variants.py 1000 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r6_dg_variants3.txt 2>&1
cat gpurun_out/r6_dg_variants3.txt

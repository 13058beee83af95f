#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
# seats: 0 dynamic (die from XCC_ID) / 64 static; fence mode << 7: 0 inv sc0, 128 agent fences, 256 inv sc1, 384 none
for v in 128 192 0 64 256 320 384; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg2.so timeout 120 python scratch/r6_dg_variants.py 129 check 2>&1 | grep -v amdgpu.ids
done
for v in 160 288 416; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so timeout 120 python scratch/r6_dg_variants.py 1000 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r6_dg_variants5.txt 2>&1
cat gpurun_out/r6_dg_variants5.txt

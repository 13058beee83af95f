#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
NUTPIE_HIP_LIB=$R/scratch/libs/dg2.so timeout 120 python scratch/r6_dg_variants.py 129 check 2>&1 | grep -v amdgpu.ids
NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so timeout 300 python scratch/r6_dg_variants.py 1000 check 2>&1 | grep -v amdgpu.ids
NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so python scratch/r6_dense_job.py 12 2>&1 | grep -v amdgpu.ids
NPHIP_DG_VARIANT=32 NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so python scratch/r6_dense_job.py 12 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_dense_lp_fused.txt 2>&1
cat gpurun_out/r6_dense_lp_fused.txt

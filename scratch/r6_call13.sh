#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for v in 0 256 0 256; do
NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so python scratch/r6_dense_job.py 12 2>&1 | grep -v amdgpu.ids
done
NPHIP_DG_VARIANT=32 NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so python scratch/r6_dense_job.py 12 2>&1 | grep -v amdgpu.ids
NPHIP_DG_VARIANT=288 NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so python scratch/r6_dense_job.py 12 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r6_dense_prefetch.txt 2>&1
cat gpurun_out/r6_dense_prefetch.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
{
NPHIP_DEBUG=1 python -c "
import sys; sys.path.insert(0, '.')
from nutpie_amd import _lib
for i in range(3): print('mfma f64 rate', _lib.mfma_f64_rate(0))
" 2>&1 | grep -v amdgpu.ids
for v in 1 9 17 25; do
  NPHIP_DG_VARIANT=$v NUTPIE_HIP_LIB=$R/scratch/libs/dg8.so timeout 120 python scratch/r6_dg_variants.py 1000 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r6_dg_variants2.txt 2>&1
cat gpurun_out/r6_dg_variants2.txt

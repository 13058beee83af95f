#!/bin/bash
# Round 5: A/B of the generated density's loop structure on config 3 (scratch/r5_segsum_ab.py once per variant)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for v in "loop 0 4" "select 0 4" "select 1 4" "select 1 8" "loop 0 4" "select 1 8"; do
  set -- $v
  NUTPIE_AMD_SEG_MODE=$1 NUTPIE_AMD_FAST_PATH=$2 NUTPIE_AMD_UNROLL_MAX=$3 python scratch/r5_segsum_ab.py 2>&1 | grep "^\[" | sed "s/^\[[a-z]*\]/[seg=$1 fast=$2 umax=$3]/"
done | tee gpurun_out/r5_segsum_ab.txt
python -m pytest tests/test_gpu_symbolic.py tests/test_gpu_torch_trace.py tests/test_gpu_density.py -x -q 2>&1 | grep -E "passed|failed|error" | tee -a gpurun_out/r5_segsum_ab.txt

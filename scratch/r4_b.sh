#!/bin/bash
# round 4, GPU call B: the fused callback leaf — parity, then timing
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4b
python -m pytest tests/test_gpu_parity.py -k "callback or device" -x -q 2>&1 | tail -8
python -m pytest tests/test_gpu_api.py tests/test_gpu_density.py -x -q 2>&1 | tail -5
for w in 0 4 8; do
  for g in 16; do python scratch/cbtime.py 1000 1024 $w $g 2>&1 | tail -1; done
done > gpurun_out/r4b/cbtime.txt 2>&1
python scratch/cbtime.py 173 512 0 16 2>&1 | tail -1 >> gpurun_out/r4b/cbtime.txt
python scratch/cbtime.py 173 512 2 16 2>&1 | tail -1 >> gpurun_out/r4b/cbtime.txt
python scratch/cbtime.py 200 1024 1 16 2>&1 | tail -1 >> gpurun_out/r4b/cbtime.txt
cat gpurun_out/r4b/cbtime.txt

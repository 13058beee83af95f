#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4d
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4d/gputests.txt; cat gpurun_out/r4d/gputests.txt
for a in "1000 1024 0 16" "1000 1024 0 0" "173 512 0 16" "200 1024 0 16" "2000 512 0 16" "4000 256 0 16"; do python scratch/cbtime.py $a 2>&1 | tail -1; done | tee gpurun_out/r4d/cbtime.txt
python scratch/dense.py 2>&1 | tail -2 | tee gpurun_out/r4d/dense.txt

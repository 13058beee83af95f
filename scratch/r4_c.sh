#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4c
python -m pytest tests/test_gpu_parity.py -k "callback or device" -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_api.py tests/test_gpu_density.py tests/test_gpu_low_rank.py -x -q 2>&1 | tail -4
NUTPIE_HIP_LIB=$R/scratch/libs/cb4prof.so python scratch/cbtime.py 1000 1024 4 16 400 100 2>&1 | tail -3
NUTPIE_HIP_LIB=$R/scratch/libs/cb4.so python scratch/cbtime.py 1000 1024 4 16 400 100 2>&1 | tail -1
python scratch/cbtime.py 1000 1024 4 16 2>&1 | tail -1
python scratch/cbtime.py 173 512 0 16 2>&1 | tail -1
python scratch/cbtime.py 200 1024 1 16 2>&1 | tail -1

#!/bin/bash
# round 5: the low-rank columns in single precision — the LR parity tests, the kernel at width, the radon job
python -m pytest tests/test_gpu_low_rank.py tests/test_gpu_density.py -q 2>&1 | tail -4
python scratch/lr_reg.py 1000 1024 16 2>&1 | grep -v Warn | tail -2
python scratch/lr_reg.py 173 512 4 2>&1 | grep -v Warn | tail -2
python scratch/lr_reg.py 4000 1024 8 2>&1 | grep -v Warn | tail -2
python scratch/r5_lowrank.py radon 2>&1 | grep -v Warn | head -4

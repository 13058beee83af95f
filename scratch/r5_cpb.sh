#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for c in 4 2; do NPHIP_JIT_CPB=$c python scratch/r5_cpb.py 512 2>&1 | grep "^\["; done
for c in 4 2 1; do NPHIP_JIT_CPB=$c python scratch/r5_cpb.py 256 2>&1 | grep "^\["; done
for c in 4 1; do NPHIP_JIT_CPB=$c python scratch/r5_cpb.py 128 2>&1 | grep "^\["; done
python scratch/r5_cpb.py 512 2>&1 | grep "^\["
} | tee gpurun_out/r5_cpb.txt

#!/bin/bash
# round 6, GPU call 1: dense probe + the GPU test suite as it stands + kernel trace of the dense job
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python scratch/r6_dense_probe.py > gpurun_out/r6_dense_probe.txt 2>&1
echo "probe rc $?" >> gpurun_out/r6_dense_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r6_gpu_tests_call1.txt 2>&1
tail -5 gpurun_out/r6_gpu_tests_call1.txt
cat gpurun_out/r6_dense_probe.txt

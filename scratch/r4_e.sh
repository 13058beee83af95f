#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_parity.py -k "callback or device" -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_api.py tests/test_gpu_density.py tests/test_gpu_symbolic.py -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
for a in "1000 1024 0 16 400 100" "1000 1024 0 0 400 100" "173 512 0 16 400 100" "200 1024 0 16"; do python scratch/cbtime.py $a 2>&1 | tail -1; done
NPHIP_CB_SERIAL=1 python scratch/cbtime.py 1000 1024 0 16 400 100 2>&1 | tail -1
NPHIP_NO_CB_SPLIT=1 python scratch/cbtime.py 1000 1024 0 16 400 100 2>&1 | tail -1
bash scratch/r4_kt.sh split_d1000 "" 1000 1024 0 16 400 100

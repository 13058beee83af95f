#!/bin/bash
# the GPU suite file by file (a crash in one file does not hide the others), each under its own timeout
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R; : > $O/r6c_gpu_tests_by_file.txt
for f in tests/test_gpu_*.py; do
  echo "== $f" >> $O/r6c_gpu_tests_by_file.txt
  timeout 900 python -m pytest $f -m gpu -q 2>&1 | grep -v "^  File\|^Extension" | grep -E "passed|failed|error|Error|fault|Aborted|FAILED" | tail -6 >> $O/r6c_gpu_tests_by_file.txt
done
cat $O/r6c_gpu_tests_by_file.txt

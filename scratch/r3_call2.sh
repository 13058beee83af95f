#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 > $O/r3_gpu_tests_2.log 2>&1; echo "all tests rc=$?"; tail -12 $O/r3_gpu_tests_2.log
for rep in 1 2; do
for v in r2 r3a cur; do
  if [ $v = cur ]; then lib=$R/nutpie_amd/libnutpie_hip.so; else lib=$R/scratch/libs/libnutpie_hip_$v.so; fi
  NUTPIE_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-job > $O/r3_ab_$v_$rep.json 2>$O/r3_ab_$v.err
  python -c "
import json,sys
d=json.load(open('$O/r3_ab_$v_$rep.json'))
print('$v', $rep, round(d['value']/1e6,2), 'M/s  kernel', round(d['roofline']['avg_kernel_ms'],3), 'ms  tuning', round(d['tuning_phase']['leapfrogs_per_s_kernel_time']/1e6,1))"
done; done

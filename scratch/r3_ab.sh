#!/bin/bash
# usage: scratch/r3_ab.sh <reps> <lib names under scratch/libs (without .so), "cur" = the in-tree library> ...: bench.py per variant, same box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
reps=$1; shift
for rep in $(seq 1 $reps); do
for v in "$@"; do
  if [ $v = cur ]; then lib=$R/nutpie_amd/libnutpie_hip.so; else lib=$R/scratch/libs/$v.so; fi
  NUTPIE_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-job $BENCH_ARGS > $O/ab_${v}_${rep}.json 2>$O/ab_${v}.err || { echo "$v failed"; tail -3 $O/ab_${v}.err; continue; }
  python -c "
import json
d=json.load(open('$O/ab_${v}_${rep}.json'))
print('$v', $rep, round(d['value']/1e6,2), 'M/s  kernel', round(d['roofline']['avg_kernel_ms'],3), 'ms  tuning', round(d['tuning_phase']['leapfrogs_per_s_kernel_time']/1e6,1))"
done; done

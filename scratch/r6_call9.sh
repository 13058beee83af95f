#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6_gpu_tests.txt 2>&1
tail -6 gpurun_out/r6_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err
echo "bench rc $?"
tail -3 gpurun_out/r6_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r6_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'roofline', {k:d['roofline'].get(k) for k in ('bound','frac','achieved','peak')})
print('job', {k:d['job'].get(k) for k in ('seconds','leapfrogs_per_s','ess_dims','ess_min','ess_min_per_s','ess_seconds_on_device','ess_cross_check')})
for k,v in d['other_configs'].items():
    print(k, {kk: v.get(kk) for kk in ('leapfrogs_per_s','gpu_over_cpu','gpu_over_cpu_tuned','us_per_round','leg_wall_s','error')}, (v.get('cpu_baseline') or {}).get('value'), ((v.get('cpu_baseline') or {}).get('tuned') or {}).get('value'))
    if 'roofline' in v and v['roofline']: print('   roofline', {kk: v['roofline'].get(kk) for kk in ('bound','frac','achieved','peak')})
    if 'bounded_job' in v: print('   bounded', v['bounded_job'].get('leapfrogs_per_s'), v['bounded_job_launch_per_evaluation'].get('leapfrogs_per_s'))
c5=d['config5_shard']; print('config5', c5.get('value'), c5.get('gpu_over_cpu'), (c5.get('cpu_baseline') or {}).get('value'), c5.get('error'))
print('cpu', d['cpu_baseline']['value'], d['gpu_over_cpu'])
P

#!/bin/bash
# same-box A/B of bench.py's timed region over developer libraries: usage r4_ab.sh lib1 lib2 ... (names under scratch/libs, "" = shipped)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for l in "$@"; do
    if [ "$l" = "shipped" ]; then unset NUTPIE_HIP_LIB; else export NUTPIE_HIP_LIB=$R/scratch/libs/$l.so; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-job --no-other-configs --no-config5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', '%.1f M/s' % (d['value']/1e6), 'kernel %.3f ms' % d['roofline']['avg_kernel_ms'])"
  done
done

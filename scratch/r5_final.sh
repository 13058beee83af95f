#!/bin/bash
# round 5, final GPU call: the whole GPU test suite, the driver-shaped bench line, rocprofv3 kernel statistics of the same command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -4
python bench.py > $O/r5_bench.json 2> $O/r5_bench.err; tail -c 200 $O/r5_bench.err
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_r5
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r5 -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --steps 20) > $O/r5_bench_under_rocprof.json 2> /tmp/kt_r5.err
python $R/profiles/summarize.py $(ls /tmp/kt_r5/*/*_results.db /tmp/kt_r5/*_results.db 2>/dev/null | head -1) 20 > $O/r5_bench_kernel_stats.txt; head -6 $O/r5_bench_kernel_stats.txt; tail -2 $O/r5_bench_kernel_stats.txt
rm -rf /tmp/kt_r5c5
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r5c5 -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --dim 10000 --steps 20) > $O/r5_bench_d10000_under_rocprof.json 2> /tmp/kt_r5c5.err
python $R/profiles/summarize.py $(ls /tmp/kt_r5c5/*/*_results.db /tmp/kt_r5c5/*_results.db 2>/dev/null | head -1) 20 > $O/r5_bench_d10000_kernel_stats.txt; tail -2 $O/r5_bench_d10000_kernel_stats.txt

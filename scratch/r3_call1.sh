#!/bin/bash
# round 3, GPU call 1: tests, bench line, kernel-trace stats, PMC at the timed configuration (D = 1000 and D = 10 000)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -x -k "divergence or reference_fixtures or halfnormal or bridgestan" > $O/r3_gpu_tests_new.log 2>&1; echo "new tests rc=$?"; tail -5 $O/r3_gpu_tests_new.log
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 > $O/r3_gpu_tests_1.log 2>&1; echo "all tests rc=$?"; tail -5 $O/r3_gpu_tests_1.log
timeout 600 python bench.py > $O/r3_bench_1.json 2> $O/r3_bench_1.err; echo "bench rc=$?"; head -c 600 $O/r3_bench_1.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r3_kt_d1000 -o kt -- python $R/bench.py --no-cpu-baseline --no-job > $O/r3_bench_under_rocprof.json 2> $O/r3_kt_d1000.err
python $R/profiles/summarize.py $(ls $O/r3_kt_d1000/*/*_results.db $O/r3_kt_d1000/*_results.db 2>/dev/null | head -1) > $O/r3_bench_kernel_stats.txt 2>&1; head -5 $O/r3_bench_kernel_stats.txt
cd $R
timeout 900 bash scratch/pmc_bench.sh r3_d1000 ""
python profiles/pmc_summary.py $(dirname $(ls $O/pmc_r3_d1000/*/*_results.db $O/pmc_r3_d1000/*_results.db 2>/dev/null | head -1)) -97 > $O/r3_d1000_timed_config_pmc.txt 2>&1; cat $O/r3_d1000_timed_config_pmc.txt
timeout 1200 bash scratch/pmc_bench.sh r3_d10000 "--dim 10000"
python profiles/pmc_summary.py $(dirname $(ls $O/pmc_r3_d10000/*/*_results.db $O/pmc_r3_d10000/*_results.db 2>/dev/null | head -1)) -38 > $O/r3_d10000_timed_config_pmc.txt 2>&1; cat $O/r3_d10000_timed_config_pmc.txt
du -sh $O; rm -rf $O/pmc_r3_d1000 $O/pmc_r3_d10000 $O/r3_kt_d1000

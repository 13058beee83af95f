#!/bin/bash
# round 3: the judged numbers — GPU suite, bench line (both dims), kernel-trace stats, PMC at the timed configuration
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; T=${1:-r3_final}
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 > $O/${T}_gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/${T}_gpu_tests.log | tail -2
timeout 600 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"; head -c 200 $O/${T}_bench.json; echo
timeout 600 python bench.py --dim 10000 > $O/${T}_bench_d10000.json 2> $O/${T}_bench_d10000.err; echo "bench d10000 rc=$?"; head -c 200 $O/${T}_bench_d10000.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_kt -o kt -- python $R/bench.py --no-cpu-baseline --no-job > $O/${T}_bench_under_rocprof.json 2> $O/${T}_kt.err
python $R/profiles/summarize.py $(ls $O/${T}_kt/*/*_results.db $O/${T}_kt/*_results.db 2>/dev/null | head -1) > $O/${T}_bench_kernel_stats.txt 2>&1; head -4 $O/${T}_bench_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${T}_kt10k -o kt -- python $R/bench.py --dim 10000 --no-cpu-baseline --no-job > $O/${T}_bench_d10000_under_rocprof.json 2> $O/${T}_kt10k.err
python $R/profiles/summarize.py $(ls $O/${T}_kt10k/*/*_results.db $O/${T}_kt10k/*_results.db 2>/dev/null | head -1) > $O/${T}_bench_d10000_kernel_stats.txt 2>&1; head -4 $O/${T}_bench_d10000_kernel_stats.txt
cd $R
timeout 900 bash scratch/pmc_bench.sh ${T}_d1000 "" > /dev/null
python profiles/pmc_summary.py $(dirname $(ls $O/pmc_${T}_d1000/*/*_results.db $O/pmc_${T}_d1000/*_results.db 2>/dev/null | head -1)) -97 > $O/${T}_d1000_timed_config_pmc.txt 2>&1; cat $O/${T}_d1000_timed_config_pmc.txt
timeout 1200 bash scratch/pmc_bench.sh ${T}_d10000 "--dim 10000" > /dev/null
python profiles/pmc_summary.py $(dirname $(ls $O/pmc_${T}_d10000/*/*_results.db $O/pmc_${T}_d10000/*_results.db 2>/dev/null | head -1)) -38 > $O/${T}_d10000_timed_config_pmc.txt 2>&1; cat $O/${T}_d10000_timed_config_pmc.txt
timeout 300 python scratch/c3gen.py 512 2>&1 | grep -v amdgpu.ids > $O/${T}_config3.txt; cat $O/${T}_config3.txt
rm -rf $O/pmc_${T}_d1000 $O/pmc_${T}_d10000 $O/${T}_kt $O/${T}_kt10k

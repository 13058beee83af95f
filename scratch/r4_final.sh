#!/bin/bash
# round 4: the judged numbers — GPU suite, smoke, bench line (plain and under torch.distributed.run), kernel-trace stats of the same command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_final; mkdir -p $O
cd $R
timeout 1100 python -m pytest tests -m gpu -q --maxfail=15 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 150 $O/bench.json; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-job --no-other-configs > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "bench under torch.distributed.run rc=$?"; tail -1 $O/bench_torchrun.json | head -c 150; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r4_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-job --no-other-configs --no-config5 > $O/bench_under_rocprof.json 2> $O/kt.err
python $R/profiles/summarize.py $(ls /tmp/r4_kt/*/*_results.db /tmp/r4_kt/*_results.db 2>/dev/null | head -1) 20 > $O/bench_kernel_stats.txt 2>&1; head -4 $O/bench_kernel_stats.txt; tail -1 $O/bench_kernel_stats.txt

#!/bin/bash
# round 6, third validation call (after the call placement / draw-end passes of the last session): the whole GPU test suite, smoke, the driver-shaped
# bench line (also as the driver starts it for N > 1), rocprofv3 kernel statistics of the same command at D = 1000 and D = 10 000
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/pmc
cd $R
cat $O/r6c_gpu_tests_by_file.txt | grep -E "passed|failed" | tr "\n" " "; echo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r6c_bench.json 2> $O/r6c_bench.err; tail -c 200 $O/r6c_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-other-configs > $O/r6c_bench_under_torch_distributed_run.json 2> $O/r6c_tdr.err; tail -c 300 $O/r6c_bench_under_torch_distributed_run.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_r6c
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r6c -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --steps 20) > $O/r6c_bench_under_rocprof.json 2> /tmp/kt_r6c.err
python $R/profiles/summarize.py $(ls /tmp/kt_r6c/*/*_results.db /tmp/kt_r6c/*_results.db 2>/dev/null | head -1) 20 > $O/r6c_bench_kernel_stats.txt; head -4 $O/r6c_bench_kernel_stats.txt; tail -2 $O/r6c_bench_kernel_stats.txt
rm -rf /tmp/kt_r6c5
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r6c5 -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --dim 10000 --steps 20) > $O/r6c_bench_d10000_under_rocprof.json 2> /tmp/kt_r6c5.err
python $R/profiles/summarize.py $(ls /tmp/kt_r6c5/*/*_results.db /tmp/kt_r6c5/*_results.db 2>/dev/null | head -1) 20 > $O/r6c_bench_d10000_kernel_stats.txt; tail -2 $O/r6c_bench_d10000_kernel_stats.txt

#!/bin/bash
# round 5, final GPU call: the whole GPU test suite, smoke, the driver-shaped bench line (also as the driver starts it for N > 1), rocprofv3 kernel
# statistics of the same command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r5_gpu_tests.txt; cat $O/r5_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r5_bench.json 2> $O/r5_bench.err; tail -c 200 $O/r5_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-other-configs > $O/r5_bench_under_torch_distributed_run.json 2> $O/r5_tdr.err; tail -c 300 $O/r5_bench_under_torch_distributed_run.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_r5
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r5 -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --steps 20) > $O/r5_bench_under_rocprof.json 2> /tmp/kt_r5.err
python $R/profiles/summarize.py $(ls /tmp/kt_r5/*/*_results.db /tmp/kt_r5/*_results.db 2>/dev/null | head -1) 20 > $O/r5_bench_kernel_stats.txt; head -4 $O/r5_bench_kernel_stats.txt; tail -2 $O/r5_bench_kernel_stats.txt
rm -rf /tmp/kt_r5c5
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r5c5 -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --dim 10000 --steps 20) > $O/r5_bench_d10000_under_rocprof.json 2> /tmp/kt_r5c5.err
python $R/profiles/summarize.py $(ls /tmp/kt_r5c5/*/*_results.db /tmp/kt_r5c5/*_results.db 2>/dev/null | head -1) 20 > $O/r5_bench_d10000_kernel_stats.txt; tail -2 $O/r5_bench_d10000_kernel_stats.txt

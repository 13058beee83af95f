#!/bin/bash
# round 4: the judged numbers — GPU suite, smoke, bench line (plain and under torch.distributed.run), kernel-trace stats, PMC at the timed configuration
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_final; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=15 > $O/gpu_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $O/gpu_tests.log | tail -2
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 150 $O/bench.json; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-job --no-other-configs > $O/bench_torchrun.json 2> $O/bench_torchrun.err; echo "bench under torch.distributed.run rc=$?"; tail -1 $O/bench_torchrun.json | head -c 150; echo
timeout 600 python bench.py --dim 10000 --no-config5 --no-other-configs > $O/bench_d10000.json 2> $O/bench_d10000.err; echo "bench d10000 rc=$?"; head -c 150 $O/bench_d10000.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/r4_kt -o kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-job --no-other-configs --no-config5 > $O/bench_under_rocprof.json 2> $O/kt.err
python $R/profiles/summarize.py $(ls /tmp/r4_kt/*/*_results.db /tmp/r4_kt/*_results.db 2>/dev/null | head -1) > $O/bench_kernel_stats.txt 2>&1; head -4 $O/bench_kernel_stats.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/r4_pmc -o $name -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-job --no-other-configs --no-config5 > /tmp/r4_pmc_$name.log 2>&1
done
python $R/profiles/pmc_summary.py /tmp/r4_pmc -20 > $O/d1000_timed_config_pmc.txt 2>&1; cat $O/d1000_timed_config_pmc.txt

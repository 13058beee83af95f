#!/bin/bash
# round 6, second validation call (after the low-rank estimator kernel, the dense form choice, the lean kernels beyond 20 chunks): the whole GPU test
# suite, smoke, the driver-shaped bench line (also as the driver starts it for N > 1), rocprofv3 kernel statistics of the same command, kernel
# statistics and PMC passes of the estimator kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/pmc
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r6_gpu_tests.txt; cat $O/r6_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r6_bench.json 2> $O/r6_bench.err; tail -c 200 $O/r6_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-other-configs > $O/r6_bench_under_torch_distributed_run.json 2> $O/r6_tdr.err; tail -c 300 $O/r6_bench_under_torch_distributed_run.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_r6
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r6 -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --steps 20) > $O/r6_bench_under_rocprof.json 2> /tmp/kt_r6.err
python $R/profiles/summarize.py $(ls /tmp/kt_r6/*/*_results.db /tmp/kt_r6/*_results.db 2>/dev/null | head -1) 20 > $O/r6_bench_kernel_stats.txt; head -4 $O/r6_bench_kernel_stats.txt; tail -2 $O/r6_bench_kernel_stats.txt
# the estimator kernel: statistics of 10 dispatches on real windows (243 chains), then PMC passes
rm -rf /tmp/kt_r6e
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r6e -o kt -- python scratch/r6_lr_kernel_job.py 10) > $O/r6_lr_kernel_under_rocprof.txt 2> /tmp/kt_r6e.err
python - <<PY > $O/r6_low_rank_estimator_kernel_stats.txt
import sqlite3, glob
f = (glob.glob("/tmp/kt_r6e/*/*_results.db") + glob.glob("/tmp/kt_r6e/*_results.db"))[0]
db = sqlite3.connect(f); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
print("# rocprofv3 --kernel-trace --stats -- python scratch/r6_lr_kernel_job.py 10: dispatches of k_lr_estimate (the last 10 are 243 chains on the window [240, 340) with the device to themselves)")
try:
    rows = list(cur.execute("select name, start, duration from kernels where name like '%k_lr_estimate%' order by start"))
except Exception as e:
    rows = []
    print("tables:", tabs, e)
d = [r[2] / 1e3 for r in rows]
if d:
    print(f"dispatches {len(d)}; all: mean {sum(d) / len(d):.1f} us; the last 10: mean {sum(d[-10:]) / 10:.1f} us, min {min(d[-10:]):.1f}, max {max(d[-10:]):.1f}")
PY
cat $O/r6_low_rank_estimator_kernel_stats.txt; cat $O/r6_lr_kernel_under_rocprof.txt | tail -1
PMC_KERNEL=k_lr_estimate bash $R/scratch/r5_pmc.sh r6_low_rank_estimator -10 python scratch/r6_lr_kernel_job.py 10
cd /tmp
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_r6_lr_lds -o $name -- python scratch/r6_lr_kernel_job.py 10) > /tmp/pmc_r6_lr_$name.log 2>&1 || tail -3 /tmp/pmc_r6_lr_$name.log
done
PMC_KERNEL=k_lr_estimate python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_r6_lr_lds/*/*_results.db /tmp/pmc_r6_lr_lds/*_results.db 2>/dev/null | head -1)) -10 > $O/pmc/r6_low_rank_estimator_lds.txt
cat $O/pmc/r6_low_rank_estimator_lds.txt

#!/bin/bash
# round 6, validation call: the whole GPU test suite, smoke, the driver-shaped bench line (also as the driver starts it for N > 1), rocprofv3 kernel
# statistics of the same command, PMC passes of the dense resident kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/pmc
cd $R
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/r6_gpu_tests.txt; cat $O/r6_gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > $O/r6_bench.json 2> $O/r6_bench.err; tail -c 200 $O/r6_bench.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-other-configs > $O/r6_bench_under_torch_distributed_run.json 2> $O/r6_tdr.err; tail -c 300 $O/r6_bench_under_torch_distributed_run.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_r6
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r6 -o kt -- python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --steps 20) > $O/r6_bench_under_rocprof.json 2> /tmp/kt_r6.err
python $R/profiles/summarize.py $(ls /tmp/kt_r6/*/*_results.db /tmp/kt_r6/*_results.db 2>/dev/null | head -1) 20 > $O/r6_bench_kernel_stats.txt; head -4 $O/r6_bench_kernel_stats.txt; tail -2 $O/r6_bench_kernel_stats.txt
rm -rf /tmp/kt_r6d
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r6d -o kt -- python scratch/r6_dense_job.py 12) > $O/r6_dense_under_rocprof.txt 2> /tmp/kt_r6d.err
python $R/profiles/summarize.py $(ls /tmp/kt_r6d/*/*_results.db /tmp/kt_r6d/*_results.db 2>/dev/null | head -1) 12 > $O/r6_dense_kernel_stats.txt; head -3 $O/r6_dense_kernel_stats.txt; tail -1 $O/r6_dense_kernel_stats.txt; cat $O/r6_dense_under_rocprof.txt
bash $R/scratch/r5_pmc.sh r6_dense_resident -12 python scratch/r6_dense_job.py 12
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_r6_dense_mfma -o $name -- python scratch/r6_dense_job.py 12) > /tmp/pmc_r6_$name.log 2>&1 || tail -3 /tmp/pmc_r6_$name.log
done
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_r6_dense_mfma/*/*_results.db /tmp/pmc_r6_dense_mfma/*_results.db 2>/dev/null | head -1)) -12 > $O/pmc/r6_dense_resident_mfma.txt
cat $O/pmc/r6_dense_resident_mfma.txt

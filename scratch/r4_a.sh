#!/bin/bash
# round 4, GPU call A: the new bench line; baseline of the launch-per-evaluation kernels (kernel-trace + PMC)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r4a
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
echo "bench rc=$?"; tail -c 600 gpurun_out/r4a/bench.err
python -m pytest tests/test_gpu_parity.py -k "device_callback_bit" -x -q 2>&1 | tail -5
cd /tmp; export TMPDIR=/tmp
for w in 0 4 8; do
  for g in 0 16; do python $R/scratch/cbtime.py 1000 1024 $w $g 2>&1 | tail -1; done
done > $R/gpurun_out/r4a/cbtime.txt 2>&1
python $R/scratch/cbtime.py 173 512 0 16 >> $R/gpurun_out/r4a/cbtime.txt 2>&1
cat $R/gpurun_out/r4a/cbtime.txt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4a/kt -o kt -- python $R/scratch/cbtime.py 1000 1024 0 0 > $R/gpurun_out/r4a/kt.log 2>&1
python $R/profiles/summarize.py $(ls $R/gpurun_out/r4a/kt/*/*_results.db $R/gpurun_out/r4a/kt/*_results.db 2>/dev/null | head -1) > $R/gpurun_out/r4a/kt_stats.txt 2>&1; head -12 $R/gpurun_out/r4a/kt_stats.txt
rocprofv3 -L > $R/gpurun_out/r4a/counters.txt 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_IFETCH SQ_WAVES"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/r4a/pmc -o $name -- python $R/scratch/cbtime.py 1000 1024 0 0 > $R/gpurun_out/r4a/pmc_$name.log 2>&1
done
python $R/profiles/pmc_summary.py $R/gpurun_out/r4a/pmc > $R/gpurun_out/r4a/pmc_summary.txt 2>&1; cat $R/gpurun_out/r4a/pmc_summary.txt
find $R/gpurun_out/r4a -name "*.db" -size +20M -delete

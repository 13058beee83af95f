#!/bin/bash
# PMC passes of bench.py itself (the timed configuration): usage: scratch/pmc_bench.sh <tag> "<bench.py args>"
# one --pmc group per pass (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass -d $R/gpurun_out/pmc_$1 -o $name -- python $R/bench.py --no-cpu-baseline --no-job $2 > $R/gpurun_out/pmc_$1_$name.log 2>&1
done
ls $R/gpurun_out/pmc_$1

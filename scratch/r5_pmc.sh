#!/bin/bash
# PMC passes of one command (one --pmc group per pass, kernel trace only): usage scratch/r5_pmc.sh <tag> <last_n or 0> <command...>
# -> gpurun_out/pmc/<tag>.txt (profiles/pmc_summary.py) with the command's own "leapfrogs= launches=" line on top
R=$GRAFT_REPO_ROOT; tag=$1; last=$2; shift 2
cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc; rm -rf /tmp/pmc_$tag
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o $name -- "$@") > /tmp/pmc_${tag}_$name.log 2>&1 || tail -3 /tmp/pmc_${tag}_$name.log
done
grep -h "leapfrogs=" /tmp/pmc_${tag}_FETCH_SIZE.log | tail -1 > $R/gpurun_out/pmc/$tag.txt
if [ "$last" = "auto" ]; then last=-$(grep -h -o " launches=[0-9]*" /tmp/pmc_${tag}_FETCH_SIZE.log | tail -1 | cut -d= -f2); fi
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_$tag/*/*_results.db /tmp/pmc_$tag/*_results.db 2>/dev/null | head -1)) $last >> $R/gpurun_out/pmc/$tag.txt
cat $R/gpurun_out/pmc/$tag.txt

#!/bin/bash
# instruction-cache counters of one scratch/ab.py run per developer library: usage scratch/r6_icache_pmc.sh <lib name under scratch/libs> ...
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc
for v in "$@"; do
  rm -rf /tmp/pmc_ic_$v
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQC_ICACHE_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU"; do
    name=$(echo $pass | tr ' ' '_' | cut -c1-40)
    (cd $R && NUTPIE_HIP_LIB=$R/scratch/libs/$v.so rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_ic_$v -o $name -- python scratch/ab.py "run(1000,1024,False,W=1,E=2048,steps=10,warm=10)") > /tmp/pmc_ic_${v}_$name.log 2>&1 || tail -3 /tmp/pmc_ic_${v}_$name.log
  done
  echo "== $v" > $R/gpurun_out/pmc/icache_$v.txt
  python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_ic_$v/*/*_results.db /tmp/pmc_ic_$v/*_results.db 2>/dev/null | head -1)) 10 >> $R/gpurun_out/pmc/icache_$v.txt
  cat $R/gpurun_out/pmc/icache_$v.txt
done

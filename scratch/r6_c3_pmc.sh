#!/bin/bash
# round 6: PMC passes of config 3's resident density kernel at the round-6 build (the round-5 passes + where the waiting goes)
R=$GRAFT_REPO_ROOT; cd $R
bash scratch/r5_pmc.sh r6_config3_compiled_density 0 python scratch/r5_pmc_jobs.py c3 > /dev/null
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pmc_r6_c3b
for pass in "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_r6_c3b -o $name -- python scratch/r5_pmc_jobs.py c3) > /tmp/pmc_r6_c3b_$name.log 2>&1 || tail -3 /tmp/pmc_r6_c3b_$name.log
done
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_r6_c3b/*/*_results.db /tmp/pmc_r6_c3b/*_results.db 2>/dev/null | head -1)) 0 > $R/gpurun_out/pmc/r6_config3_compiled_density_waits.txt
cat $R/gpurun_out/pmc/r6_config3_compiled_density.txt $R/gpurun_out/pmc/r6_config3_compiled_density_waits.txt

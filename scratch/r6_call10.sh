#!/bin/bash
# round 6: profiles of the dense Gaussian's resident kernel: phase attribution, rocprofv3 kernel statistics, PMC passes (traffic, issue, MFMA)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O/pmc
cd $R
python scratch/r6_dense_job.py 12 2>&1 | grep -v amdgpu.ids > $O/r6_dense_job.txt
NPHIP_DG_VARIANT=32 python scratch/r6_dense_job.py 12 2>&1 | grep -v amdgpu.ids >> $O/r6_dense_job.txt
cat $O/r6_dense_job.txt
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt_r6d
(cd $R && rocprofv3 --kernel-trace --stats -d /tmp/kt_r6d -o kt -- python scratch/r6_dense_job.py 12) > $O/r6_dense_under_rocprof.txt 2> /tmp/kt_r6d.err
python $R/profiles/summarize.py $(ls /tmp/kt_r6d/*/*_results.db /tmp/kt_r6d/*_results.db 2>/dev/null | head -1) 12 > $O/r6_dense_kernel_stats.txt; head -6 $O/r6_dense_kernel_stats.txt; tail -2 $O/r6_dense_kernel_stats.txt
bash scratch/r5_pmc.sh r6_dense_resident -12 python scratch/r6_dense_job.py 12
cd /tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  (cd $R && rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_r6_dense_resident -o $name -- python scratch/r6_dense_job.py 12) > /tmp/pmc_r6_$name.log 2>&1 || tail -3 /tmp/pmc_r6_$name.log
done
python $R/profiles/pmc_summary.py $(dirname $(ls /tmp/pmc_r6_dense_resident/*/*_results.db /tmp/pmc_r6_dense_resident/*_results.db 2>/dev/null | head -1)) -12 > $O/pmc/r6_dense_resident_all.txt
cat $O/pmc/r6_dense_resident_all.txt

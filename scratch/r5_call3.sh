#!/bin/bash
# round 5: PMC passes of the kernels at the final build (after the halving reductions and the single-precision columns)
B="python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --steps 20"
scratch/r5_pmc.sh r5_d1000_timed_config -20 $B > /dev/null
scratch/r5_pmc.sh r5_d10000_timed_config -20 $B --dim 10000 > /dev/null
scratch/r5_pmc.sh r5_config3_compiled_density 0 python scratch/r5_pmc_jobs.py c3 > /dev/null
scratch/r5_pmc.sh r5_config3_traced_torch_density 0 python scratch/r5_pmc_jobs.py c3traced > /dev/null
scratch/r5_pmc.sh r5_low_rank_d173_k4 auto python scratch/r5_pmc_jobs.py lr 173 4 > /dev/null
scratch/r5_pmc.sh r5_low_rank_d1000_k16 auto python scratch/r5_pmc_jobs.py lr 1000 16 > /dev/null
head -2 gpurun_out/pmc/r5_*.txt | cut -c1-160

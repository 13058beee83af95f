#!/bin/bash
# round 5, last GPU call: PMC of config 3's two compiled kernels at the final generator / build, then scratch/r5_final.sh (GPU suite, smoke, bench, kernel stats)
scratch/r5_pmc.sh r5_config3_compiled_density 0 python scratch/r5_pmc_jobs.py c3 > /dev/null
scratch/r5_pmc.sh r5_config3_traced_torch_density 0 python scratch/r5_pmc_jobs.py c3traced > /dev/null
head -2 gpurun_out/pmc/r5_config3_*.txt
scratch/r5_final.sh

#!/bin/bash
B="python bench.py --no-cpu-baseline --no-job --no-other-configs --no-config5 --steps 20"
scratch/r5_pmc.sh r5_d1000_timed_config -20 $B > /dev/null
scratch/r5_pmc.sh r5_d10000_timed_config -20 $B --dim 10000 > /dev/null
scratch/r5_final.sh

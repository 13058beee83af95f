#!/bin/bash
# round 6: the step at D = 1025 — one wave per chain with 9 .. 12 chunks per lane and no LDS ring (dev libraries) against the shipped two waves per chain
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for pair in "9 1100" "9 1152" "10 1280" "12 1536"; do set -- $pair
  echo "== D=$2: W=1, $1 chunks per lane, no ring (dev library)"; NUTPIE_HIP_LIB=$R/scratch/libs/w1_$1.so python scratch/ab.py "run($2, 1024, False, W=1, E=1024, steps=12, warm=30)" 2>&1 | grep "^dim"
  echo "== D=$2: shipped (W=2)"; python scratch/ab.py "run($2, 1024, False, E=1024, steps=12, warm=30)" 2>&1 | grep "^dim"
done
} > gpurun_out/r6_step_at_d1025.txt 2>&1
cat gpurun_out/r6_step_at_d1025.txt

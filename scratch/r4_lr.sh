#!/bin/bash
# Round 4: the low-rank metric on the register-resident leaf — kernel against kernel, jobs, the driver's time line.
# (gpurun -- 'bash scratch/r4_lr.sh'; output: gpurun_out/r4_lr.txt -> profiles/r4_low_rank_register_kernel.txt)
out=gpurun_out/r4_lr.txt; : > $out
for a in "173 512 4" "173 512 8" "300 1024 4" "1000 1024 4" "1000 1024 16" "500 256 16"; do python scratch/lr_reg.py $a 2>&1 | grep "^D=" >> $out; done
python scratch/lr_density.py 2>&1 | grep -v amdgpu.ids | grep "D=" >> $out
python scratch/lowrank_demo_compiled.py 2>&1 | grep "^D=" >> $out
for m in radon demo500; do python scratch/lr_driver_time.py $m 2>&1 | grep "^$m\|^hand-ins" | cut -c1-330 | tail -2 >> $out; done
python scratch/lr_launch_log.py radon lockstep 2>&1 | grep "^total" | cut -c1-200 | sed 's/^/lockstep driver (round 3): /' >> $out
python scratch/lr_launch_log.py radon 2>&1 | grep "^total" | cut -c1-200 | sed 's/^/per-chain hand-in: /' >> $out
cat $out

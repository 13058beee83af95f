#!/bin/bash
# the low-rank metric on the register-resident leaf with two / four waves per chain against the memory-resident kernels
for a in "1500 1024 4" "2000 1024 8" "3000 1024 4" "4000 512 16"; do python scratch/lr_reg.py $a 2>&1 | grep "^D="; done

#!/bin/bash
# round 6: lean 4-wave kernels with 22 / 24 chunks per wave — the gradient rebuilt at the start of every leaf (REGRAD) against carried (the first spilling builds)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== shipped library (gradient carried, 216 / 456 bytes of scratch per lane)"
python scratch/ab.py "run(11264, 1024, False, E=32, steps=12, warm=40)" "run(12000, 1024, False, E=32, steps=12, warm=40)" 2>&1 | grep "^dim"
echo "== gradient rebuilt per leaf (248 / 280 bytes)"
NUTPIE_HIP_LIB=scratch/libs/lean4_22_rg.so python scratch/ab.py "run(11264, 1024, False, W=4, E=32, steps=12, warm=40)" 2>&1 | grep "^dim"
NUTPIE_HIP_LIB=scratch/libs/lean4_24_rg.so python scratch/ab.py "run(12000, 1024, False, W=4, E=32, steps=12, warm=40)" 2>&1 | grep "^dim"
done

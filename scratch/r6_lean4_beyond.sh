#!/bin/bash
# round 6: the lean 4-wave kernel beyond 20 chunks per wave (D > 10 240) with its spills, against the memory-resident kernels that run there today
cd $GRAFT_REPO_ROOT
echo "== shipped: D = 10240 (lean, 20 chunks), 11264 / 12000 / 12288 (memory-resident, 8 waves per chain)"
python scratch/ab.py "run(10240, 1024, False, E=32, steps=12, warm=40)" "run(11264, 1024, False, E=32, steps=12, warm=40)" "run(12000, 1024, False, E=32, steps=12, warm=40)" "run(12288, 1024, False, E=32, steps=12, warm=40)" 2>&1 | grep "^dim"
echo "== developer library, 22 chunks per wave (216 bytes of scratch per lane): D = 11264"
NPHIP_DEV_LEAN4_MAX=24 NUTPIE_HIP_LIB=scratch/libs/lean4_22.so python scratch/ab.py "run(11264, 1024, False, W=4, E=32, steps=12, warm=40)" 2>&1 | grep "^dim"
echo "== developer library, 24 chunks per wave (456 bytes of scratch per lane): D = 12000, 12288"
NPHIP_DEV_LEAN4_MAX=24 NUTPIE_HIP_LIB=scratch/libs/lean4_24.so python scratch/ab.py "run(12000, 1024, False, W=4, E=32, steps=12, warm=40)" "run(12288, 1024, False, W=4, E=32, steps=12, warm=40)" 2>&1 | grep "^dim"

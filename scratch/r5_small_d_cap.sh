#!/bin/bash
# round 5: the small one-wave kernels (2, 3 chunks per lane) with and without the two-waves-per-SIMD register cap, at 1024 and 4096 chains
R=$GRAFT_REPO_ROOT
for pair in "2 256" "3 384"; do set -- $pair
  for chains in 1024 4096; do
    for v in cap nocap cap nocap; do
      echo -n "NV=$1 D=$2 chains=$chains $v: "; NUTPIE_HIP_LIB=$R/scratch/libs/$v$1.so python scratch/ab.py "run($2, $chains, False, E=512, steps=16, warm=8)" 2>&1 | grep "^dim" | sed 's/.*E=512: //'
    done
  done
done

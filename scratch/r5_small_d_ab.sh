#!/bin/bash
# same-box A/B of the halving reductions on the small register kernels (NV = 2, 3, 5 chunks: D = 256, 384, 640)
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for pair in "2 256" "3 384" "5 640"; do set -- $pair
  for v in p h; do
    echo -n "NV=$1 D=$2 $v: "; NUTPIE_HIP_LIB=$R/scratch/libs/$v$1.so python scratch/ab.py "run($2, 1024, False, E=512, steps=24, warm=12)" 2>&1 | grep "^dim" | sed 's/.*E=512: //'
  done
done; done

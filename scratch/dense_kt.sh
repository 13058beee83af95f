#!/bin/bash
# kernel-trace statistics of the dense Gaussian job for the given waves per chain: usage dense_kt.sh <w> ...
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  rm -rf $R/gpurun_out/dense_kt
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dense_kt -o kt -- python $R/scratch/dense_prof.py $w 2>&1 | grep "dense gaussian"
  python $R/profiles/summarize.py $(ls $R/gpurun_out/dense_kt/*/*_results.db $R/gpurun_out/dense_kt/*_results.db 2>/dev/null | head -1) | sed -n 3,5p
done
rm -rf $R/gpurun_out/dense_kt

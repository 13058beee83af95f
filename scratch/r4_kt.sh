#!/bin/bash
# kernel-trace statistics of one cbtime.py run: usage r4_kt.sh <tag> <lib or ""> <cbtime args...>
R=$GRAFT_REPO_ROOT; tag=$1; lib=$2; shift 2
cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/kt
[ -n "$lib" ] && export NUTPIE_HIP_LIB=$R/$lib
rm -rf /tmp/kt_$tag
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -- python $R/scratch/cbtime.py "$@" 2>&1 | grep "scaled normal" | tail -1
python $R/profiles/summarize.py $(ls /tmp/kt_$tag/*/*_results.db /tmp/kt_$tag/*_results.db 2>/dev/null | head -1) | head -8 | tee $R/gpurun_out/kt/$tag.txt

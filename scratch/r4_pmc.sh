#!/bin/bash
# PMC passes of one cbtime.py run: usage r4_pmc.sh <tag> <lib or ""> <cbtime args...>
R=$GRAFT_REPO_ROOT; tag=$1; lib=$2; shift 2
cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc
[ -n "$lib" ] && export NUTPIE_HIP_LIB=$R/$lib
rm -rf /tmp/pmc_$tag
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass -d /tmp/pmc_$tag -o $name -- python $R/scratch/cbtime.py "$@" > /tmp/pmc_$tag_$name.log 2>&1 || tail -3 /tmp/pmc_$tag_$name.log
done
python $R/profiles/pmc_summary.py /tmp/pmc_$tag | tee $R/gpurun_out/pmc/$tag.txt

#!/bin/bash
# HBM traffic + SQ wait counters of selected kernels: scripts/pmc_quick.sh <tag> '<kernel regex>' [bench args]
tag=$1; rx=$2; shift; shift
export TMPDIR=/tmp
out=gpurun_out
dirs=""
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/${tag}_q$i -o q --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained 0 --no-kernel-profile "$@" > $out/${tag}_q$i.log 2>&1
  dirs="$dirs $out/${tag}_q$i"
done
python3 scripts/pmc_kernel.py "$rx" $dirs > $out/${tag}_pmc.txt
rm -rf $dirs
cat $out/${tag}_pmc.txt

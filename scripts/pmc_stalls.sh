#!/bin/bash
# Issue / wait / LDS counters of the SQ block for a set of kernels (separate --pmc passes, summaries only).
# SQ counters only: a pass with the derived TCP_*_sum counters made rocprofv3 abort and hang on this pool.
#   scripts/pmc_stalls.sh <tag> '<kernel regex>' [bench args]
set -e -o pipefail
tag=$1; rx=$2; shift; shift
export TMPDIR=/tmp
out=gpurun_out
i=0
dirs=""
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $out/${tag}_st$i -o st --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained 0 --no-kernel-profile "$@" > $out/${tag}_st$i.log 2>&1
  dirs="$dirs $out/${tag}_st$i"
  echo "pass $i done"
done
python3 scripts/pmc_kernel.py "$rx" $dirs > $out/${tag}_stalls.txt
rm -rf $dirs

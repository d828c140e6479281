#!/bin/bash
# Collect the measurement artifacts of one round on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh <tag> [bench args]
# Writes gpurun_out/<tag>_bench.json, <tag>_kernel_stats.csv, <tag>_pmc_traffic.json; copy them into profiles/.
set -e -o pipefail
tag=$1; shift
steps=10
export TMPDIR=/tmp
out=gpurun_out
mkdir -p $out
python3 bench.py --steps 100 --warmup 10 "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench.json; echo
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o st --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --sustained 0 --no-kernel-profile "$@" > $out/${tag}_stats.log 2>&1
cp $(find $out/${tag}_stats -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc_r -o r --output-format csv -- python3 bench.py --steps $steps --warmup 1 --no-cpu-baseline --sustained 0 --no-kernel-profile "$@" > $out/${tag}_pmc_r.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc_w -o w --output-format csv -- python3 bench.py --steps $steps --warmup 1 --no-cpu-baseline --sustained 0 --no-kernel-profile "$@" > $out/${tag}_pmc_w.log 2>&1
python3 scripts/pmc_traffic.py $(find $out/${tag}_pmc_r -name '*counter_collection.csv' | head -1) $(find $out/${tag}_pmc_w -name '*counter_collection.csv' | head -1) $steps $out/${tag}_pmc_traffic.json ${GRID:-256} ${PREC:-fp64}
# SQ counters of the two particle-mesh kernels (8 SQ slots per pass; GRBM_GUI_ACTIVE rides along in the GRBM block)
if [ -z "$NO_SQ" ]; then
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" \
             "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_FLOPS_FP64 SQ_BUSY_CU_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d $out/${tag}_sq$i -o sq --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustained 0 --no-kernel-profile "$@" > $out/${tag}_sq$i.log 2>&1
  done
  python3 scripts/pmc_sq.py $out/${tag}_kernel_stats.csv $out/${tag}_sq_tile81.json $out/${tag}_sq1 $out/${tag}_sq2 $out/${tag}_sq3 > $out/${tag}_sq_tile81.txt
  rm -rf $out/${tag}_sq1 $out/${tag}_sq2 $out/${tag}_sq3
fi
# keep only the summaries (the raw traces are large)
rm -rf $out/${tag}_stats $out/${tag}_pmc_r $out/${tag}_pmc_w

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
rocprofv3 --kernel-trace --stats -d $out/${tag}_stats -o st --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-profile "$@" > $out/${tag}_stats.log 2>&1
cp $(find $out/${tag}_stats -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $out/${tag}_pmc_r -o r --output-format csv -- python3 bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-kernel-profile "$@" > $out/${tag}_pmc_r.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/${tag}_pmc_w -o w --output-format csv -- python3 bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-kernel-profile "$@" > $out/${tag}_pmc_w.log 2>&1
python3 scripts/pmc_traffic.py $(find $out/${tag}_pmc_r -name '*counter_collection.csv' | head -1) $(find $out/${tag}_pmc_w -name '*counter_collection.csv' | head -1) $steps $out/${tag}_pmc_traffic.json ${GRID:-256} ${PREC:-fp64}
# keep only the summaries (the raw traces are large)
rm -rf $out/${tag}_stats $out/${tag}_pmc_r $out/${tag}_pmc_w

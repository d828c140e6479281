#!/bin/bash
# Non-default force variants at 256^3 (never the headline): bench lines + rocprofv3 kernel summaries.
#   scripts/variants_bench.sh <tag>     -> gpurun_out/<tag>_variants.txt, gpurun_out/<tag>_kernel_stats_<name>.csv
tag=${1:-r03}; out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
line() {  # name, env, bench args
  env $2 timeout -k 10 400 python3 bench.py --steps 40 --warmup 4 --no-cpu-baseline --sustained 0 $3 2>$out/vb.err > $out/vb.json || { echo "FAILED $1"; tail -3 $out/vb.err; return; }
  python3 - "$1" $out/vb.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"].get("kernels") or {}
print("%-34s %8.2f steps/s %7.4f ms frac %.3f | " % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]) +
      " ".join("%s=%.3f" % (n.split("+")[0][:14], v["ms_per_step"]) for n, v in k.items()), flush=True)
PY
  cp $out/vb.json $out/${tag}_bench_$1.json
}
stats() {  # name, bench args
  rocprofv3 --kernel-trace --stats -d $out/vb_stats -o st --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --sustained 0 --no-kernel-profile $2 > $out/vb_stats.log 2>&1
  cp $(find $out/vb_stats -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats_$1.csv && rm -rf $out/vb_stats
}
{
line cic_calch1_tiles "-u X" "--mk 1 --calc-h 1"
line cic_calch1_direct "BCHMC_NO_TILES_LOW=1" "--mk 1 --calc-h 1"
line tsc_calch1_tiles "-u X" "--mk 2 --calc-h 1"
line tsc_calch1_direct "BCHMC_NO_TILES_LOW=1" "--mk 2 --calc-h 1"
line sph_calch3_tiles "-u X" "--mk 3 --calc-h 3"
line sph_calch3_direct "BCHMC_NO_TILES_LOW=1" "--mk 3 --calc-h 3"
line alpt_norsd_planes "-u X" "--no-rsd --alpt"
line alpt_norsd_3d "BCHMC_NO_ALPT_PLANES=1" "--no-rsd --alpt"
line zeld_norsd "-u X" "--no-rsd"
} | tee $out/${tag}_variants.txt
stats cic_calch1 "--mk 1 --calc-h 1"
stats sph_calch3 "--mk 3 --calc-h 3"
stats alpt "--no-rsd --alpt"

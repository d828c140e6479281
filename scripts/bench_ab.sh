#!/bin/bash
# A/B a BCHMC_* environment switch on the default bench: scripts/bench_ab.sh VAR "bench args"
var=$1; shift
for v in 0 1; do
  env $var=$v timeout -k 10 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null > gpurun_out/ab_$v.json || exit 1
  python - "$var=$v" gpurun_out/ab_$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["ms_per_step"], json.dumps(d["roofline"].get("kernels")))
PY
done

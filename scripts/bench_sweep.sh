#!/bin/bash
# Sweep one BCHMC_* environment variable over values on the bench: scripts/bench_sweep.sh VAR "v1 v2 ..." [bench args]
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v timeout -k 10 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --sustained 0 "$@" 2>/dev/null > gpurun_out/sw.json || exit 1
  python - "$var=$v" gpurun_out/sw.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"].get("kernels") or {}
print(sys.argv[1], d["value"], " ".join(f"{n}={v['ms_per_step']:.3f}" for n, v in k.items()))
PY
done

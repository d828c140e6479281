#!/bin/bash
# Same-box A/B of environment settings on the bench: scripts/ab_env.sh [-r ROUNDS] [-a "bench args"] "VAR=a VAR2=b" "VAR=c" ...
# Each quoted argument is one configuration ("-" = no variables); configurations alternate ROUNDS times (default 2).
rounds=2; args=""
while getopts "r:a:" o; do case $o in r) rounds=$OPTARG;; a) args=$OPTARG;; esac; done
shift $((OPTIND-1))
mkdir -p gpurun_out
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    envs=$cfg; [ "$cfg" = "-" ] && envs=""
    env $envs timeout -k 10 300 python bench.py --steps 40 --warmup 4 --no-cpu-baseline --sustained 0 $args 2>gpurun_out/ab_env.err > gpurun_out/ab_env.json || { echo "FAILED: $cfg"; tail -3 gpurun_out/ab_env.err; exit 1; }
    python - "$cfg" gpurun_out/ab_env.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["roofline"].get("kernels") or {}
print("%-46s %8.2f steps/s %7.4f ms | " % (sys.argv[1], d["value"], d["ms_per_step"]) + " ".join("%s=%.3f" % (n.split("+")[0][:14], v["ms_per_step"]) for n, v in k.items()), flush=True)
PY
    grep -h "bchmc:" gpurun_out/ab_env.err | tail -3
  done
done

#!/bin/bash
# long-trajectory rate under different slot policies: scripts/sustained_ab.sh
for cfg in "-" "BCHMC_SORT_CAP_FIXED=1 BCHMC_SORT_CAP=16384" "BCHMC_SORT_CAP_FIXED=1 BCHMC_SORT_CAP=32768" "-"; do
  envs=$cfg; [ "$cfg" = "-" ] && envs="BCHMC_X=0"
  env $envs BCHMC_VERBOSE=1 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-profile --sustained 3.5 2> gpurun_out/sab.err > gpurun_out/sab.json
  python3 -c "
import json; d=json.load(open('gpurun_out/sab.json')); s=d['sustained']; print('$cfg', d['value'], 'sustained', s['value'], s['steps'], s['fifths_steps_per_s'])"
  grep "record slots" gpurun_out/sab.err | tail -4
done

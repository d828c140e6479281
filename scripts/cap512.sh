#!/bin/bash
# record-slot partition at 512^3 fp64: steps/s for pinned partitions and the default policy
for cfg in "$@"; do
  envs=$cfg; [ "$cfg" = "-" ] && envs="BCHMC_X=0"
  env $envs BCHMC_VERBOSE=1 python3 bench.py --nx 512 --steps 20 --warmup 2 --no-cpu-baseline --sustained 0 2> gpurun_out/c512.err > gpurun_out/c512.json
  python3 -c "
import json; d=json.load(open('gpurun_out/c512.json')); k=d['roofline']['kernels']; print('$cfg', d['value'], d['ms_per_step'], ' '.join('%s=%.2f'%(n.split('+')[0][:12],v['ms_per_step']) for n,v in k.items()))"
  grep -i "bchmc" gpurun_out/c512.err | tail -6
done

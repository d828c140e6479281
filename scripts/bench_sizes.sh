#!/bin/bash
# Headline numbers at the other BASELINE grid sizes (run through gpurun): one line per configuration.
run() { python3 bench.py --no-cpu-baseline --sustained 0 "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['grid'], d['dtype'], d['value'], 'steps/s', d['ms_per_step'], 'ms/step', 'frac', round(d['roofline']['frac'],3))"; }
run --nx 64 --steps 200 --warmup 20
run --nx 128 --steps 200 --warmup 20
run --nx 256 --steps 100 --warmup 10
run --nx 256 --steps 100 --warmup 10 --fp32
run --nx 512 --steps 20 --warmup 2 --fp32
run --nx 512 --steps 20 --warmup 2
# BASELINE config 2: 128^3, Poissonian likelihood, Zel'dovich, 50 steps
run --nx 128 --steps 50 --warmup 5 --likelihood 0 --no-rsd

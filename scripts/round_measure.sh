#!/bin/bash
# The measurement set of a round on the GPU box (two gpurun calls: part 1, part 2): scripts/round_measure.sh <tag> <1|2>
tag=${1:-r03}; part=${2:-1}; out=gpurun_out; mkdir -p $out
export TMPDIR=/tmp
if [ "$part" = 1 ]; then
  scripts/profile_round.sh ${tag}_v1 > $out/${tag}_profile_round.log 2>&1; tail -2 $out/${tag}_profile_round.log
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 300 $out/${tag}_bench.json; echo
  python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustained 0 --chains-per-gpu 2 > $out/${tag}_two_chains.json 2> $out/${tag}_two_chains.err
  python3 scripts/attempt_bench.py > $out/${tag}_attempt_bench.json 2> $out/${tag}_attempt_bench.err; cat $out/${tag}_attempt_bench.json
  scripts/variants_bench.sh $tag
else
  scripts/bench_sizes.sh > $out/${tag}_bench_sizes.txt 2>&1; cat $out/${tag}_bench_sizes.txt
  rocprofv3 --kernel-trace --stats -d $out/f32_stats -o st --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --sustained 0 --no-kernel-profile --fp32 > $out/f32_stats.log 2>&1
  cp $(find $out/f32_stats -name '*kernel_stats.csv' | head -1) $out/${tag}_kernel_stats_256f32.csv && rm -rf $out/f32_stats
  python3 scripts/soak.py > $out/${tag}_soak.log 2>&1; tail -5 $out/${tag}_soak.log
fi

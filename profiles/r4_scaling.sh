#!/bin/bash
# Multi-GPU REHEARSAL on one GPU (no multi-GPU node is available to the build): every rank's shard of an N-GPU job rendered alone,
# `bench.py --simulate-shard r/N` for every r of N = 2, 4, 8 on C4 (256 spp per step) and N = 8 on C5 (64 spp per step), one call.
# profiles/r4_scaling.py turns the lines into profiles/r4_scaling.json.   usage: bash profiles/r4_scaling.sh
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r4_scaling; mkdir -p $OUT
cd $R
run() { tag=$1; shift; timeout -k 10 280 python3 bench.py --no-cpu --no-counters "$@" > $OUT/$tag.json 2> $OUT/$tag.err || { echo "FAILED $tag"; tail -3 $OUT/$tag.err; exit 1; }; python3 -c "
import json; d=json.load(open('$OUT/$tag.json')); print('$tag', d['value'], d['ms_per_step'])"; }
run c4_full --steps 3 --warmup 1
for N in 2 4 8; do for ((r=0; r<N; r++)); do run c4_${r}_of_$N --steps 3 --warmup 1 --simulate-shard $r/$N; done; done
run c5_full --scene c5 --width 4096 --height 4096 --vthreads 16 --steps 2 --warmup 1
for ((r=0; r<8; r++)); do run c5_${r}_of_8 --scene c5 --width 4096 --height 4096 --vthreads 16 --steps 2 --warmup 1 --simulate-shard $r/8; done
echo done

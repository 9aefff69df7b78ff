#!/bin/bash
# Round 6, step 13 (beyond the review's item): if the size of the mapped pieces matters that much to forty streams, does it matter to the trace
# kernel's random gathers?  The two arrays it gathers from (quantised nodes 25 MB, padded triangle records 64 MB on C4) re-placed after the
# build: 0 as hipMalloc left them, 1 physically contiguous, 2 in 2 MB chunks, 3 in 64 MB chunks.  Trace-kernel time per step, N processes each.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
N=${1:-3}; SCENES=${2:-"c4 s4 c5"}
{
for sc in $SCENES; do
  case $sc in
    c3) A="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1" ;;
    c4) A="--steps 2 --warmup 1" ;;
    c5) A="--scene c5 --width 4096 --height 4096 --vthreads 8 --steps 4 --warmup 1" ;;
    s4) A="--scene s4 --steps 2 --warmup 1" ;;
  esac
  for i in $(seq 1 $N); do
    for m in 0 1 2 3; do
      ART_SCENE_PLACEMENT=$m python3 $R/bench.py $A --no-cpu --no-counters > $OUT/bm13.json 2> $OUT/bm13.err
      python3 - $OUT/bm13.json $sc $m $i <<PY
import json,sys
f,sc,m,i=sys.argv[1:5]
try:
    L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
    print("%s scene placement %s run %s: trace %.1f ms/step  shade %.3f ms/batch  %.1f Mrays/s" % (sc, m, i, L["trace_ms_per_step"], L["stages"]["shade"]["ms_per_batch"], L["value"]), flush=True)
except Exception as x:
    print(sc, m, i, "FAILED", x, open("$OUT/bm13.err").read()[-300:], flush=True)
PY
    done
  done
done
} 2>&1 | tee $OUT/bimodal_probe13.txt

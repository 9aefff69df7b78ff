#!/bin/bash
# Round 6, review item 1 step (i): what does k_trace_coop lose at 7 / 6 / 5 workgroups per CU (= waves per SIMD), alone?
# And the cheapest possible rehearsal of an overlap: TWO contexts on the one GPU (own streams, own half of the pixel tiles, nothing
# orders them against each other), each with a persistent grid of 8 / 6 / 4 workgroups per CU.
# usage: bash profiles/r6_overlap/step1.sh [scenes] [reps]      (one gpurun call; plain bench runs, HIP-event times, no profiler)
SCENES=${1:-"c4 c5 s4 c3"}; REPS=${2:-2}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r6; mkdir -p $OUT
{
for sc in $SCENES; do
  case $sc in
    c4) ARGS="" ;;
    c3) ARGS="--scene c3 --width 1024 --height 1024 --vthreads 16" ;;
    c5) ARGS="--scene c5 --width 4096 --height 4096 --vthreads 8" ;;
    s4) ARGS="--scene s4" ;;
  esac
  for rep in $(seq 1 $REPS); do
    for v in 1x8 1x7 1x6 1x5 2x8 2x6 2x4 2x5; do
      ctx=${v%x*}; bpc=${v#*x}
      C=""; [ "$ctx" -gt 1 ] && C="--contexts $ctx"
      timeout -k 10 240 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-counters $ARGS $C --opt blocks_per_cu=$bpc --opt shade_per=4 > $OUT/s1_${sc}_${v}_$rep.json 2> $OUT/s1_${sc}_${v}_$rep.err
      python3 - $OUT/s1_${sc}_${v}_$rep.json $sc $v $rep <<PY
import json,sys
f,sc,v,rep=sys.argv[1:5]
try:
    L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
    st=L.get("stages") or {}
    print("%s %s rep %s: %.1f Mrays/s, %.1f ms/step, trace %.1f ms/step, shade %s ms/batch, fold %s" % (sc, v, rep, L["value"], L["ms_per_step"], L["trace_ms_per_step"],
          (st.get("shade") or {}).get("ms_per_batch"), (st.get("fold_accumulate") or {}).get("ms_per_batch")), flush=True)
except Exception as e:
    print(sc, v, rep, "FAILED", e, flush=True)
PY
    done
  done
done
} 2>&1 | tee $OUT/step1.txt

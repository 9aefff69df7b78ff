#!/bin/bash
# Round 6, review item 4, step 9.  probe8: asked for as physically CONTIGUOUS memory the path state makes the stage slower than either mode
# (C3 15.0 ms per batch against 12.1 / 10.4; C4 38.7 against 33.9) -- every time: six of six processes.  So the physical layout IS the
# variable, and the more regular it is the worse: in contiguous memory the fifteen fields of a bank sit at physical offsets that are
# multiples of the stride (C3: 2^28 bytes), and the memory's channel / bank selection sees the same bits for all of them.  In contiguous
# memory physical offsets are ours to choose: sweep the pad between the fields (option hot_pad, items) -- deterministic now.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
SC=${1:-c3}; PADS=${2:-"0 64 1088 4160 16448 66624 279616 541760 1118272"}; CONT=${3:-1}
case $SC in
  c3) A="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1" ;;
  c4) A="--steps 2 --warmup 1" ;;
  c5) A="--scene c5 --width 4096 --height 4096 --vthreads 8 --steps 4 --warmup 1" ;;
  s4) A="--scene s4 --steps 2 --warmup 1" ;;
esac
ARGS="$A --no-cpu --no-counters --opt shade_per=4"
{
for pad in $PADS; do
    python3 $R/bench.py $ARGS --opt paths_contiguous=$CONT --opt hot_pad=$pad > $OUT/bm9_${SC}_$pad.json 2> $OUT/bm9_${SC}_$pad.err
    python3 - $OUT/bm9_${SC}_$pad.json $SC $pad $CONT <<PY
import json,sys
f,sc,pad,cont=sys.argv[1:5]
try:
    L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
    st=L["stages"]
    print("%s contiguous %s hot_pad %8s items (%9d B): shade %.3f ms/batch  fold %.3f  raygen %.3f  trace %.1f ms/step  %.1f Mrays/s  w/t %.4f" % (sc, cont, pad, 4*int(pad), st["shade"]["ms_per_batch"], st["fold_accumulate"]["ms_per_batch"],
          st["raygen"]["ms_per_batch"], L["trace_ms_per_step"], L["value"], L["trace_ms_per_step"]/L["ms_per_step"]), flush=True)
except Exception as x:
    print(sc, pad, "FAILED", x, flush=True)
PY
done
} 2>&1 | tee -a $OUT/bimodal_probe9.txt

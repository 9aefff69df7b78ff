#!/bin/bash
# Round 6, review item 4, step 12: which spread backing, if any, is fast EVERY time?  Chunk size x holes on C3, three processes each, plus
# plain hipMalloc as the box's reference.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
ARGS="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1 --no-cpu --no-counters --opt shade_per=4"
{
for i in 1 2 3; do
  for v in 0:1 16:0 16:1 64:0 64:1 256:0 1024:0 1024:1 4096:0 4096:1; do
    sp=${v%:*}; h=${v#*:}
    python3 $R/bench.py $ARGS --opt paths_spread=$sp --opt paths_spread_holes=$h > $OUT/bm12.json 2> $OUT/bm12.err
    python3 - $OUT/bm12.json $sp $h $i <<PY
import json,sys
f,sp,h,i=sys.argv[1:5]
try:
    L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
    print("c3 chunk %5s MB holes %s run %s: shade %.3f ms/batch  raygen %.3f  %.1f Mrays/s" % (sp, h, i, L["stages"]["shade"]["ms_per_batch"], L["stages"]["raygen"]["ms_per_batch"], L["value"]), flush=True)
except Exception as x:
    print(sp, h, i, "FAILED", x, flush=True)
PY
  done
done
} 2>&1 | tee $OUT/bimodal_probe12.txt

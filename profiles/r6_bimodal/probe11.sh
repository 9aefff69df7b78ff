#!/bin/bash
# Round 6, review item 4, step 11.  probe10: re-allocated into a deliberately fragmented device memory (holes of 2 GB / 256 MB between blocks
# that stay) the path state gives the FAST stage every time, and the mode flips inside one process when the path state moves to other
# physical memory.  The same layout without seizing the device: option paths_spread = chunk size in MB (art_api.cpp alloc_spread: one address
# range over separately created physical chunks, a spacer chunk created between two of them and released at the end).  N processes per setting.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
N=${1:-4}; SC=${2:-c3}; SETS=${3:-"0 2048 256"}; EXTRA=${4:-"--opt shade_per=4"}
case $SC in
  c3) A="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1" ;;
  c4) A="--steps 2 --warmup 1" ;;
  c5) A="--scene c5 --width 4096 --height 4096 --vthreads 8 --steps 4 --warmup 1" ;;
  s4) A="--scene s4 --steps 2 --warmup 1" ;;
esac
ARGS="$A --no-cpu --no-counters $EXTRA"
{
for i in $(seq 1 $N); do
  for sp in $SETS; do
    t0=$(date +%s.%N)
    ART_DEBUG_ADDR=1 python3 $R/bench.py $ARGS --opt paths_spread=$sp > $OUT/bm11_${SC}_${sp}_$i.json 2> $OUT/bm11_${SC}_${sp}_$i.err
    t1=$(date +%s.%N)
    python3 - $OUT/bm11_${SC}_${sp}_$i.json $OUT/bm11_${SC}_${sp}_$i.err $SC $sp $i $t0 $t1 <<PY
import json,sys,re
f,e,sc,sp,i,t0,t1=sys.argv[1:8]
try:
    L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
    a=[l for l in open(e) if l.startswith("ART_DEBUG_ADDR")]
    got=re.search(r"spread (\d)", [x for x in a if "paths" in x][-1]).group(1) if a else "?"
    al=[x for x in a if "alloc_spread" in x]
    if al: got += " (" + al[-1].split(":")[-1].strip() + ")"
    st=L["stages"]
    print("%s paths_spread %5s MB (spread: %s) run %s: shade %.3f ms/batch  fold %.3f  raygen %.3f  trace %.1f ms/step  %.1f Mrays/s  w/t %.4f  upload+alloc+render wall %.1f s" % (sc, sp, got, i, st["shade"]["ms_per_batch"], st["fold_accumulate"]["ms_per_batch"],
          st["raygen"]["ms_per_batch"], L["trace_ms_per_step"], L["value"], L["trace_ms_per_step"]/L["ms_per_step"], float(t1)-float(t0)), flush=True)
except Exception as x:
    print(sc, sp, i, "FAILED", x, open(e).read()[-400:], flush=True)
PY
  done
done
} 2>&1 | tee -a $OUT/bimodal_probe11.txt

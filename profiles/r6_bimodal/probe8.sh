#!/bin/bash
# Round 6, review item 4, step 8.  probe7: a fast and a slow process issue the same instructions, hit and miss L2 equally often and send the
# same requests to the fabric -- but the slow one waits 16 % longer for its L1 -> L2 reads (958 against 826 cycles) and stalls 28 % more
# on pending misses: the memory behind L2 answers more slowly.  Same requests, slower answers, following the process, alternating: the
# physical pages the driver found for the 35 GB of path state -- a contiguous range in one process, scattered blocks in the next (the
# previous process' memory still being cleared), i.e. DRAM row locality of 28 concurrent streams.  Test: the path state asked for as
# physically contiguous memory (option paths_contiguous = 1) against hipMalloc, alternately, N processes each.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
N=${1:-6}; SC=${2:-c3}
case $SC in
  c3) A="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1" ;;
  c4) A="--steps 2 --warmup 1" ;;
  c5) A="--scene c5 --width 4096 --height 4096 --vthreads 8 --steps 4 --warmup 1" ;;
  s4) A="--scene s4 --steps 2 --warmup 1" ;;
esac
ARGS="$A --no-cpu --no-counters --opt shade_per=4"
{
for i in $(seq 1 $N); do
  for pc in 0 1; do
    t0=$(date +%s.%N)
    ART_DEBUG_ADDR=1 python3 $R/bench.py $ARGS --opt paths_contiguous=$pc > $OUT/bm8_${SC}_${pc}_$i.json 2> $OUT/bm8_${SC}_${pc}_$i.err
    t1=$(date +%s.%N)
    python3 - $OUT/bm8_${SC}_${pc}_$i.json $OUT/bm8_${SC}_${pc}_$i.err $SC $pc $i $t0 $t1 <<PY
import json,sys,re
f,e,sc,pc,i,t0,t1=sys.argv[1:8]
try:
    L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
    a=[l for l in open(e) if l.startswith("ART_DEBUG_ADDR")]
    got=re.search(r"contiguous (\d)", a[-1]).group(1) if a else "?"
    st=L["stages"]
    print("%s asked %s got %s run %s: shade %.3f ms/batch  fold %.3f  raygen %.3f  trace %.1f ms/step  %.1f Mrays/s  w/t %.4f  process wall %.1f s" % (sc, pc, got, i, st["shade"]["ms_per_batch"], st["fold_accumulate"]["ms_per_batch"],
          st["raygen"]["ms_per_batch"], L["trace_ms_per_step"], L["value"], L["trace_ms_per_step"]/L["ms_per_step"], float(t1)-float(t0)), flush=True)
except Exception as x:
    print(sc, pc, i, "FAILED", x, open(e).read()[-300:], flush=True)
PY
  done
done
} 2>&1 | tee -a $OUT/bimodal_probe8.txt

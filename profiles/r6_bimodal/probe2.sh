#!/bin/bash
# Round 6, review item 4, step 2.  probe1 (one process, the path state freed and allocated again six times, four frame sizes): ONE mode
# throughout, whatever the stride -- the mode belongs to the PROCESS.  What a process draws at random is its address-space layout (ROCm
# hands out device addresses from the process' own mmap region: 0x7435... above).  So: the same bench line from N processes with the
# kernel's address randomisation on, then off (setarch -R: every process gets the same addresses), the path state's address next to the
# stage's time.  One run per process; nothing is repeated to provoke a mode.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
N=${1:-8}; M=${2:-5}
ARGS="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1 --no-cpu --no-counters --opt shade_per=4"
one() {   # $1 tag, rest: launcher prefix
  tag=$1; shift
  ART_DEBUG_ADDR=1 "$@" python3 $R/bench.py $ARGS > $OUT/bm_$tag.json 2> $OUT/bm_$tag.err
  python3 - $OUT/bm_$tag.json $OUT/bm_$tag.err $tag <<PY
import json,sys,re
f,e,tag=sys.argv[1:4]
try:
    L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
    a=[l for l in open(e) if l.startswith("ART_DEBUG_ADDR")]
    m=re.search(r"paths (\S+) .* hot0 (\S+) hot1 (\S+) .* cold (\S+) live \S+ counters (\S+)", a[-1]) if a else None
    st=L["stages"]
    print("%s: shade %.3f ms/batch  fold %.3f  raygen %.3f  trace %.3f ms/launch  %.1f Mrays/s | paths %s hot0 %s counters %s" % (tag, st["shade"]["ms_per_batch"], st["fold_accumulate"]["ms_per_batch"],
          st["raygen"]["ms_per_batch"], L["roofline"]["avg_launch_ms"] if L.get("roofline") else L["trace_ms_per_step"]/9.0, L["value"], m.group(1) if m else "?", m.group(2) if m else "?", m.group(5) if m else "?"), flush=True)
except Exception as x:
    print(tag, "FAILED", x, flush=True)
PY
}
{
for i in $(seq 1 $N); do one aslr_$i env; done
if setarch x86_64 -R true 2>/dev/null; then
  for i in $(seq 1 $M); do one noaslr_$i setarch x86_64 -R; done
else echo "setarch -R refused on this box"; fi
} 2>&1 | tee $OUT/bimodal_probe2.txt

#!/bin/bash
# Round 6, review item 4, step 3.  probe2: with address randomisation OFF every process gets the same device addresses (paths
# 0x7feeb1000000 ...) and the stage is STILL fast in three processes and slow in two -- virtual addresses are out.  In the fast processes the
# VALU-bound trace kernel is consistently 0.5 % SLOWER (11.91 against 11.85 ms per launch) while the memory-bound stage is 16 % faster:
# the signature of a clock / power state that favours the memory side.  So: the same bench line from N processes, 40 steps each (5 s of
# GPU work), rocm-smi's clocks and power sampled beside it.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
N=${1:-8}
ARGS="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 40 --warmup 2 --no-cpu --no-counters --opt shade_per=4"
{
rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40
for i in $(seq 1 $N); do
  python3 $R/bench.py $ARGS > $OUT/bm3_$i.json 2> $OUT/bm3_$i.err &
  pid=$!
  : > $OUT/bm3_$i.smi
  while kill -0 $pid 2>/dev/null; do
    rocm-smi --showclocks --showpower --json >> $OUT/bm3_$i.smi 2>/dev/null; echo >> $OUT/bm3_$i.smi
    sleep 0.2
  done
  wait $pid
  python3 - $OUT/bm3_$i.json $OUT/bm3_$i.smi $i <<PY
import json,sys,re
f,smi,i=sys.argv[1:4]
L=[json.loads(l) for l in open(f) if l.startswith('{"metric"')][-1]
S=[]
for l in open(smi):
    l=l.strip()
    if not l.startswith("{"): continue
    try: S.append(json.loads(l))
    except Exception: pass
def num(v):
    m=re.search(r"([0-9.]+)", str(v)); return float(m.group(1)) if m else None
keys={}
for s in S:
    c=s.get("card0", {})
    for k,v in c.items():
        x=num(v)
        if x is not None: keys.setdefault(k,[]).append(x)
# the samples of the busy phase: the upper half by power
busy=sorted(range(len(S)), key=lambda j: -(keys.get("Current Socket Graphics Package Power (W)", keys.get("Average Graphics Package Power (W)",[0]*len(S)))[j] if j < len(keys.get("Current Socket Graphics Package Power (W)", keys.get("Average Graphics Package Power (W)",[]))) else 0))[:max(1,len(S)//3)]
def mean(k):
    v=keys.get(k,[]); w=[v[j] for j in busy if j < len(v)]
    return round(sum(w)/len(w),1) if w else None
st=L["stages"]
print("run %s: shade %.3f ms/batch trace %.3f ms/launch %.1f Mrays/s | %d smi samples, busy third: %s" % (i, st["shade"]["ms_per_batch"], L["trace_ms_per_step"]/9.0, L["value"], len(S),
      ", ".join("%s=%s" % (k.replace(" clock speed:","").replace("(","").replace(")",""), mean(k)) for k in sorted(keys) if ("clk" in k.lower() or "clock" in k.lower() or "Power" in k))), flush=True)
PY
done
} 2>&1 | tee $OUT/bimodal_probe3.txt

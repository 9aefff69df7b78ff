#!/bin/bash
# Round 6, review item 4, step 5.  probe4: inside one process the mode does not move when 1 ... 100 GB of dummy allocations are held while
# the path state is allocated -- physical placement of the DATA is out as well.  What a process also places once is its CODE: the stage
# kernel is ~55 KB of instructions per instantiation (two are used per pass) against a 64 KB instruction cache shared by two CUs, the trace kernel
# 4 KB.  So: N processes under rocprofv3, kernel durations and the instruction-cache / wait counters of the SAME process side by side.
# One counter pass per process (a process IS a mode); no trace domain besides --kernel-trace.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
N=${1:-8}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|SQC_|IFETCH|INST_CACHE" | head -40 > $OUT/bm5_counters_available.txt
ARGS="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1 --no-cpu --no-counters --opt shade_per=4"
PMC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES"
{
cat $OUT/bm5_counters_available.txt | cut -c1-160
for i in $(seq 1 $N); do
  d=$OUT/bm5_$i; rm -rf $d; mkdir -p $d
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $d/p -- python3 $R/bench.py $ARGS > $d/log 2>&1
  python3 - $d $i <<PY
import csv,glob,sys,collections
d,i=sys.argv[1:3]
try:
    kt=glob.glob(d+"/p/*/*kernel_trace.csv")[0]; cc=glob.glob(d+"/p/*/*counter_collection.csv")[0]
except Exception as e:
    print("run", i, "no output", e, open(d+"/log").read()[-400:]); sys.exit(0)
dur=collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n=r["Kernel_Name"].split("(")[0].replace("void art::","")
    dur[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
cnt=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cc)):
    n=r["Kernel_Name"].split("(")[0].replace("void art::","")
    cnt[n][r["Counter_Name"]]+=float(r["Counter_Value"])
for n in sorted(dur):
    if not n.startswith("k_shade_compact<4, false") and not n.startswith("k_trace_coop"): continue
    c=cnt[n]; L=len(dur[n])
    print("run %s %-34s x%-3d avg %.3f ms | %s" % (i, n[:34], L, sum(dur[n])/L, "  ".join("%s=%.4g" % (k.replace("SQC_","").replace("SQ_",""), v/L) for k,v in sorted(c.items()))), flush=True)
PY
  rm -rf $d/p
done
} 2>&1 | tee $OUT/bimodal_probe5.txt

#!/bin/bash
# Round 6, review item 4, step 7.  Streams / hardware queues inside one process change nothing (probe6).  Boxes differ: on some every
# process is slow (probe1, 3, 6), on one every process was fast (r6_shade/ab1: 16 of 16), on some the modes alternate process by process
# (probe2, probe5).  If this box alternates: a wider counter set of a fast and a slow process side by side -- waves and their placement
# (SQ per XCC if the tool reports the dimension), LDS conflicts, L2 / fabric requests, vector-L1 stalls.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r6; mkdir -p $OUT
N=${1:-6}
cd /tmp && export TMPDIR=/tmp
ARGS="--scene c3 --width 1024 --height 1024 --vthreads 16 --steps 3 --warmup 1 --no-cpu --no-counters --opt shade_per=4"
run_set() {  # $1 name, rest counters
  name=$1; shift
  for i in $(seq 1 $N); do
    d=$OUT/bm7_${name}_$i; rm -rf $d; mkdir -p $d
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d/p -- python3 $R/bench.py $ARGS > $d/log 2>&1
    python3 - $d $name $i <<PY
import csv,glob,sys,collections
d,name,i=sys.argv[1:4]
try:
    kt=glob.glob(d+"/p/*/*kernel_trace.csv")[0]; cc=glob.glob(d+"/p/*/*counter_collection.csv")[0]
except Exception as e:
    print(name, "run", i, "no output:", open(d+"/log").read()[-300:].replace("\n"," | ")); sys.exit(0)
dur=collections.defaultdict(list)
for r in csv.DictReader(open(kt)):
    n=r["Kernel_Name"].split("(")[0].replace("void art::","")
    dur[n].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
rows=list(csv.DictReader(open(cc)))
if i=="1": print(name, "counter csv columns:", list(rows[0].keys()))
cnt=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void art::","")
    cnt[n][r["Counter_Name"]]+=float(r["Counter_Value"])
for n in sorted(dur):
    if not n.startswith("k_shade_compact<4, false") and not n.startswith("k_trace_coop") and not n.startswith("k_fold_level"): continue
    c=cnt[n]; L=len(dur[n])
    print("%s run %s %-30s x%-3d avg %.3f ms | %s" % (name, i, n[:30], L, sum(dur[n])/L, "  ".join("%s=%.4g" % (k.replace("_sum",""), v/L) for k,v in sorted(c.items()))), flush=True)
PY
    [ "$i" -gt 2 ] && rm -rf $d/p
  done
}
{
run_set sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
run_set mem TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
} 2>&1 | tee $OUT/bimodal_probe7.txt

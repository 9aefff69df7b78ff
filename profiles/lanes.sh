#!/bin/bash
# VALU lane utilisation per kernel: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = mean active lanes per VALU instruction (of 64).
#   usage: bash profiles/lanes.sh <name> [bench.py workload options]   -> gpurun_out/lanes_<name>.txt
NAME=${1:-c4}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/lanes_$NAME
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-counters "$@" > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
python3 - $OUT <<'PY' | tee $OUT.txt
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1]+"/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ","").replace("art::","")[:28]][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_ACTIVE_INST_VALU",0))[:6]:
    a=v.get("SQ_ACTIVE_INST_VALU",0); t=v.get("SQ_THREAD_CYCLES_VALU",0)
    print("%-28s lanes/VALU inst %.1f  insts %.3g active %.3g thread_cycles %.3g busy %.3g gui %.3g waves %.3g" % (k, t/a if a else 0, v.get("SQ_INSTS_VALU",0), a, t, v.get("SQ_BUSY_CYCLES",0), v.get("GRBM_GUI_ACTIVE",0), v.get("SQ_WAVES",0)))
PY

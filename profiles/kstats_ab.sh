#!/bin/bash
# per-kernel average durations (rocprofv3 --kernel-trace --stats) of build variants on the same box.  usage: bash profiles/kstats_ab.sh "<v1> <v2>" [bench options]
VARS=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  lib=$R/ada-ray-tracer_amd/libart_hip.so; [ $v != base ] && lib=$R/ada-ray-tracer_amd/libart_hip_$v.so
  rm -rf $R/gpurun_out/ks_$v; mkdir -p $R/gpurun_out/ks_$v
  ART_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$v/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-counters "$@" > $R/gpurun_out/ks_$v/log 2>&1
  python3 - $R/gpurun_out/ks_$v $v <<PY
import csv,glob,sys,json
d,v=sys.argv[1],sys.argv[2]
f=glob.glob(d+"/trace/*/*kernel_stats.csv")[0]
val=[json.loads(l)["value"] for l in open(d+"/log") if l.startswith('{"metric"')][-1]
print(v, "whole %.1f" % val, " | ".join("%s %.3f ms x%s" % (r["Name"].split("(")[0].replace("art::","").replace("void ","")[:18], float(r["AverageNs"])/1e6, r["Calls"]) for r in list(csv.DictReader(open(f)))[:5]))
PY
done

#!/bin/bash
# A/B of build variants of the same source on the GPU box (through gpurun from the repo root).
#   make -C ada-ray-tracer_amd OUT=libart_hip_<v>.so BUILD=build_<v> EXTRA="-D..." libart_hip_<v>.so      (variants; "base" = libart_hip.so)
# usage: bash profiles/ab.sh "<v1> <v2> ..." <scene> [bench options]   -> gpurun_out/ab_<scene>_<variant>.json
VARS=$1; SC=${2:-c4}; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for v in $VARS; do
  lib=$R/ada-ray-tracer_amd/libart_hip.so; [ $v != base ] && lib=$R/ada-ray-tracer_amd/libart_hip_$v.so
  [ -f $lib ] || { echo "no $lib"; continue; }
  ART_LIB=$lib timeout -k 10 200 python3 $R/bench.py --scene $SC --steps 2 --warmup 1 --no-cpu "$@" > $R/gpurun_out/ab_${SC}_$v.json 2>>$R/gpurun_out/ab.err || { echo "FAILED $v"; exit 1; }
  python3 - $R/gpurun_out/ab_${SC}_$v.json $SC $v <<PY
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("%s %-8s whole %8.1f Mrays/s  trace launch %8.3f ms  trace-only %8.1f  NV %.2f" % (sys.argv[2], sys.argv[3], d["value"], r["avg_launch_ms"], r["trace_Mrays_per_s"], r["node_visits_per_ray"]))
PY
done

#!/bin/bash
# Sensitivity of k_trace_coop to extra work per node step, on the GPU box (through gpurun from the repo root).
# Diagnostic builds of the same source: make OUT=libart_hip_<v>.so BUILD=build_<v> EXTRA=-DART_DIAG_VALU=16 | -DART_DIAG_LOAD=1|2
# usage: bash profiles/diag.sh <scene> [bench options]   -> gpurun_out/diag_<scene>_<variant>.json
SC=${1:-c4}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for v in base valu16 load1 load2; do
  lib=$R/ada-ray-tracer_amd/libart_hip.so; [ $v != base ] && lib=$R/ada-ray-tracer_amd/libart_hip_$v.so
  [ -f $lib ] || continue
  ART_LIB=$lib timeout -k 10 200 python3 $R/bench.py --scene $SC --steps 2 --warmup 1 --no-cpu "$@" > $R/gpurun_out/diag_${SC}_$v.json 2>>$R/gpurun_out/diag.err || { echo "FAILED $v"; exit 1; }
  python3 - $R/gpurun_out/diag_${SC}_$v.json $SC $v <<PY
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; print("%s %-8s whole %8.1f Mrays/s  trace launch %8.3f ms  trace-only %8.1f" % (sys.argv[2], sys.argv[3], d["value"], r["avg_launch_ms"], r["trace_Mrays_per_s"]))
PY
done

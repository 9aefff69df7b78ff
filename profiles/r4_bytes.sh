#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of one kernel for library variants (two PMC passes each).  usage: bash profiles/r4_bytes.sh "<variants>" <kernel substring> [bench options]
VARS=$1; K=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for V in $VARS; do
  lib=$R/ada-ray-tracer_amd/libart_hip.so; [ $V != base ] && lib=$R/ada-ray-tracer_amd/libart_hip_$V.so
  OUT=$R/gpurun_out/r4/bytes_$V; rm -rf $OUT; mkdir -p $OUT
  B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-counters $*"
  ART_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/i -- $B > $OUT/i.log 2>&1 || exit 1
  ART_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/j -- $B > $OUT/j.log 2>&1 || exit 1
  python3 - $OUT "$K" $V <<'PY'
import csv,glob,sys,collections
def per_dispatch(d,counter,kern):
    out={}
    for f in glob.glob(d+"/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and r["Counter_Name"]==counter:
                out[int(r["Dispatch_Id"])]=out.get(int(r["Dispatch_Id"]),0)+float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]
F=per_dispatch(sys.argv[1]+"/i","FETCH_SIZE",sys.argv[2]); W=per_dispatch(sys.argv[1]+"/j","WRITE_SIZE",sys.argv[2])
print(sys.argv[3], sys.argv[2], "fetch GB", [round(x*2*1024/1e9,2) for x in F[:9]], "write GB", [round(x*1024/1e9,2) for x in W[:9]])
PY
done

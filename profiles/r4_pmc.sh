#!/bin/bash
# PMC passes for the wavefront stages (not only the trace kernel) of one library variant; prints per-launch averages for one kernel.
#   usage: bash profiles/r4_pmc.sh <variant|base> <kernel substring> [bench.py options]   -> gpurun_out/r4/pmc_<variant>_<kernel>.txt
V=$1; K=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
lib=$R/ada-ray-tracer_amd/libart_hip.so; [ $V != base ] && lib=$R/ada-ray-tracer_amd/libart_hip_$V.so
OUT=$R/gpurun_out/r4/pmc_$V; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu --no-counters $*"
pass() { n=$1; shift; ART_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -- $B > $OUT/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $OUT/$n.log; return 1; }; }
pass a GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU || exit 1
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM || exit 1
pass c SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU || exit 1
pass d GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max || exit 1
pass e TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || exit 1
pass f TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum || exit 1
pass g TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum || exit 1
pass h TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
pass i FETCH_SIZE || exit 1
pass j WRITE_SIZE || exit 1
pass k TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum || true
python3 - $OUT "$K" <<'PY' | tee $OUT/../pmc_${V}_$K.txt
import csv,glob,sys,collections
agg=collections.defaultdict(float); n=collections.Counter(); dur=[0.0,0]
for d in sorted(glob.glob(sys.argv[1]+"/*/")):
    disp=set()
    for f in glob.glob(d+"*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if sys.argv[2] in r["Kernel_Name"]:
                agg[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    for f in glob.glob(d+"*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if sys.argv[2] in r["Kernel_Name"] and d.rstrip("/").endswith("/a"):
                dur[0]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"])); dur[1]+=1
    for c in list(agg):
        if c not in n: n[c]=len(disp)
print("kernel", sys.argv[2], "launches", dur[1], "avg ms under pmc", dur[0]/max(1,dur[1])/1e6)
for c in sorted(agg): print("  %-36s %14.4g per launch" % (c, agg[c]/max(1,n[c])))
PY

R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_q1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-counters"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $B > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcc -- $B > $OUT/tcc.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq -- $B > $OUT/sq.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for name in ("fetch","tcc","sq"):
    f = sorted(glob.glob("$OUT/%s/*/*_counter_collection.csv" % name))[-1]
    agg = collections.defaultdict(float); disp = {}
    for r in csv.DictReader(open(f)):
        if "k_trace_coop" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    n = len(disp); ns = sum(disp.values())
    print(name, "dispatches", n, "total_ms", ns/1e6, dict(agg))
    if name == "fetch": print("  fetch GB/s raw", agg["FETCH_SIZE"]*1024/ns, " x2:", 2*agg["FETCH_SIZE"]*1024/ns)
    if name == "tcc": print("  L2 hit", agg["TCC_HIT_sum"]/(agg["TCC_HIT_sum"]+agg["TCC_MISS_sum"]))
    if name == "sq": print("  VALU per SIMD-cycle: insts", agg["SQ_INSTS_VALU"], "cycles(GRBM/8)", agg["GRBM_GUI_ACTIVE"]/8, "-> cycles per VALU inst per SIMD", (agg["GRBM_GUI_ACTIVE"]/8)*1024/max(1,agg["SQ_INSTS_VALU"]), "active_valu/wave_cycles", agg["SQ_ACTIVE_INST_VALU"]/agg["SQ_WAVE_CYCLES"], "wait_any", agg["SQ_WAIT_ANY"]/agg["SQ_WAVE_CYCLES"], "wait_inst_any", agg["SQ_WAIT_INST_ANY"]/agg["SQ_WAVE_CYCLES"])
PY

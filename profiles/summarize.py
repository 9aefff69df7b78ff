#!/usr/bin/env python3
"""Turns gpurun_out/prof_final (raw rocprofv3 csv, scratch) into the committed summaries under profiles/<tag>/:
kernel_stats.csv (rocprofv3 --kernel-trace --stats) and pmc_summary.json (per-kernel counter sums, per-launch HBM traffic
of the trace kernel).  usage: python profiles/summarize.py r2_final c4"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: take the most recent run of a pass"""
    return max(glob.glob(pattern), key=os.path.getmtime)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# FETCH_SIZE -> bytes, per access shape (round 5, profiles/calib_shapes.hip -> profiles/r5_calib/calib_shapes.json, tables >> 256 MiB with a
# known byte count):  random 64-byte packets read by 4 lanes x 16 B (the node visit) x0.998,  random 64-byte records read by one lane
# (triangle test, shading record) x0.998 / x1.001,  coalesced 4-B- and 16-B-per-lane streams x2.000,  stores x1.000 (WRITE_SIZE is exact).
# Rounds 1-4 doubled FETCH_SIZE for every kernel on the strength of ONE calibration of 128-byte segment gathers (round 1's 8-wide node).
#   k_trace_coop      its fetches are node packets and triangle records (3968 B per ray on C4 against 64 B of streamed trace record): x1
#   stream kernels    k_fold_level, k_accumulate, k_resolve_last, k_resolve: x2
#   k_shade_compact   a mix: the item's own words are streams (x2), the triangle's shading record is a gather (x1).  With the item counts of
#                     the bench line taken under rocprof (stages.items_in_per_bounce) the streamed part is known: true = raw + streamed / 2;
#                     without them the factor of the synthetic mix kernel (k_stage_mix<64>: 1.417) is used.
FETCH_FACTOR_TRACE = 1.0
FETCH_FACTOR_STREAM = 2.0
FETCH_FACTOR_STAGE_MIX = 1.417
STAGE_STREAM_IN_B, STAGE_STREAM_IN_B0 = 60.0, 16.0          # bench.py STAGE_IN_B / STAGE_IN_B0: the streamed words an input item is read with


def main(tag, name="c4"):
    """tag: directory under profiles/ (r3_final for the bench.py default workload C4, which bench.py quotes; r3_c3 ...);
    name: the collect.sh run to read (gpurun_out/prof_<name>)"""
    global SRC
    SRC = os.path.join(ROOT, "gpurun_out", "prof_" + name)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    shutil.copyfile(newest(os.path.join(SRC, "trace", "*", "*_kernel_stats.csv")), os.path.join(dst, "kernel_stats.csv"))
    out = {}
    for name in ("fetch", "write", "sq", "tcc", "ta", "ta2", "tcp", "tcp2", "tcp3", "sq2"):
        try:
            rows = list(csv.DictReader(open(newest(os.path.join(SRC, name, "*", "*_counter_collection.csv")))))
        except ValueError:
            continue                                              # a pass this collection did not take
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(dict)
        for r in rows:
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k][r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, v in agg.items():
            o = out.setdefault(k, {})
            o.update(v)
            o["dispatches_" + name] = len(disp[k]); o["total_ns_" + name] = sum(disp[k].values())
    for k, o in out.items():
        if "k_trace_coop" in k and "FETCH_SIZE" in o:
            n = o["dispatches_fetch"]
            fetch_b = o["FETCH_SIZE"] * 1024.0 / n; write_b = o["WRITE_SIZE"] * 1024.0 / o["dispatches_write"]
            o["hbm_bytes_per_launch_raw"] = fetch_b + write_b                       # FETCH_SIZE/WRITE_SIZE are in KiB
            o["hbm_bytes_per_launch_fetch_x2"] = 2.0 * fetch_b + write_b            # gfx950: FETCH_SIZE may count 128-B requests as 64 B
            o["avg_launch_ms_fetch_pass"] = o["total_ns_fetch"] / n / 1e6
            o["traffic_GBps_raw"] = o["hbm_bytes_per_launch_raw"] / (o["total_ns_fetch"] / n) 
            o["traffic_GBps_fetch_x2"] = o["hbm_bytes_per_launch_fetch_x2"] / (o["total_ns_fetch"] / n)
            # round 5: the calibrated figure (random 64-byte packets: FETCH_SIZE is exact) -- what bench.py quotes as roofline.traffic
            o["fetch_factor_calibrated"] = FETCH_FACTOR_TRACE
            o["hbm_bytes_per_launch_calibrated"] = FETCH_FACTOR_TRACE * fetch_b + write_b
            o["traffic_GBps_calibrated"] = o["hbm_bytes_per_launch_calibrated"] / (o["total_ns_fetch"] / n)
            if "GRBM_GUI_ACTIVE" in o:
                o["effective_clock_GHz"] = o["GRBM_GUI_ACTIVE"] / 8.0 / o["total_ns_sq"]
            if "TCC_HIT_sum" in o:
                o["l2_hit_rate"] = o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"])
            # utilisation of the units the kernel could be limited by (collect_util.sh).  GRBM_GUI_ACTIVE / 8 = busy cycles of one XCD summed
            # over the launches; SQ_* are in quad-cycles summed over the waves (8 per SIMD); TA / TCP: one instance per CU (256).
            if "GRBM_GUI_ACTIVE" in o and "TA_BUSY_avr" in o:
                cyc = o["GRBM_GUI_ACTIVE"] / 8.0
                u = {"ta_busy_avg": o["TA_BUSY_avr"] / cyc, "ta_busy_max_instance": o["TA_BUSY_max"] / cyc}
                if "TCP_PENDING_STALL_CYCLES_sum" in o: u["tcp_pending_stall"] = o["TCP_PENDING_STALL_CYCLES_sum"] / 256.0 / cyc
                if "TCP_TCC_READ_REQ_sum" in o: u["l1_to_l2_read_latency_cycles"] = o["TCP_TCC_READ_REQ_LATENCY_sum"] / o["TCP_TCC_READ_REQ_sum"]
                if "SQ_WAVE_CYCLES" in o:
                    u["valu_issue_slots_if_4_cycles_each"] = 8.0 * o["SQ_ACTIVE_INST_VALU"] / o["SQ_WAVE_CYCLES"]
                    if "SQ_ACTIVE_INST_LDS" in o: u["lds_active"] = 8.0 * o["SQ_ACTIVE_INST_LDS"] / o["SQ_WAVE_CYCLES"]
                if "SQ_INSTS_VMEM_RD" in o:
                    u["vmem_read_insts_per_launch"] = o["SQ_INSTS_VMEM_RD"] / o["dispatches_sq2"]; u["lds_insts_per_launch"] = o["SQ_INSTS_LDS"] / o["dispatches_sq2"]
                o["utilisation"] = u
            # issue-boundness (profiles/valu_mix.py): VALU instructions per launch and SIMD x their mean issue time / launch time
            mix_path = os.path.join(dst, "valu_mix.json")
            if os.path.exists(mix_path) and "SQ_INSTS_VALU" in o:
                mix = json.load(open(mix_path))
                per_simd = o["SQ_INSTS_VALU"] / o["dispatches_sq"] / 1024.0
                launch_ns = o["total_ns_sq"] / o["dispatches_sq"]
                o["valu_mean_issue_ns"] = mix["mean_issue_ns_per_valu_instruction"]
                o["valu_instructions_per_ns_and_simd"] = per_simd / launch_ns
                o["valu_issue_frac"] = per_simd * mix["mean_issue_ns_per_valu_instruction"] / launch_ns
                o["valu_issue_formula"] = "SQ_INSTS_VALU / dispatches / 1024 SIMDs x mean issue ns of the kernel's instruction mix (valu_mix.json x valu_rate2.hip) / launch ns"
    # the other kernels of a step: bytes past L2 and achieved TB/s (the stages are memory-bound; the HBM roofline is theirs)
    others = {}
    bench_line = None
    try:
        bench_line = json.load(open(os.path.join(SRC, "bench_under_rocprof.json")))
    except Exception:
        pass
    for k, o in out.items():
        if "k_trace_coop" in k or "FETCH_SIZE" not in o or "dispatches_write" not in o or "art::" not in k or "anonymous" in k or "rocprim" in k or k.replace("void ", "").replace("art::", "").strip() == "":
            continue
        n = o["dispatches_fetch"]
        raw = o["FETCH_SIZE"] * 1024.0 / n; wb = o["WRITE_SIZE"] * 1024.0 / o["dispatches_write"]; ns = o["total_ns_fetch"] / n
        how = "streams: x2"
        fb = FETCH_FACTOR_STREAM * raw
        if "k_shade_compact" in k:
            fb = FETCH_FACTOR_STAGE_MIX * raw; how = "synthetic mix factor 1.417 (k_stage_mix<64>)"
            st = (bench_line or {}).get("stages")
            if st:                                                # the streamed words of the launches this kernel name covers, from the item counts
                ins = st["items_in_per_bounce"]; nb = st["batches"]
                camera = "true" in k.split("<")[1]
                launches_per_batch = 1 if camera else max(1, len(ins) - 1)
                streamed = (ins[0] * STAGE_STREAM_IN_B0) if camera else sum(v * STAGE_STREAM_IN_B for v in ins[1:]) / launches_per_batch
                fb = raw + 0.5 * min(streamed, 2.0 * raw); how = "raw + streamed words / 2 (item counts of the bench line under rocprof)"
        elif "k_raygen" in k:
            fb = raw; how = "no streamed reads: x1"
        e = {"launches": n, "avg_launch_ms": ns / 1e6, "fetch_raw_GB_per_launch": raw / 1e9, "fetch_GB_per_launch": fb / 1e9, "fetch_factor": fb / raw if raw > 0 else None, "fetch_factor_how": how,
             "write_GB_per_launch": wb / 1e9, "traffic_TBps": (fb + wb) / ns / 1e3,
             "frac_of_hbm_peak_8TBps": (fb + wb) / ns / 8000.0, "traffic_TBps_if_fetch_x2": (2.0 * raw + wb) / ns / 1e3}
        if "SQ_INSTS_VALU" in o:
            e["valu_instructions_per_ns_and_simd"] = o["SQ_INSTS_VALU"] / o["dispatches_sq"] / 1024.0 / (o["total_ns_sq"] / o["dispatches_sq"])
        if "TCC_HIT_sum" in o:
            e["l2_hit_rate"] = o["TCC_HIT_sum"] / max(1.0, o["TCC_HIT_sum"] + o["TCC_MISS_sum"])
        if "TA_BUSY_avr" in o and "GRBM_GUI_ACTIVE" in o:
            e["ta_busy_avg"] = o["TA_BUSY_avr"] / (o["GRBM_GUI_ACTIVE"] / 8.0)
        others[k.replace("void ", "").replace("art::", "")] = e
    out["stage_kernels"] = others
    b = os.path.join(SRC, "bench_under_rocprof.json")
    if os.path.exists(b):
        shutil.copyfile(b, os.path.join(dst, "bench_under_rocprof.json"))
        # bench.py only quotes this profile for runs with the same fingerprint (source hash, scene, frame, options, tree)
        out["fingerprint"] = json.load(open(b))["config"]["fingerprint"]
    for k in list(out):
        if "k_trace_coop" in k:
            out["k_trace_coop"] = out[k]
    json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r3_final", sys.argv[2] if len(sys.argv) > 2 else "c4")

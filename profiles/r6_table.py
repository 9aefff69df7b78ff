#!/usr/bin/env python3
"""DESIGN.md section 6 table rows from the bench lines of a collection (profiles/<tag>/configs/*.json).  usage: python3 profiles/r6_table.py r6_final"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6_final"
d = os.path.join(ROOT, "profiles", tag, "configs")
def L(n):
    return json.load(open(os.path.join(d, n + ".json")))
rows = [("C2", "c2", "512×512 × 16"), ("C3", "c3", "1024×1024 × 64"), ("C4", "c4", "1920×1080 × 256"), ("C4, the driver's command line (20 steps, 5 warm-up)", "c4_steps20_warmup5", "1920×1080 × 256"),
        ("C5", "c5", "4096×4096 × 1024"), ("S4", "s4", "1920×1080 × 256"), ("I64", "i64", "1920×1080 × 64 per step"), ("C4, host buffers handed over every pass", "c4_host", ""), ("C4, two contexts on the one GPU", "c4_ctx2", ""), ("C4, eight contexts on the one GPU", "c4_ctx8", "")]
for name, f, frame in rows:
    try:
        x = L(f)
    except Exception as e:
        print("|", name, "| missing |"); continue
    r = x.get("roofline") or {}; w = x.get("whole_job_roofline") or {}; c = x.get("cpu_baseline") or {}; st = x.get("stages") or {}
    spp = x["config"]["spp_per_step"]; full = {"c2": 16, "c3": 64, "c5": 1024}.get(f, 256 if f != "i64" else 64)
    t = x["ms_per_step"] * full / spp / 1e3
    print("| %s | %s | %s | **%.0f** | %s | %s | %s | %s | %s | %s | %s |" % (name, frame, ("%.2f ms" % (t * 1e3)) if t < 0.01 else ("%.3f s" % t), x["value"], r.get("trace_Mrays_per_s"), r.get("whole_over_trace_only"), r.get("frac"), w.get("frac"),
          x["config"]["rays_per_sample"], (st.get("shade") or {}).get("ms_per_batch"), c.get("value")))
reps = []
for k in range(2, 9):
    try: reps.append(L("c2_rep%d" % k)["value"])
    except Exception: pass
print("C2 repetitions:", reps)

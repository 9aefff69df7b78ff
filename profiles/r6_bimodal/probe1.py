#!/usr/bin/env python3
"""Round 6, review item 4: C3's shade stage is bimodal between processes (1.13 / 1.37 ms per launch, the same binary).  Step 1 of the
diagnosis, from inside ONE process: render the same C3 batch (1024 x 1024 x 64 spp) again and again, tearing the backend down and
bringing it up between the renders (art_shutdown frees the path state; the next render allocates it again), and print where the path
state landed (ART_DEBUG_ADDR) next to the stage's time per launch.  If the mode flips inside one process, it follows the ALLOCATION
(virtual address or physical placement), not the process.  Then the same with frames around C3's: is it the power-of-two stride?
usage (gpurun): ART_DEBUG_ADDR=1 python3 profiles/r6_bimodal/probe1.py [cycles]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
art = ge.load_package()
from ada_ray_tracer_amd import scenes

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sd = scenes.synthetic_scene(100000, 3)


def run(W, H, T, tag, reps=1):
    be = art.Backend(0)
    be.set_option("shade_per", 4)
    be.upload_scene(sd); be.resize(W, H)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, T, seed=1)
    spp = be.render_pass_device(p, 0)                      # warm
    for r in range(reps):
        g0 = be.stage_stats(); s0 = be.stats()
        spp = be.render_pass_device(p, spp)
        g1 = be.stage_stats(); s1 = be.stats()
        n = g1.shade_launches - g0.shade_launches
        print(json.dumps({"tag": tag, "W": W, "H": H, "spp": 4 * T, "rep": r, "shade_ms_per_launch": round((g1.shade_ms - g0.shade_ms) / max(1, n), 4), "launches": int(n),
                          "fold_ms": round(g1.fold_ms - g0.fold_ms, 3), "raygen_ms": round(g1.raygen_ms - g0.raygen_ms, 3),
                          "trace_ms_per_launch": round((s1.trace_ms - s0.trace_ms) / max(1, s1.trace_launches - s0.trace_launches), 4)}), flush=True)
    be.shutdown()


for c in range(cycles):
    run(1024, 1024, 16, "c3 cycle %d" % c, reps=2)
for (W, H) in ((1000, 1000), (1024, 1000), (1056, 1024), (1024, 1024)):
    run(W, H, 16, "frame %dx%d" % (W, H), reps=1)

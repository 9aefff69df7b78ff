#!/usr/bin/env python3
"""Round 6, review item 4, step 4.  probe2: identical device addresses in every process (address randomisation off) and the stage is still
fast in some processes and slow in others, alternating run by run -- what is left to differ is WHICH PHYSICAL MEMORY the driver hands the
35 GB of path state.  From inside one process: hold a dummy allocation of X GB while the path state is allocated (so that the driver has to
take other physical blocks for it), render the same C3 batch, release everything, next X.  If the stage's time follows X, the mode is the
physical placement of the path state.
usage (gpurun): ART_DEBUG_ADDR=1 python3 profiles/r6_bimodal/probe4.py"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
art = ge.load_package()
from ada_ray_tracer_amd import scenes

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
sd = scenes.synthetic_scene(100000, 3)


def run(tag, dummies_gb):
    be = art.Backend(0)
    be.set_option("shade_per", 4)
    be.upload_scene(sd)
    held = []
    for g in dummies_gb:                                   # held while the path state is allocated (the first render pass allocates it)
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), int(g * (1 << 30)))
        if rc != 0:
            print("hipMalloc of %s GB failed: %d" % (g, rc)); break
        held.append(p)
    be.resize(1024, 1024)
    prm = art.Backend.pass_params(art.PT_MIS, True, 8, 16, seed=1)
    spp = be.render_pass_device(prm, 0)
    g0 = be.stage_stats(); s0 = be.stats()
    for _ in range(2):
        spp = be.render_pass_device(prm, spp)
    g1 = be.stage_stats(); s1 = be.stats()
    n = g1.shade_launches - g0.shade_launches
    print(json.dumps({"tag": tag, "held_GB": dummies_gb, "shade_ms_per_batch": round((g1.shade_ms - g0.shade_ms) / 2, 3), "shade_ms_per_launch": round((g1.shade_ms - g0.shade_ms) / max(1, n), 4),
                      "trace_ms_per_launch": round((s1.trace_ms - s0.trace_ms) / max(1, s1.trace_launches - s0.trace_launches), 4)}), flush=True)
    be.shutdown()
    for p in held:
        hip.hipFree(p)


run("nothing held", [])
run("nothing held (again)", [])
for g in (1, 3, 7, 16, 33, 40, 64, 100):
    run("%d GB held" % g, [g])
run("nothing held (after)", [])
run("two blocks held", [5, 11])
run("nothing held (last)", [])

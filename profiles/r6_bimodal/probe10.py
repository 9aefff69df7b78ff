#!/usr/bin/env python3
"""Round 6, review item 4, step 10.  probe8 / probe9: physically contiguous path state = the slowest stage (C3 15.0 ms per batch), whatever the
pad between the fields; the driver's default placement gives 12.1 or 10.4 depending on the process.  If regularity of the physical layout
is what hurts, a deliberately FRAGMENTED device memory should help: fill the device with dummy blocks of G MB, free every other one (free
memory = holes of G MB between blocks that stay), allocate the path state into the holes, render C3, compare with the same process before
the fragmentation.  One process per G (argv[1], MB; 0 = no fragmentation, twice).
usage (gpurun): ART_DEBUG_ADDR=1 python3 profiles/r6_bimodal/probe10.py <G in MB>"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
art = ge.load_package()
from ada_ray_tracer_amd import scenes

G = int(sys.argv[1]) if len(sys.argv) > 1 else 0
SCENE = sys.argv[2] if len(sys.argv) > 2 else "c3"        # c3 | c4 | s4 | c5
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
sd = {"c3": lambda: scenes.synthetic_scene(100000, 3), "c4": lambda: scenes.synthetic_scene(1000000, 4), "s4": lambda: scenes.structured_scene(1000000),
      "c5": lambda: scenes.mixed_scene(20000, 5)}[SCENE]()
W, H, T = {"c3": (1024, 1024, 16), "c4": (1920, 1080, 16), "s4": (1920, 1080, 16), "c5": (4096, 4096, 2)}[SCENE]


def render(tag):
    be.resize(W, H)
    prm = art.Backend.pass_params(art.PT_MIS, True, 8, T, seed=1)
    spp = be.render_pass_device(prm, 0)
    g0 = be.stage_stats(); s0 = be.stats()
    for _ in range(2):
        spp = be.render_pass_device(prm, spp)
    g1 = be.stage_stats(); s1 = be.stats()
    print(json.dumps({"scene": SCENE, "G_MB": G, "tag": tag, "shade_ms_per_batch": round((g1.shade_ms - g0.shade_ms) / 2, 3), "fold_ms_per_batch": round((g1.fold_ms - g0.fold_ms) / 2, 3),
                      "raygen_ms_per_batch": round((g1.raygen_ms - g0.raygen_ms) / 2, 3),
                      "trace_ms_per_launch": round((s1.trace_ms - s0.trace_ms) / max(1, s1.trace_launches - s0.trace_launches), 4)}), flush=True)


be = art.Backend(0)
be.set_option("shade_per", 4)
be.upload_scene(sd)
render("as the driver places it")
if G > 0:
    be.set_option("paths_contiguous", 0)                   # (releases the path state)
    fr, tot = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    n = int((fr.value - (6 << 30)) // (G << 20))           # leave 6 GB alone
    blocks = []
    for k in range(n):
        p = C.c_void_p()
        if hip.hipMalloc(C.byref(p), G << 20) != 0:
            break
        blocks.append(p)
    for k in range(0, len(blocks), 2):
        hip.hipFree(blocks[k]); blocks[k] = None
    hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
    print("fragmented: %d blocks of %d MB held, %.1f GB free in holes of %d MB" % (sum(b is not None for b in blocks), G, fr.value / 2**30, G), flush=True)
    render("into holes of %d MB" % G)
    be.set_option("paths_contiguous", 0)
    for b in blocks:
        if b is not None:
            hip.hipFree(b)
    render("after the blocks were freed")
be.shutdown()

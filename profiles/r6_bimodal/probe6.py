#!/usr/bin/env python3
"""Round 6, review item 4, step 6.  probe5 (eight processes under rocprofv3): the modes ALTERNATE process by process (slow, fast, slow, fast,
...: 1.391 / 1.166 ms per launch), the instruction cache misses the same 4 000 times in both, and what the slow mode has more of is
SQ_WAIT_INST_ANY (+43 %): waves that could issue and did not get the slot -- more waves crowded on fewer SIMDs, i.e. the DISPATCH differs,
not the memory system.  What a process gets from the driver in turn is its hardware queue.  From inside one process: the same C3 render
on the null stream and on eight freshly created streams (HIP spreads streams over its hardware queues), the stage's time per launch each.
usage (gpurun): python3 profiles/r6_bimodal/probe6.py"""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
art = ge.load_package()
from ada_ray_tracer_amd import scenes

hip = C.CDLL("libamdhip64.so")
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
hip.hipStreamCreateWithPriority.argtypes = [C.POINTER(C.c_void_p), C.c_uint, C.c_int]
sd = scenes.synthetic_scene(100000, 3)
be = art.Backend(0)
be.set_option("shade_per", 4)
be.upload_scene(sd); be.resize(1024, 1024)
prm = art.Backend.pass_params(art.PT_MIS, True, 8, 16, seed=1)
spp = be.render_pass_device(prm, 0)


def measure(tag):
    global spp
    g0 = be.stage_stats(); s0 = be.stats()
    for _ in range(2):
        spp = be.render_pass_device(prm, spp)
    g1 = be.stage_stats(); s1 = be.stats()
    n = g1.shade_launches - g0.shade_launches
    print(json.dumps({"stream": tag, "shade_ms_per_launch": round((g1.shade_ms - g0.shade_ms) / max(1, n), 4), "fold_ms_per_batch": round((g1.fold_ms - g0.fold_ms) / 2, 3),
                      "trace_ms_per_launch": round((s1.trace_ms - s0.trace_ms) / max(1, s1.trace_launches - s0.trace_launches), 4)}), flush=True)


measure("null")
streams = []
for k in range(8):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0          # hipStreamNonBlocking
    streams.append(s)
    be.set_stream(s.value)
    measure("created #%d" % k)
be.set_stream(None)
measure("null again")
for k in (0, 1, 2, 3):
    be.set_stream(streams[k].value)
    measure("created #%d again" % k)
for prio in (-1, 0, 1):                                                   # other priorities = other hardware queues
    s = C.c_void_p()
    if hip.hipStreamCreateWithPriority(C.byref(s), 1, prio) == 0:
        be.set_stream(s.value)
        measure("priority %d" % prio)
be.set_stream(None)
be.shutdown()

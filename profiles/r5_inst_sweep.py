#!/usr/bin/env python3
"""I64 through the cooperative kernel under the trace kernel's scheduling options (node_min, refill_min, ray_chunk): are the optima measured on
the flattened C4 tree also the instanced kernel's?  usage: python profiles/r5_inst_sweep.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "profiles"))
import __graft_entry__ as ge  # noqa: E402
import r5_instanced as ri     # noqa: E402

art = ge.load_package()
from ada_ray_tracer_amd import scenes  # noqa: E402
sd = scenes.instanced_scene(64, 20000)
be = art.Backend(0)
DEF = {"node_min": 0, "refill_min": 2, "ray_chunk": 48}
out = []
import numpy as np  # noqa: E402
c = np.array([0.0, 2.3, 2.3]); tr = []
for i in range(64):
    M = np.array(list(sd.instances[i].m), np.float64).reshape(3, 4)
    M[:, 3] = c + 0.25 * (M[:, 3] - c); M[:, :3] *= 1.5
    tr.append((int(sd.instances[i].mesh), M))
cluster = scenes.instanced_scene(64, 20000, transforms=tr)
full = len(sys.argv) > 1 and sys.argv[1] == "full"
for name, vals in ((("node_min", (4, 2, 3, 5, 6, 4)), ("refill_min", (1, 3, 4)), ("ray_chunk", (32, 64))) if full else (("node_min", (4, 1, 2, 3, -4, -1, -2)),)):
    for v in vals:
        scene = cluster if v < 0 else sd
        v = abs(v)
        be.set_option(name, v)
        r = ri.run(art, be, scene, "%s%s = %d" % ("CLUSTER " if scene is cluster else "", name, v))
        be.set_option(name, DEF[name])
        out.append({"variant": r["variant"], "Mrays_per_s": r["Mrays_per_s"], "wave_iters_per_kray": r["counters"]["wave_iters_per_kray"]})
        print(out[-1], flush=True)
be.shutdown()

#!/usr/bin/env python3
"""Instanced scene I64 (64 instances of two ~20 k-triangle meshes): Mrays/s of the render loop through (a) the cooperative kernel crossing the
instance boundary, (b) the one-ray-per-lane two-level kernel (option inst_coop = 0), (c) the explicitly flattened upload (1.28 M triangles,
one tree) -- the same picture, bit for bit (tests/test_gpu_instanced.py).  usage: python profiles/r5_instanced.py > profiles/r5_instanced.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402


def run(art, be, sd, label, opts=()):
    for k, v in opts:
        be.set_option(k, v)
    t0 = time.time(); be.upload_scene(sd); t_up = time.time() - t0
    info = be.bvh_info()
    be.resize(1920, 1080)
    prm = art.Backend.pass_params(art.PT_MIS, True, 8, 16, seed=1)       # 64 spp per step
    spp = be.render_pass_device(prm, 0)
    be.synchronize(); s0 = be.stats(); t0 = time.perf_counter()
    for _ in range(2):
        spp = be.render_pass_device(prm, spp)
    be.synchronize(); dt = time.perf_counter() - t0; s1 = be.stats()
    # traversal counters per ray (the counting variant of the same kernel, one untimed 4-spp pass)
    be.set_option("count_tests", 1)
    c0 = be.stats(); be.render_pass_device(art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=1), spp); c1 = be.stats()
    be.set_option("count_tests", 0)
    nr = max(1, c1.traced_rays - c0.traced_rays)
    counters = {"node_visits_per_ray": round((c1.node_visits - c0.node_visits) / nr, 2), "leaf_visits_per_ray": round((c1.leaf_visits - c0.leaf_visits) / nr, 2),
                "tri_tests_per_ray": round((c1.tri_tests - c0.tri_tests) / nr, 2), "box_tests_per_ray": round((c1.box_tests - c0.box_tests) / nr, 2),
                "wave_iters_per_kray": round(1000.0 * (c1.wave_iters - c0.wave_iters) / nr, 1), "node_phase_iters_per_kray": round(1000.0 * (c1.node_phase_iters - c0.node_phase_iters) / nr, 1),
                "leaf_phase_iters_per_kray": round(1000.0 * (c1.leaf_phase_iters - c0.leaf_phase_iters) / nr, 1)}
    for k, v in opts:
        be.set_option(k, 1 if k == "inst_coop" else 0 if k == "inst_open" else 0)
    return {"counters": counters, "variant": label, "Mrays_per_s": round((s1.rays - s0.rays) / dt / 1e6, 1), "ms_per_64spp_step": round(dt * 500.0, 1), "trace_ms_per_step": round((s1.trace_ms - s0.trace_ms) / 2, 1),
            "scene_upload_s": round(t_up, 2), "tree_nodes": info.n_nodes, "tree_triangle_records": info.n_tris}


def main():
    art = ge.load_package()
    import hostsim
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(64, 20000)
    flat = hostsim.flattened_copy(art, sd)
    be = art.Backend(0)
    out = [run(art, be, sd, "instanced, cooperative kernel (k_trace_coop<.., INST>), inst_open = 0 (default: chosen by the build -- whole instances here)")]
    for k in (1, 2, 4, 8, 16, 64):         # entry points per instance the instance tree ends at (art_instanced_build.h)
        out.append(run(art, be, sd, "instanced, cooperative kernel, inst_open = %d" % k, [("inst_open", k)]))
    out += [run(art, be, sd, "instanced, one ray per lane (k_trace_inst)", [("inst_coop", 0)]),
            run(art, be, flat, "flattened upload: one tree over 1.28 M world-space triangles")]
    # the same 64 instances pulled together into one interpenetrating cluster (translations shrunk to a quarter around the centre of the box,
    # scales x 1.5): the boxes of whole instances nearly coincide -- the case opening instances is for
    sd2 = scenes.instanced_cluster(64, 20000)
    for k in (0, 1, 4, 16, 64):
        out.append(run(art, be, sd2, "CLUSTER of the 64 instances, cooperative kernel, inst_open = %d" % k, [("inst_open", k)]))
    out.append(run(art, be, hostsim.flattened_copy(art, sd2), "CLUSTER, flattened upload"))
    print(json.dumps({"workload": "I64, 1920x1080, PT_MIS depth 8, 2x2 AA, 64 spp per step, 2 timed steps", "results": out}, indent=1))
    be.shutdown()


if __name__ == "__main__":
    main()

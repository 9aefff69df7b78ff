#!/usr/bin/env python3
"""Round 6, review item 3: ONE whole BASELINE frame against the oracle, every pixel, once.

C3 (100 k random triangles + 3 sphere lights in the Cornell box), 1024 x 1024, PT_MIS depth 8, 2x2 AA -- the GPU frame through the C ABI
against the oracle's orc_render_pass (the restatement of DoPass / PathTrace, ray_tracer-integrators.adb:25-71, 203-301):

  1. the product's exported tree is checked on its own first (tests/bvh_check.py: every triangle exactly once, every child box strictly
     encloses what hangs below it, stack bound) and the trace kernel's hits against the oracle's O(N) scan on random rays;
  2. the WHOLE frame at --spp samples per pixel, the oracle's mesh search walking that brute-force-checked tree (an O(N) scan of the whole
     frame is ~1e14 triangle tests: out of reach);
  3. a --window x --window block of the frame with the oracle's OWN closest-hit search -- the O(N) scan over all triangles
     (oracle/art_oracle.c intersect_mesh_closest without a tree attached): no product tree anywhere on the oracle's side.

Not part of the driver's `-m gpu` budget (about two minutes of 16 host cores): run under gpurun,
    python3 profiles/r6_parity/whole_frame.py --out gpurun_out/r6/c3_whole_frame.json
The oracle is test infrastructure; this script is a checker like tests/, nothing of the product imports it.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c3", choices=["c3", "c4", "c5", "i64"], help="c3 (default): 100 k triangles, 1024 x 1024; c4: 1 M triangles, 1920 x 1080; c5: mixed scene, 4096 x 4096 "
                    "(c4 / c5: choose --spp so that the oracle finishes -- about 30 s of 16 cores per 100 M rays at 1 M triangles)")
    ap.add_argument("--spp", type=int, default=64, help="samples per pixel of the whole-frame comparison (C3's stated count: 64)")
    ap.add_argument("--window", type=int, default=48, help="side of the block compared with the oracle's O(N) scan (48 x 48 x 16 spp on C3: 1.8e10 triangle tests)")
    ap.add_argument("--window-spp", type=int, default=16)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--oracle-chunk", type=int, default=16, help="Threads_Num of one oracle Render_Pass call (4 spp each); the calls accumulate")
    ap.add_argument("--out", default="")
    # a frame whose oracle render does not fit one gpurun call (20 minutes): the oracle's accum buffer, sample count and ray count are carried
    # from call to call in a file; every call the GPU renders the whole count so far in ONE pass from zero and the two frames are compared
    ap.add_argument("--carry-in", default="", help=".npz of an earlier call (accum, spp, rays): the oracle continues from there")
    ap.add_argument("--carry-out", default="", help="where to leave the oracle's state for the next call")
    ap.add_argument("--oracle-spp", type=int, default=0, help="samples per pixel the oracle adds in this call (0: all of --spp); the GPU renders carried + added")
    ap.add_argument("--skip-tree", action="store_true", help="skip step 1 (done by the first call of a chain)")
    args = ap.parse_args()
    art = ge.load_package()
    import bvh_check
    import conv
    import orc
    from ada_ray_tracer_amd import scenes
    carried = np.load(args.carry_in) if args.carry_in else None
    spp_in = int(carried["spp"]) if carried is not None else 0
    add = args.oracle_spp if args.oracle_spp > 0 else args.spp
    if carried is not None or args.oracle_spp > 0:
        args.spp = spp_in + add                          # this call compares the frame at carried + added samples
    T = args.spp // 4
    if args.config == "c3":
        W, H, ntris = 1024, 1024, 100000
        sd = scenes.synthetic_scene(ntris, 3)
        name = "C3: synthetic %d triangles + 3 sphere lights in the Cornell box" % ntris
    elif args.config == "c4":
        W, H, ntris = 1920, 1080, 1000000
        sd = scenes.synthetic_scene(ntris, 4)
        name = "C4: synthetic %d triangles + 3 sphere lights in the Cornell box" % ntris
    elif args.config == "c5":
        W, H, ntris = 4096, 4096, 20000
        sd = scenes.mixed_scene(ntris, 5)
        name = "C5: spheres (glass / Phong / diffuse / emissive) + %d-triangle mesh" % ntris
    else:
        # I64: the GPU renders the INSTANCED scene through the two-level tree (k_trace_coop<.., INST>); the oracle has no instancing and
        # renders the explicitly flattened scene (tests/hostsim.flattened_copy: the product's flatten_instances), its mesh search walking
        # the tree of the flattened UPLOAD -- checked like the others in step 1.  The picture must be the same bit for bit.
        import hostsim
        W, H = 1920, 1080
        inst_sd = scenes.instanced_scene(64, 20000)
        sd = hostsim.flattened_copy(art, inst_sd)
        ntris = int(sd.desc.meshes[0].ntris)
        name = "I64: 64 instances of two ~20000-triangle meshes (%d triangles flattened) + 3 sphere lights, rendered through the two-level tree; oracle on the flattened scene" % ntris
    be = art.Backend(0)
    be.upload_scene(sd)
    out = {"config": "%s, %dx%d, PT_MIS depth 8, 2x2 AA" % (name, W, H), "seed": 1}

    # ---- 1. the tree the oracle is going to walk, checked without any traversal, and the product's hits against the O(N) scan
    nodes, tris, info = be.export_bvh()
    pos, _, idx, _, _ = [a for a, m in zip(sd._mesh_arrays, sd.meshes) if m.mode == art.MESH_CLOSEST][0]
    r = bvh_check.check_tree(nodes, tris, info.n_nodes, info.max_stack, info.node_width, pos, idx) if not args.skip_tree else {"records": int(info.n_tris), "depth": -1, "worst_stack": -1}
    rng = np.random.default_rng(20260)
    n_rays = (20000 if ntris <= 100000 else 4000) if not args.skip_tree else 8
    o = np.stack([rng.uniform(-2.4, 2.4, n_rays), rng.uniform(0.1, 4.9, n_rays), rng.uniform(0.1, 4.9, n_rays)], 1).astype(np.float32)
    d = rng.normal(size=(n_rays, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    osc_bf = conv.OracleScene(sd)                       # no tree attached: the oracle's own O(N) scan
    hg = conv.hits_to_arrays(be.trace_rays(o, d)); ho = conv.hits_to_arrays(orc.closest_hits(osc_bf.scene, o, d))
    is_hit = ho[1] == 1                                  # (t, prim, material and normal of a miss carry no meaning: compared where the oracle hit, like tests/test_gpu_parity.py)
    same_hits = bool(np.array_equal(hg[1], ho[1]) and np.array_equal(hg[2], ho[2]))
    for k in (0, 3, 4, 5):
        a, b = np.asarray(hg[k])[is_hit], np.asarray(ho[k])[is_hit]
        same_hits = same_hits and bool(np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b))
    out["tree"] = {"nodes": int(info.n_nodes), "records": int(r["records"]), "depth": int(r["depth"]), "worst_stack": int(r["worst_stack"]), "structural_check": "passed" if not args.skip_tree else "skipped (first call of the chain)",
                   "random_rays_vs_O(N)_scan": n_rays, "random_rays_hitting": int(is_hit.sum()), "random_rays_equal": bool(same_hits)}
    print("tree checked:", out["tree"], flush=True)

    # ---- 2. the whole frame
    if args.config == "i64":
        be.upload_scene(inst_sd)                         # from here on the GPU renders the instanced scene; `sd` stays the oracle's (flattened) scene
    be.resize(W, H)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, T, seed=1)
    t0 = time.time()
    gpu, screen, spp = be.render_pass(p, 0, True, True)
    t_gpu = time.time() - t0
    st = be.stats()
    assert spp == 4 * T and st.lost_paths == 0
    osc = conv.OracleScene(sd)
    osc.attach_bvh(nodes, tris, info.node_width)
    t0 = time.time()
    # the oracle in Render_Pass calls of at most --oracle-chunk tasks (g_accBuff is cumulative over calls, ray_tracer.adb:281-285: the same sum in
    # the same order as one pass of T tasks -- tests/test_gpu_stated_spp.py::test_c3_two_passes_of_eight_tasks_equal_one_pass_of_sixteen), so
    # that a long render says something every minute or two
    ref = None; rspp = 0; rays_o = 0
    if carried is not None:
        ref = np.ascontiguousarray(carried["accum"], np.float32); rspp = spp_in; rays_o = int(carried["rays"])
        print("oracle: continuing from %d spp (%d rays so far)" % (rspp, rays_o), flush=True)
    while rspp < spp:
        tc = min(args.oracle_chunk, (spp - rspp) // 4)
        ref, rspp, cnt = orc.render(osc.scene, orc.make_params(W, H, orc.PT_MIS, True, 8, tc, seed=1, nthreads=args.threads), accum=ref, spp0=rspp)
        rays_o += cnt.rays
        print("oracle: %d of %d spp, %.0f s" % (rspp, spp, time.time() - t0), flush=True)
    t_cpu = time.time() - t0
    assert rspp == spp
    if args.carry_out:
        os.makedirs(os.path.dirname(os.path.abspath(args.carry_out)), exist_ok=True)
        np.savez(args.carry_out, accum=ref, spp=np.int64(rspp), rays=np.int64(rays_o))
    class _C: pass
    cnt = _C(); cnt.rays = rays_o
    diff = bits(gpu) != bits(ref)
    mism_px = int(np.count_nonzero(diff.any(-1)))
    out["whole_frame"] = {"pixels": W * H, "spp": spp, "camera_samples": W * H * spp, "mismatching_pixels": mism_px, "mismatching_channels": int(np.count_nonzero(diff)),
                          "max_abs_diff_per_spp": float(np.abs(gpu - ref).max() / spp), "all_finite": bool(np.isfinite(ref).all() and np.isfinite(gpu).all()),
                          "rays_gpu": int(st.rays), "rays_oracle": int(cnt.rays), "rays_equal": bool(st.rays == cnt.rays),
                          "ldr_frame_equal": bool(np.array_equal(screen, orc.resolve(ref, spp))),
                          "sha256_gpu_accum": hashlib.sha256(bits(gpu).tobytes()).hexdigest(), "sha256_oracle_accum": hashlib.sha256(bits(ref).tobytes()).hexdigest(),
                          "nonblack_fraction": float((ref.sum(-1) > 0).mean()), "gpu_s": round(t_gpu, 3), "oracle_s": round(t_cpu, 1),
                          "oracle_mesh_search": "walk of the product's exported tree (checked in step 1)",
                          "oracle_samples_carried_in": spp_in, "oracle_samples_added_in_this_call": spp - spp_in}
    print("whole frame:", out["whole_frame"], flush=True)

    # ---- 3. a window of the frame with the oracle's own O(N) closest-hit scan (no product tree on the oracle's side)
    w = args.window; Tw = args.window_spp // 4
    if w > 0:
        x0 = (W - w) // 2 - W // 8; y0 = (H - w) // 2 + H // 16                      # off-centre: triangles, floor and a light's reflection in view
        ys, xs = np.mgrid[y0:y0 + w, x0:x0 + w]
        xs = xs.ravel(); ys = ys.ravel()
        if Tw != T:
            be.resize(W, H)
            gpu_w, _, spp_w = be.render_pass(art.Backend.pass_params(art.PT_MIS, True, 8, Tw, seed=1), 0)
        else:
            gpu_w, spp_w = gpu, spp
        t0 = time.time()
        ref_w, cnt_w = orc.render_pixels(osc_bf.scene, orc.make_params(W, H, orc.PT_MIS, True, 8, Tw, seed=1, nthreads=args.threads), xs, ys)
        t_bf = time.time() - t0
        dw = bits(gpu_w[ys, xs]) != bits(ref_w)
        out["window_brute_force"] = {"x0": int(x0), "y0": int(y0), "side": w, "pixels": int(xs.size), "spp": int(spp_w), "mismatching_pixels": int(np.count_nonzero(dw.any(-1))),
                                     "oracle_rays": int(cnt_w.rays), "oracle_triangle_tests": int(cnt_w.tri_tests), "oracle_s": round(t_bf, 1),
                                     "nonblack_fraction": float((ref_w.sum(-1) > 0).mean()),
                                     "sha256_gpu": hashlib.sha256(bits(gpu_w[ys, xs]).tobytes()).hexdigest(), "sha256_oracle": hashlib.sha256(bits(ref_w).tobytes()).hexdigest(),
                                     "oracle_mesh_search": "O(N) scan over all triangles (oracle/art_oracle.c intersect_mesh_closest, no tree attached)"}
        print("window:", out["window_brute_force"], flush=True)
    be.shutdown()
    ok = same_hits and out["whole_frame"]["mismatching_pixels"] == 0 and out["whole_frame"]["rays_equal"] and (w <= 0 or out["window_brute_force"]["mismatching_pixels"] == 0)
    out["verdict"] = "bit-equal" if ok else "MISMATCH"
    txt = json.dumps(out, indent=1)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, "w").write(txt + "\n")
    print(txt)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

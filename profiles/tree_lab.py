#!/usr/bin/env python3
"""CPU laboratory for tree-quality experiments (no GPU needed): builds the C4 mesh with the HOST builder under different
parameters (the GPU SAH builder builds the same tree, tests/test_gpu_lbvh.py) and counts node / leaf visits and triangle tests of the
published traversal order on a path-tracing-like ray set (camera rays, two generations of diffuse bounce rays, shadow rays towards
the lights) with the test-only host simulation (tests/host_sim).  Visits per ray are what the trace kernel's time follows
(DESIGN.md section 5), so a parameter is worth a GPU run only if it moves them here.
usage: python profiles/tree_lab.py [ntris] name=value[,name=value...] ...      e.g.  collapse=1  sah_bins=64  collapse=1,sah_bins=64"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

F = np.float32


def ray_set(art, hostsim, sd, n_cam=12000, seed=3):
    rng = np.random.default_rng(seed)
    cam = np.array([0.0, 2.55, 12.5], F)
    px = rng.random((n_cam, 2)) * [1920, 1080]
    d = np.stack([px[:, 0] - 960.0, px[:, 1] - 540.0, np.full(n_cam, -1920.0)], 1)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)
    sets = [(np.tile(cam, (n_cam, 1)), d, np.full(n_cam, np.inf, F))]
    o, dd = sets[0][0], sets[0][1]
    lights = np.array([[-1.5, 4.5, 2.25], [0.0, 4.5, 2.25], [1.5, 4.5, 2.25]], F)
    for gen in range(3):
        hits, _ = hostsim.trace(art, sd, o, dd)
        ok = np.array([h.is_hit for h in hits], bool)
        t = np.array([h.t for h in hits], F); nrm = np.array([list(h.normal) for h in hits], F)
        ok &= np.isfinite(t) & (np.linalg.norm(nrm, axis=1) > 0.5)
        p = (o + dd * t[:, None])[ok]; n = nrm[ok]; n /= np.linalg.norm(n, axis=1, keepdims=True)
        flip = np.sign(-(dd[ok] * n).sum(1))[:, None]; n = n * np.where(flip == 0, 1, flip)
        v = rng.normal(size=p.shape); v /= np.linalg.norm(v, axis=1, keepdims=True)
        nd = n + 0.999 * v; nd = (nd / np.linalg.norm(nd, axis=1, keepdims=True)).astype(F)     # cosine-distributed about the normal
        po = (p + n * 1.0e-4).astype(F)
        L = lights[rng.integers(0, 3, p.shape[0])] + rng.normal(size=p.shape).astype(F) * 0.2
        sdv = L - po; dist = np.linalg.norm(sdv, axis=1); sdv = (sdv / dist[:, None]).astype(F)
        sets.append((po, nd, np.full(po.shape[0], np.inf, F)))                                    # bounce rays
        sets.append((po, sdv, (dist * 0.999).astype(F)))                                          # shadow rays (closest hit below the light distance)
        o, dd = po, nd
    return [np.concatenate([s[k] for s in sets]) for k in range(3)]


def main():
    args = sys.argv[1:]
    ntris = int(args.pop(0)) if args and args[0].isdigit() else 1000000
    art = ge.load_package()
    import hostsim
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(ntris, 4)
    t0 = time.time()
    o, d, tf = ray_set(art, hostsim, sd)
    print("ray set: %d rays (%.1f s)" % (o.shape[0], time.time() - t0), flush=True)
    defaults = dict(collapse=0, sah_bins=32, node_cost=0.4, leaf_base=1.0, tri_cost=-1.0, max_leaf=8)
    base = None
    for spec in ["default"] + args:
        for k, v in defaults.items():
            hostsim.set_bvh_param(art, k, v)
        if spec != "default":
            for kv in spec.split(","):
                k, v = kv.split("="); hostsim.set_bvh_param(art, k, float(v))
        t0 = time.time()
        _, st = hostsim.trace(art, sd, o, d, tf)
        n = o.shape[0]
        row = dict(box=st[0] / n, tri=st[1] / n, node=st[2] / n, leaf=st[3] / n)
        if base is None:
            base = row
        print("%-40s node visits %.2f (%+.1f%%)  leaf visits %.2f (%+.1f%%)  tri tests %.2f (%+.1f%%)   [%.1f s]" % (
            spec, row["node"], 100 * (row["node"] / base["node"] - 1), row["leaf"], 100 * (row["leaf"] / base["leaf"] - 1),
            row["tri"], 100 * (row["tri"] / base["tri"] - 1), time.time() - t0), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render loop on MI355X (BASELINE.json metric), one JSON line on stdout.

Workload (N=1 and N>1): BASELINE config C4 -- synthetic 1M random triangles + 3 sphere lights inside the Cornell
box (SURVEY 8d generator, seed 0xADA5EED0+4), 1920x1080, PT_MIS, Max_Trace_Depth 8, 2x2 AA.  One "step" is one
Render_Pass of `--vthreads` x 4 samples per pixel (default 64 spp); the full 256-spp render of C4 is the default 4 steps.
Rays = closest-hit queries actually issued (camera + bounce + shadow).  Scene upload and BVH build are
outside the timed region and reported separately in `config`.

N > 1: one process per GPU (torch.distributed / RCCL), interleaved 32x32 pixel tiles per rank (strong scaling of
the fixed frame), one RCCL reduce(sum) of the float3 framebuffer to rank 0 inside the timed region.

Extra objects on the line: "roofline" (algorithmic bytes of the trace kernel / its HIP-event time, SURVEY 8d
formula 32*B + 48*T + 64 bytes per ray with B, T counted on this very workload) and "cpu_baseline" (the CPU oracle,
BVH-accelerated, on a bounded sample of the same scene and camera, timed on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def build_scene(art, args):
    from ada_ray_tracer_amd import scenes
    if args.scene == "c4":
        return scenes.synthetic_scene(args.tris, 4), "C4: synthetic %d triangles + 3 sphere lights in the Cornell box" % args.tris
    if args.scene == "c3":
        return scenes.synthetic_scene(100000, 3), "C3: synthetic 100000 triangles + 3 sphere lights in the Cornell box"
    if args.scene == "c5":
        return scenes.mixed_scene(20000, 5), "C5: spheres + 20000-triangle mesh, glass/diffuse/emissive"
    import conv
    import orc
    cs = orc.CornellScene()
    return conv.desc_from_oracle(art, cs), "C2: internal Cornell scene with data/pyramid2.vsgf"


def measured_traffic(args, W, H):
    """HBM-side traffic of the trace kernel in GB/s from the committed rocprofv3 PMC passes (profiles/collect.sh ->
    profiles/summarize.py): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch / launch time.  The factor 2 on FETCH_SIZE is the
    gfx950 correction, calibrated on this kernel's own access pattern (profiles/calib_fetch.hip: 17.15 GB reported for
    34.36 GB of known 128-byte segment gathers).  Only valid for the profiled workload (default C4); null otherwise."""
    if args.scene != "c4" or args.tris != 1000000 or (W, H) != (1920, 1080) or args.kernel != "coop":
        return None
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r1_final", "pmc_summary.json")))
        k = [x for x in d if "k_trace_coop" in x][0]
        return round(d[k]["traffic_GBps_fetch_x2"], 1)
    except Exception:
        return None


def host_cpu_share():
    """CPUs this job may actually use: min(affinity mask, cgroup quota).  A 1-GPU box exposes 256 logical CPUs but caps the
    job at a share of them; timing an oversubscribed OpenMP team would misreport both the rate and the core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def cpu_baseline(art, sd, args):
    """CPU oracle (kind "port": the reference is Ada and cannot be built or shipped) on a bounded sample of the same
    workload: same scene, same camera, reduced frame (ray distribution preserved), PT_MIS depth 8, all host threads.
    The closest-hit mesh search walks the product's exported BVH (oracle/art_oracle.c: intersect_mesh_closest)."""
    import conv
    import orc
    w, h = args.cpu_width, args.cpu_height
    osc = conv.OracleScene(sd)
    be = args._backend
    if sd.desc.n_meshes and sd.desc.meshes[0].mode == art.MESH_CLOSEST:
        nodes, tris, binfo = be.export_bvh()
        osc.attach_bvh(nodes, tris, binfo.node_width)
    cores = args.cpu_threads if args.cpu_threads > 0 else host_cpu_share()
    prm = orc.make_params(w, h, orc.PT_MIS, True, 8, 1, seed=1, nthreads=cores)
    orc.render(osc.scene, orc.make_params(32, 18, orc.PT_MIS, True, 8, 1, seed=1, nthreads=cores))      # warm threads / caches
    t0 = time.time()
    passes = 0
    cnt_total = 0
    while True:
        _, _, cnt = orc.render(osc.scene, prm)
        cnt_total += cnt.rays
        passes += 1
        if time.time() - t0 > args.cpu_seconds or passes >= 64:
            break
    dt = time.time() - t0
    return {"value": round(cnt_total / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "same scene+camera at %dx%d, %d passes x 4 spp, PT_MIS depth 8, BVH closest-hit, %.1f s" % (w, h, passes, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="c4", choices=["c2", "c3", "c4", "c5"])
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--vthreads", type=int, default=16, help="Threads_Num of the pass: spp per step = 4 * vthreads (default 64: 4 steps = the 256-spp C4 render)")
    ap.add_argument("--kernel", default="coop", choices=["coop", "simple"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the cpu_baseline leg (0 = the job's CPU share)")
    ap.add_argument("--cpu-width", type=int, default=480)
    ap.add_argument("--cpu-height", type=int, default=270)
    ap.add_argument("--no-counters", action="store_true", help="skip the untimed B/T counting pass")
    ap.add_argument("--host-buffers", action="store_true", help="hand AccumBuff/screen_buffer over in host memory every pass (Ada layout), i.e. include PCIe")
    ap.add_argument("--simulate-shard", default="", help="R/N: render only rank R's pixel tiles of an N-GPU job on this one GPU (scaling rehearsal)")
    ap.add_argument("--opt", action="append", default=[], help="backend option name=value (art_set_option), e.g. bvh_leaf_base_milli=1000")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    use_dist = world > 1 or bool(os.environ.get("ART_BENCH_FORCE_DIST"))     # FORCE: rehearse the torch/RCCL path with one rank
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
    art = ge.load_package()
    be = art.Backend(local_rank if use_dist else 0)
    args._backend = be
    be.set_option("trace_kernel", art.TRACE_COOP if args.kernel == "coop" else art.TRACE_SIMPLE)
    for kv in args.opt:
        k, v = kv.split("=")
        be.set_option(k, int(v))

    t0 = time.time()
    sd, scene_name = build_scene(art, args)
    t_gen = time.time() - t0
    t0 = time.time()
    be.upload_scene(sd)
    t_upload = time.time() - t0
    info = be.bvh_info()

    W, H = args.width, args.height
    accum_t = None
    if use_dist:
        accum_t = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
        be.bind_accum(accum_t.data_ptr())
        be.set_stream(torch.cuda.current_stream().cuda_stream)
    if args.simulate_shard:
        r_, n_ = [int(v) for v in args.simulate_shard.split("/")]
        be.set_shard(r_, n_, 32)
    else:
        be.set_shard(rank, world, 32)
    be.resize(W, H)
    prm = art.Backend.pass_params(art.PT_MIS, True, 8, args.vthreads, seed=1)
    spp = 0
    for _ in range(args.warmup):
        spp = be.render_pass_device(prm, spp)

    def sync():
        be.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()

    sync()
    s0 = be.stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if args.host_buffers:
            prm.layout = art.LAYOUT_ADA_XY
            _, _, spp = be.render_pass(prm, spp, True, True)
        else:
            spp = be.render_pass_device(prm, spp)
    if use_dist:
        be.synchronize()
        dist.reduce(accum_t, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
    be.synchronize()
    elapsed = time.perf_counter() - t0
    s1 = be.stats()
    rays = s1.rays - s0.rays
    samples = s1.samples - s0.samples
    trace_ms = s1.trace_ms - s0.trace_ms
    launches = s1.trace_launches - s0.trace_launches
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rs = torch.tensor([float(rays), float(samples)], dtype=torch.float64, device="cuda")
        dist.all_reduce(rs, op=dist.ReduceOp.SUM)
        total_rays, total_samples = float(rs[0].item()), float(rs[1].item())
    else:
        total_rays, total_samples = float(rays), float(samples)

    if rank == 0:
        # ---- roofline of the dominant kernel (trace): algorithmic bytes / HIP-event time, this rank's launches
        roofline = None
        if not args.no_counters:
            be.set_option("count_tests", 1)
            c0 = be.stats()
            be.render_pass_device(art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=1), spp)   # untimed, counting variant
            c1 = be.stats()
            be.set_option("count_tests", 0)
            n = max(1, c1.traced_rays - c0.traced_rays)
            B = (c1.box_tests - c0.box_tests) / n
            T = (c1.tri_tests - c0.tri_tests) / n
            NV = (c1.node_visits - c0.node_visits) / n
            LV = (c1.leaf_visits - c0.leaf_visits) / n
            itn = max(1, c1.node_phase_iters - c0.node_phase_iters); itl = max(1, c1.leaf_phase_iters - c0.leaf_phase_iters)
            occ = {"groups_per_node_phase": round((c1.node_visits - c0.node_visits) / itn, 2),
                   "groups_per_leaf_phase": round((c1.leaf_visits - c0.leaf_visits) / itl, 2),
                   "tris_per_leaf_visit": round((c1.tri_tests - c0.tri_tests) / max(1, c1.leaf_visits - c0.leaf_visits), 2),
                   "node_phase_iters": int(itn), "leaf_phase_iters": int(itl), "wave_iters": int(c1.wave_iters - c0.wave_iters)}
            bytes_per_ray = 32.0 * B + 48.0 * T + 64.0
            achieved = rays * bytes_per_ray / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
            roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": measured_traffic(args, W, H),
                        "kernel": "k_trace_coop" if args.kernel == "coop" else "k_trace_simple",
                        "bytes_per_ray": round(bytes_per_ray, 1), "box_tests_per_ray": round(B, 2), "tri_tests_per_ray": round(T, 2),
                        "node_visits_per_ray": round(NV, 2), "leaf_visits_per_ray": round(LV, 2), "wave_occupancy": occ,
                        "avg_launch_ms": round(trace_ms / max(1, launches), 4), "launches": int(launches),
                        "trace_Mrays_per_s": round(rays / (trace_ms * 1e-3) / 1e6, 2) if trace_ms > 0 else None}
        cpu = None
        if not args.no_cpu and world == 1:          # timed on rank 0 at N = 1 only
            cpu = cpu_baseline(art, sd, args)
        value = total_rays / elapsed / 1e6
        line = {
            "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(1, args.steps), 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, PT_MIS depth 8, 2x2 AA, %d spp per step" % (scene_name, W, H, 4 * args.vthreads),
                       "spp_per_step": 4 * args.vthreads, "rays_per_sample": round(total_rays / max(1.0, total_samples), 3),
                       "Msamples_per_s": round(total_samples / elapsed / 1e6, 3), "parallelism": "pixel-tiles x%d" % world,
                       "bvh_width": info.node_width, "bvh_nodes": info.n_nodes, "bvh_build_ms": round(info.build_ms, 1), "bvh_max_stack": info.max_stack, "scene_gen_s": round(t_gen, 2),
                       "scene_upload_s": round(t_upload, 2)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    be.shutdown()


if __name__ == "__main__":
    main()

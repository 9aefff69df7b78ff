#!/usr/bin/env python3
"""bench.py -- Mrays/s of the render loop on MI355X (BASELINE.json metric), one JSON line on stdout.

Workload (N=1 and N>1): BASELINE config C4 -- synthetic 1M random triangles + 3 sphere lights inside the Cornell
box (SURVEY 8d generator, seed 0xADA5EED0+4), 1920x1080, PT_MIS, Max_Trace_Depth 8, 2x2 AA.  One "step" is one
Render_Pass of `--vthreads` x 4 samples per pixel: by default Threads_Num = 64, i.e. ONE STEP IS THE WHOLE 256-spp RENDER OF C4
(round 3; rounds 1-2 cut it into four 64-spp passes.  On one GPU nothing changes -- a pass is run in batches of at most 128 M
paths = 64 spp of the full frame either way -- but a GPU that owns 1/8 of the pixels now gets 66 M-path batches instead of 17 M-path
ones, so its late bounces still fill the chip).
Rays = closest-hit queries actually issued (camera + bounce + shadow).  Scene upload and BVH build are
outside the timed region and reported separately in `config`.

N > 1, two ways to drive the same sharding (interleaved 32x32 pixel tiles, strong scaling of the fixed frame, ONE RCCL
reduce(sum) of the float3 framebuffer to rank / device 0 inside the timed region):
  * launched by torchrun (WORLD_SIZE > 1): one process per GPU, torch.distributed (backend "nccl" = RCCL) does the reduce;
  * plain `python bench.py --gpus N`: ONE process, art_init_devices(N) -- the library shards and reduces by itself
    (what the single-task Ada host gets).

Extra objects on the line:
  "roofline"      the trace kernel against the memory roofline it is bound by.  achieved = rays x algorithmic bytes per ray /
                  trace-kernel time (HIP events around every launch, on the launch stream); algorithmic bytes per ray =
                  node_bytes x node visits + 48 x triangle tests + 64, the visits and tests counted on this very workload by
                  the counting variant of the kernel (equal to the oracle's walk of the exported tree); peak = 8 TB/s HBM3E.
                  "traffic" = bytes that actually left L2 per second (rocprofv3 FETCH_SIZE + WRITE_SIZE passes, FETCH_SIZE
                  times the factor calibrated on the kernel's access shape -- x1 for the trace kernel's 64-byte gathers, profiles/r5_calib),
                  only when the committed profile was taken on this source, scene and options; else null.
  "cpu_baseline"  the CPU oracle (kind "port": the reference is Ada and cannot be built or shipped) on a bounded sample of
                  the same workload, timed on this host; for --scene c2 also "mode_a", the reference-faithful organisation
                  (brute-force mesh, Threads_Num = 28 whole-frame tasks).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
GATHER_CEILING_GBPS = 8000.0   # MI355X_MICROARCH.md "Indexed rows": random rows out of the Infinity Cache, 8.6 TB/s (38 MB table) .. 7.4-7.9 (151 MB)
PROFILE_TAG = "r6_final"

# Algorithmic bytes of the wavefront stages per work item (the layout of csrc/art_scene.h HotField).  in: the
# item's hit record 16 + flags, shadow word, shadow epsilon, previous pdf, slot 20 + extension ray 24 (bounce 0: the hit record alone, the
# rest is recomputed) + the 64-byte shading record of the triangle it hit; fold record 16 per input item.  out, per kept item: two 64-byte
# trace records, hit record 16, shadow word 4, ray 24, flags / pdf / epsilon / slot 16, explicit colour 12.
STAGE_IN_B, STAGE_IN_B0, STAGE_GATHER_B, STAGE_FOLDREC_B, STAGE_OUT_B = 60.0, 16.0, 64.0, 16.0, 200.0
FOLD_B = 56.0            # k_fold_level per item of a level: w 12 + child 4 + e 12 + the level below 12 + its child word 4, writes 12
RAYGEN_B = 80.0          # per camera path: trace record 64 + hit record 16
ACCUM_B = 12.0           # k_accumulate reads the per-sample radiance


def build_scene(art, args):
    from ada_ray_tracer_amd import scenes
    if args.scene == "c4":
        return scenes.synthetic_scene(args.tris, 4), "C4: synthetic %d triangles + 3 sphere lights in the Cornell box" % args.tris
    if args.scene == "c3":
        return scenes.synthetic_scene(100000, 3), "C3: synthetic 100000 triangles + 3 sphere lights in the Cornell box"
    if args.scene == "c5":
        return scenes.mixed_scene(20000, 5), "C5: spheres + 20000-triangle mesh, glass/diffuse/emissive"
    if args.scene == "s4":
        sd = scenes.structured_scene(args.tris)
        return sd, "S4 (not a BASELINE config): structured meshes, %d triangles (tessellated torus + regular grid) + 3 sphere lights in the Cornell box" % sd.desc.meshes[0].ntris
    if args.scene == "i64":
        sd = scenes.instanced_scene(64, 20000)
        return sd, "I64 (not a BASELINE config): 64 instances of two ~20000-triangle meshes (1.28 M triangles if flattened) + 3 sphere lights in the Cornell box, rendered through the two-level tree"
    if args.scene == "c1":
        return scenes.eight_sphere_scene(), "C1: Cornell-box-style 8-sphere scene"
    return scenes.reference_scene(), "C2: internal Cornell scene with data/pyramid2.vsgf (Scene.Init by the product's host layer)"


def source_fingerprint():
    """sha256 over the kernel / driver sources: a committed PMC profile only describes the build it was taken on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ada-ray-tracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def workload_fingerprint(args, W, H, info, opts):
    return {"scene": args.scene, "tris": args.tris if args.scene == "c4" else None, "width": W, "height": H, "spp_per_step": 4 * args.vthreads,
            "kernel": args.kernel, "bvh_width": info.node_width, "bvh_nodes": info.n_nodes, "options": sorted(opts), "source": source_fingerprint()}


def profiled(args, fp):
    """The committed rocprofv3 PMC summary (profiles/collect.sh -> profiles/summarize.py) if it was taken on exactly this source,
    scene and options; None otherwise -- a number measured on another build must not sit next to this run's."""
    try:
        tag = PROFILE_TAG if args.scene == "c4" else PROFILE_TAG.split("_")[0] + "_" + args.scene      # r6_final for the default workload, r6_c3 / r6_c5 / r6_s4 for the others
        d = json.load(open(os.path.join(ROOT, "profiles", tag, "pmc_summary.json")))
        if d.get("fingerprint") != fp:
            return None
        return d["k_trace_coop"]
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


def cpu_baseline(art, sd, args, be):
    """CPU oracle (kind "port") on a bounded sample of the same workload: same scene, same camera, reduced frame (ray
    distribution preserved), PT_MIS depth 8, all host threads of the job's share.  The closest-hit mesh search walks the product's
    exported BVH (oracle/art_oracle.c: intersect_mesh_closest) -- the reference's own O(N) scan is timed as mode_a where feasible."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conv
    import orc
    w, h = args.cpu_width, args.cpu_height
    osc = conv.OracleScene(sd)
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
    out = {"value": round(cnt_total / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "sample": "same scene+camera at %dx%d, %d passes x 4 spp, PT_MIS depth 8, BVH closest-hit, %.1f s" % (w, h, passes, dt),
           "mode_a": None}
    if args.scene in ("c1", "c2"):
        # BASELINE.md section 3, mode A: what the Ada program does -- brute-force mesh scan (IntersectMeshBF), Threads_Num = 28 tasks
        # that each render the whole frame (Path_Trace_Thread), private frames instead of the global GNAT.Task_Lock
        osa = conv.OracleScene(sd)                                  # no BVH attached: geometry.adb:266-323 order and cost
        pa = orc.make_params(w, h, orc.PT_MIS, True, 8, 28, seed=1, nthreads=cores)
        t0 = time.time()
        n_pass = 0; rays_a = 0
        while True:
            _, _, cnt = orc.render_tasks(osa.scene, pa)
            rays_a += cnt.rays; n_pass += 1
            if time.time() - t0 > args.cpu_seconds or n_pass >= 16:
                break
        dta = time.time() - t0
        out["mode_a"] = {"value": round(rays_a / dta / 1e6, 4), "unit": "Mrays/s", "cores": cores, "threads_num": 28,
                         "sample": "Threads_Num = 28 whole-frame tasks (112 spp per pass) on %d OS threads, %dx%d, %d passes, brute-force mesh, "
                                   "private frames instead of GNAT.Task_Lock, %.1f s" % (cores, w, h, n_pass, dta)}
    else:
        out["mode_a_note"] = "the reference's O(N) mesh scan is ~8 ms per ray at 1M triangles on 8 cores: not timed here; --scene c2 reports mode_a"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="c4", choices=["c1", "c2", "c3", "c4", "c5", "s4", "i64"])
    ap.add_argument("--tris", type=int, default=1000000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--vthreads", type=int, default=64, help="Threads_Num of the pass: spp per step = 4 * vthreads (default 64 -> 256 spp: one step = the whole C4 render)")
    ap.add_argument("--kernel", default="coop", choices=["coop", "simple"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the cpu_baseline leg (0 = the job's CPU share)")
    ap.add_argument("--cpu-width", type=int, default=480)
    ap.add_argument("--cpu-height", type=int, default=270)
    ap.add_argument("--no-counters", action="store_true", help="skip the untimed B/T counting pass")
    ap.add_argument("--host-buffers", action="store_true", help="hand AccumBuff/screen_buffer over in host memory every pass (Ada layout), i.e. include PCIe")
    ap.add_argument("--simulate-shard", default="", help="R/N: render only rank R's pixel tiles of an N-GPU job on this one GPU (scaling rehearsal)")
    ap.add_argument("--contexts", type=int, default=0, help="rehearse the one-process N-device path with N contexts on GPU 0 (a 1-GPU box)")
    ap.add_argument("--opt", action="append", default=[], help="backend option name=value (art_set_option), e.g. bvh_leaf_base_milli=1000")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    use_dist = world > 1 or bool(os.environ.get("ART_BENCH_FORCE_DIST"))     # FORCE: rehearse the torch/RCCL path with one rank
    in_library = (not use_dist) and (args.gpus > 1 or args.contexts > 1)       # one process drives all the GPUs through art_init_devices
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    art = ge.load_package()
    if in_library:
        be = art.Backend(devices=([0] * args.contexts) if args.contexts > 1 else args.gpus)
        n_gpus = args.contexts if args.contexts > 1 else args.gpus
    else:
        be = art.Backend(local_rank if use_dist else 0)
        n_gpus = world
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
    if not in_library:
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
    g0 = be.stage_stats()
    r0 = be.reduce_info() if in_library else None
    reduce_ms_torch = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if args.host_buffers:
            prm.layout = art.LAYOUT_ADA_XY
            _, _, spp = be.render_pass(prm, spp, True, True)
        else:
            spp = be.render_pass_device(prm, spp)
    if use_dist:
        be.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dist.reduce(accum_t, dst=0, op=dist.ReduceOp.SUM)
        e1.record()
        torch.cuda.synchronize()
        reduce_ms_torch = float(e0.elapsed_time(e1))
    elif in_library:
        be.reduce()                       # the RCCL reduce of the float3 framebuffer to device 0, inside the timed region
    be.synchronize()
    elapsed = time.perf_counter() - t0
    s1 = be.stats()
    g1 = be.stage_stats()
    r1 = be.reduce_info() if in_library else None
    rays = s1.rays - s0.rays
    samples = s1.samples - s0.samples
    trace_ms = s1.trace_ms - s0.trace_ms
    launches = s1.trace_launches - s0.trace_launches
    if use_dist:
        elapsed_local = elapsed
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rs = torch.tensor([float(rays), float(samples)], dtype=torch.float64, device="cuda")
        dist.all_reduce(rs, op=dist.ReduceOp.SUM)
        total_rays, total_samples = float(rs[0].item()), float(rs[1].item())
        rays_dev0 = rays
        # every rank's own wall time of the timed steps, and the ranks the collective really ran on (the line must not be taken on trust)
        per_rank = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([elapsed_local], dtype=torch.float64, device="cuda"))
        multi = {"mode": "one process per GPU (torch.distributed)", "backend": dist.get_backend(), "rccl_ranks": int(dist.get_world_size()),
                 "reduce_ms": round(reduce_ms_torch, 3), "reduce_note": "torch.cuda events around dist.reduce on rank 0 (one reduce per run, inside the timed region)",
                 "per_device_ms_per_step": [round(float(t.item()) * 1e3 / max(1, args.steps), 3) for t in per_rank],
                 # what the slowest rank makes the others wait per step (tile-deal imbalance + start skew): max over ranks - own wall time
                 "per_device_idle_ms_per_step": [round((elapsed - float(t.item())) * 1e3 / max(1, args.steps), 3) for t in per_rank]}
    else:
        total_rays, total_samples = float(rays), float(samples)
        rays_dev0 = rays / n_gpus if in_library else rays          # stats(): trace_ms is device 0's, rays the whole job's
        multi = None
        if in_library:
            multi = {"mode": "one process (art_init_devices)", "backend": "rccl" if r1.rccl_ranks > 0 else "local adds (contexts on one GPU)",
                     "rccl_ranks": int(r1.rccl_ranks), "devices": int(r1.devices), "reduces": int(r1.reduces - r0.reduces),
                     "reduce_ms": round(r1.reduce_ms - r0.reduce_ms, 3), "reduce_note": "HIP events around the grouped ncclReduce on device 0's stream (inside the timed region)",
                     "per_device_ms_per_step": [round((r1.device_pass_ms[k] - r0.device_pass_ms[k]) / max(1, args.steps), 3) for k in range(int(r1.devices))],
                     # host-clock marks on every device's stream (hipLaunchHostFunc: one clock for all devices): busy = end - start of a pass,
                     # idle = the slowest device's end - this device's end (tile-deal imbalance), start_skew = this device's start - the first
                     # device's start (the host enqueues device after device); passes_overlapped = passes in which every device had started
                     # before any had finished (the host never waits between devices: ray_tracer.adb:271-277)
                     "per_device_busy_ms_per_step": [round((r1.device_busy_ms[k] - r0.device_busy_ms[k]) / max(1, args.steps), 3) for k in range(int(r1.devices))],
                     "per_device_idle_ms_per_step": [round((r1.device_idle_ms[k] - r0.device_idle_ms[k]) / max(1, args.steps), 3) for k in range(int(r1.devices))],
                     "per_device_start_skew_ms_per_step": [round((r1.device_start_skew_ms[k] - r0.device_start_skew_ms[k]) / max(1, args.steps), 3) for k in range(int(r1.devices))],
                     "passes": int(r1.passes - r0.passes), "passes_overlapped": int(r1.passes_overlapped - r0.passes_overlapped)}

    if rank == 0:
        # ---- roofline of the dominant kernel (trace): algorithmic bytes / HIP-event time, device 0's launches
        roofline = None
        if not args.no_counters and info.n_tris > 0:      # a scene without a BVH mesh never launches the trace kernel; one-process N-GPU mode: device 0's launches
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
            # algorithmic bytes per ray of THIS data layout: a node visit reads the node packet once (64 B quantised at
            # width 4, 256 B binary32 at width 8), a triangle test reads a 48-B record, a ray costs 32 B in + 32 B out
            node_bytes = 64.0 if info.node_width == 4 else 256.0
            if args.scene == "i64" and "inst_coop=0" in args.opt:
                node_bytes = 128.0                          # option inst_coop = 0: k_trace_inst walks binary32 4-wide packets (the cooperative kernel: 64-byte quantised nodes, as everywhere)
            bytes_per_ray = node_bytes * NV + 48.0 * T + 64.0
            achieved = rays_dev0 * bytes_per_ray / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
            fp = workload_fingerprint(args, W, H, info, args.opt)
            prof = None if in_library else profiled(args, fp)
            # What the kernel is limited by, from the committed counter passes of THIS build (profiles/summarize.py), next to the contract figure:
            #   issue       = fraction of the SIMDs' time spent issuing VALU instructions = SQ_INSTS_VALU per launch / 1024 SIMDs x the mean
            #                 issue time of the kernel's own instruction mix (profiles/valu_mix.py x profiles/valu_rate2.hip) / launch time
            #   fabric_frac = bytes that left L2 (traffic) / the guide's ceiling for random multi-line gathers out of the Infinity Cache
            #                 (8.6 TB/s for a 38 MB table, 7.4-7.9 for 151 MB; the 89 MB working set of C4 is priced at 8.0)
            issue = round(prof["valu_issue_frac"], 4) if prof and "valu_issue_frac" in prof else None
            traffic = prof.get("traffic_GBps_calibrated", prof.get("traffic_GBps_raw")) if prof else None      # FETCH_SIZE x the factor calibrated on the kernel's own access shape (x1: profiles/r5_calib)
            fabric = round(traffic / GATHER_CEILING_GBPS, 4) if prof else None
            over = achieved > HBM_PEAK_GBPS      # only the 8-wide option: its 256-byte binary32 nodes are mostly served by L2, so the
            cands = [("hbm", 0.0 if over else achieved / HBM_PEAK_GBPS)] + ([("issue", issue)] if issue is not None else []) + ([("fabric", fabric)] if fabric is not None else [])
            bound = max(cands, key=lambda kv: kv[1])[0] if prof else "hbm"
            roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",   # the contract's roofline of this path (HBM bandwidth, no MFMA); achieved = ALGORITHMIC bytes per second
                        "limited_by": bound,
                        "bound_note": "bound = the roofline the contract prices this path against (HBM bandwidth).  limited_by = the largest of frac (algorithmic bytes / HBM peak), issue (VALU issue time / SIMD time) and fabric_frac (bytes past L2 / gather ceiling) -- what the kernel itself sits on; 'hbm' when no profile of this build is committed",
                        "issue": issue, "fabric_frac": fabric,
                        "device": ("device 0 of %d (one process, art_init_devices); counters summed over the devices" % n_gpus) if in_library else None,
                        "frac": None if over else round(achieved / HBM_PEAK_GBPS, 4),
                        "frac_note": "algorithmic bytes exceed the HBM peak (cache hits): not a roofline fraction" if over else None,
                        "traffic": round(traffic, 1) if prof else None,
                        "traffic_note": "FETCH_SIZE x 1.0 + WRITE_SIZE per launch / launch time: random 64-byte node packets and triangle records are counted exactly (profiles/r5_calib/calibration; rounds 1-4 doubled FETCH_SIZE on a calibration of 128-byte gathers the kernel no longer does)" if prof else None,
                        "traffic_source": ("profiles/%s/pmc_summary.json (same source %s, scene, options)" % (PROFILE_TAG if args.scene == "c4" else PROFILE_TAG.split("_")[0] + "_" + args.scene, fp["source"])) if prof else None,
                        "kernel": ("k_trace_inst (two-level search, one ray per lane)" if "inst_coop=0" in args.opt else "k_trace_coop<.., INST> (two-level tree)") if args.scene == "i64" else "k_trace_coop" if args.kernel == "coop" else "k_trace_simple",
                        "bound_evidence": "frac = algorithmic bytes against the HBM peak (the contract's roofline).  What limits the kernel itself is VALU issue: "
                                          "issue = SQ_INSTS_VALU per launch / 1024 SIMDs x the mean issue time of the kernel's own instruction mix (profiles/valu_mix.py x "
                                          "profiles/valu_rate2.hip) / launch time; fabric_frac = bytes past L2 / 8.0 TB/s gather ceiling; both from profiles/%s/pmc_summary.json "
                                          "when its fingerprint equals this run's.  Sensitivity builds: profiles/r2_sensitivity.json" % PROFILE_TAG,
                        "bytes_per_ray": round(bytes_per_ray, 1), "node_bytes": node_bytes, "box_tests_per_ray": round(B, 2), "tri_tests_per_ray": round(T, 2),
                        "node_visits_per_ray": round(NV, 2), "leaf_visits_per_ray": round(LV, 2),
                        "survey_8d_formula_GBps": round(rays_dev0 * (32.0 * B + 48.0 * T + 64.0) / (trace_ms * 1e-3) / 1e9, 1) if trace_ms > 0 else None,
                        "l2_hit_rate": round(prof["l2_hit_rate"], 4) if prof and "l2_hit_rate" in prof else None,
                        "wave_occupancy": occ,
                        # the stages around the trace kernel: whole job over trace-only = trace-kernel time / wall time of the timed steps
                        "whole_over_trace_only": round((trace_ms * 1e-3) / elapsed, 4) if (elapsed > 0 and not in_library) else None,
                        "avg_launch_ms": round(trace_ms / max(1, launches), 4), "launches": int(launches),
                        "trace_Mrays_per_s": round(rays_dev0 / (trace_ms * 1e-3) / 1e6, 2) if trace_ms > 0 else None}
        # ---- the stages around the trace kernel and the whole-job roofline (device 0 / this rank; cumulative counters of the timed steps)
        stages = None
        whole = None
        if args.kernel == "coop" and g1.batches > g0.batches:
            nb = g1.batches - g0.batches
            ins = [g1.items_in[k] - g0.items_in[k] for k in range(16)]; outs = [g1.items_out[k] - g0.items_out[k] for k in range(16)]
            depth = max(k + 1 for k in range(16) if ins[k] > 0)
            shade_b = sum(ins[k] * ((STAGE_IN_B0 if k == 0 else STAGE_IN_B) + STAGE_GATHER_B + STAGE_FOLDREC_B) + outs[k] * STAGE_OUT_B for k in range(depth))
            fold_b = sum(ins[k] * FOLD_B for k in range(depth)) + ins[0] * ACCUM_B
            raygen_b = ins[0] * RAYGEN_B
            shade_ms, fold_ms, raygen_ms = g1.shade_ms - g0.shade_ms, g1.fold_ms - g0.fold_ms, g1.raygen_ms - g0.raygen_ms
            def leg(b, ms):
                return {"ms_per_batch": round(ms / nb, 3), "algorithmic_GB_per_batch": round(b / nb / 1e9, 3), "achieved_GBps": round(b / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                        "frac_of_hbm_peak": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if ms > 0 else None}
            stages = {"batches": int(nb), "items_in_per_bounce": [int(v // nb) for v in ins[:depth]], "items_kept_per_bounce": [int(v // nb) for v in outs[:depth]],
                      "bytes_per_item": {"in": STAGE_IN_B, "in_bounce0": STAGE_IN_B0, "shading_record_gather": STAGE_GATHER_B, "fold_record": STAGE_FOLDREC_B, "out_per_kept_item": STAGE_OUT_B},
                      "shade": leg(shade_b, shade_ms), "fold_accumulate": leg(fold_b, fold_ms), "raygen": leg(raygen_b, raygen_ms),
                      "note": "GPU ms: HIP events on the launch stream around every launch of the kind; bytes: algorithmic (what the layout makes a stage read and write), "
                              "not counters -- profiles/%s holds the counter passes (FETCH_SIZE x the factor calibrated per access shape, profiles/r5_calib)" % PROFILE_TAG}
            if roofline is not None and elapsed > 0:
                trace_b = rays_dev0 * roofline["bytes_per_ray"]
                total_b = trace_b + shade_b + fold_b + raygen_b
                whole = {"algorithmic_GB_per_step": round(total_b / max(1, args.steps) / 1e9, 2), "achieved_GBps": round(total_b / elapsed / 1e9, 1),
                         "frac": round(total_b / elapsed / 1e9 / HBM_PEAK_GBPS, 4), "peak": HBM_PEAK_GBPS,
                         "shares_of_gpu_time": {"trace": round(trace_ms * 1e-3 / elapsed, 4), "shade": round(shade_ms * 1e-3 / elapsed, 4),
                                                "fold_accumulate": round(fold_ms * 1e-3 / elapsed, 4), "raygen": round(raygen_ms * 1e-3 / elapsed, 4)},
                         "note": "every kernel's algorithmic bytes of the timed steps / the driver-visible wall time / HBM peak"}
        cpu = None
        if not args.no_cpu and n_gpus == 1 and args.scene != "i64":          # timed on rank 0 at N = 1 only
            cpu = cpu_baseline(art, sd, args, be)
        elif args.scene == "i64":
            cpu = {"value": None, "note": "the oracle has no instancing: tests/test_gpu_instanced.py compares with its render of the flattened scene"}
        elif n_gpus > 1:
            cpu = {"value": None, "note": "the CPU leg is timed at N = 1 only (bench contract); see the N = 1 line"}
        value = total_rays / elapsed / 1e6
        spp_step = 4 * args.vthreads
        render_256_s = elapsed / max(1, args.steps) * (256.0 / spp_step)
        line = {
            "metric": "Mrays/s", "value": round(value, 3), "unit": "Mrays/s", "n_gpus": (1 if args.contexts > 1 else n_gpus), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(1, args.steps), 3), "higher_is_better": True, "scaling": "strong",
            "trace_ms_per_step": round(trace_ms / max(1, args.steps), 3),        # HIP events around every trace launch (device 0 / this rank)
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, PT_MIS depth 8, 2x2 AA, %d spp per step" % (scene_name, W, H, 4 * args.vthreads),
                       "spp_per_step": 4 * args.vthreads, "rays_per_sample": round(total_rays / max(1.0, total_samples), 3),
                       "Msamples_per_s": round(total_samples / elapsed / 1e6, 3),
                       "parallelism": ("pixel-tiles x%d, %s" % (n_gpus, "one process (art_init_devices)" if in_library else "one process per GPU (torch.distributed)")) if n_gpus > 1 else "pixel-tiles x1",
                       "contexts_on_one_gpu": args.contexts if args.contexts > 1 else None,
                       "bvh_width": info.node_width, "bvh_nodes": info.n_nodes, "bvh_build_ms": round(info.build_ms, 1), "bvh_max_stack": info.max_stack, "scene_gen_s": round(t_gen, 2),
                       "scene_upload_s": round(t_upload, 3),
                       # what a caller waits for: scene upload incl. the BVH build (outside the timed region) + the 256-spp render at the measured rate
                       "end_to_end_s": round(t_upload + render_256_s, 3), "render_256spp_s": round(render_256_s, 3), "fingerprint": workload_fingerprint(args, W, H, info, args.opt)},
            "roofline": roofline, "whole_job_roofline": whole, "stages": stages, "multi_gpu": multi, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    be.shutdown()


if __name__ == "__main__":
    main()

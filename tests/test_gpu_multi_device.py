"""SURVEY 8(e) inside the C ABI: one process, n devices (art_init_devices).  A GPU box of this pool has one GPU, so the n-device path is
rehearsed with n contexts on that GPU (same ordinal repeated: per-context streams, replicated scene, tile ownership, path buffers and
the framebuffer sum are all the real thing; only the RCCL call itself is replaced by a local sum, because a communicator cannot hold
one GPU twice).  Runs in a child process: the library is a process-wide singleton and the session's backend fixture owns it here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import __graft_entry__ as ge
art = ge.load_package()
from ada_ray_tracer_amd import scenes
import conv, orc
out = {}
sd = scenes.synthetic_scene(3000, 3)
p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=5)
imgs = {}
for name, devs in (("one", None), ("init_devices_1", [0]), ("three_contexts", [0, 0, 0]), ("eight_contexts", [0] * 8)):
    be = art.Backend(0) if devs is None else art.Backend(devices=devs)
    if name == "three_contexts":
        be.set_option("bvh_builder", 1)          # every context builds its own LBVH tree
    be.upload_scene(sd)
    be.resize(100, 72)
    spp = be.render_pass_device(p, 0)
    accum, screen, spp = be.render_pass(p, spp, True, True)       # second pass: per-device buffers keep accumulating, then the reduce
    st = be.stats()
    dbg = be.debug_hit_pass(art.Backend.pass_params(art.RT_DEBUG, False, 8, 1))
    imgs[name] = (accum.copy(), screen.copy(), spp, st.rays, st.samples, dbg[2].copy())
    if devs is not None and len(devs) > 1:
        try:
            be.set_shard(0, 2, 32); out[name + "_set_shard_refused"] = False
        except art.ArtError:
            out[name + "_set_shard_refused"] = True
    be.shutdown()
ref, rspp, cnt = orc.render(conv.OracleScene(sd).scene, orc.make_params(100, 72, orc.PT_MIS, True, 8, 2, seed=5), passes=2)
base = imgs["one"]
out["oracle_equal"] = bool(np.array_equal(base[0].view(np.uint32), ref.view(np.uint32))) and base[2] == rspp and base[3] == cnt.rays
for name, v in imgs.items():
    out[name] = bool(np.array_equal(v[0].view(np.uint32), base[0].view(np.uint32)) and np.array_equal(v[1], base[1]) and v[2:5] == base[2:5]
                     and np.array_equal(v[5], base[5]))
# an instanced scene (the two-level tree replicated per context) through the same path
isd = scenes.instanced_scene(6, 200)
iimgs = {}
for name, devs in (("one", None), ("three_contexts", [0, 0, 0])):
    be = art.Backend(0) if devs is None else art.Backend(devices=devs)
    be.upload_scene(isd); be.resize(100, 72)
    accum, screen, spp = be.render_pass(p, 0, True, True)
    iimgs[name] = (accum.copy(), screen.copy(), spp, be.stats().rays)
    be.shutdown()
out["instanced_three_contexts"] = bool(np.array_equal(iimgs["one"][0].view(np.uint32), iimgs["three_contexts"][0].view(np.uint32))
                                       and np.array_equal(iimgs["one"][1], iimgs["three_contexts"][1]) and iimgs["one"][2:] == iimgs["three_contexts"][2:]
                                       and float(np.abs(iimgs["one"][0]).sum()) > 0.0)
# Round 6: the host only enqueues -- no device's pass may wait for another device's (until round 5 the items-per-thread trial of the
# shade stage blocked the host twice per trial batch, so the first batches of the devices ran one device at a time).  Three contexts, three
# passes of two batches each enqueued back to back: art_get_reduce_info's host-clock marks (hipLaunchHostFunc on every stream) must show
# every device started before any finished, in every pass -- Render_Pass releases all its workers before it waits for one
# (ray_tracer.adb:271-277).
be = art.Backend(devices=[0, 0, 0])
be.set_option("batch_paths", 1 << 21)
be.upload_scene(scenes.synthetic_scene(20000, 3)); be.resize(1280, 720)
p4 = art.Backend.pass_params(art.PT_MIS, True, 8, 4, seed=5)
spp = 0
for _ in range(3):
    spp = be.render_pass_device(p4, spp)
ri = be.reduce_info()
out["enqueue"] = {"passes": ri.passes, "overlapped": ri.passes_overlapped, "busy_positive": all(ri.device_busy_ms[k] > 0.0 for k in range(3)),
                  "idle_nonnegative": all(ri.device_idle_ms[k] >= 0.0 for k in range(3)), "one_device_never_idle": min(ri.device_idle_ms[k] for k in range(3)) == 0.0 or ri.passes > 1,
                  "skew_below_busy": all(ri.device_start_skew_ms[k] < ri.device_busy_ms[k] for k in range(3)), "first_device_skew_zero": min(ri.device_start_skew_ms[k] for k in range(3)) == 0.0 or ri.passes > 1}
out["enqueue_ms"] = {"busy": list(ri.device_busy_ms)[:3], "idle": list(ri.device_idle_ms)[:3], "skew": list(ri.device_start_skew_ms)[:3]}
be.shutdown()
print(json.dumps(out))
'''


def test_n_contexts_in_one_process_give_the_single_device_image(art):
    r = subprocess.run([sys.executable, "-c", SCRIPT, art.ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    ms = out.pop("enqueue_ms")
    assert out.pop("enqueue") == {"passes": 3, "overlapped": 3, "busy_positive": True, "idle_nonnegative": True, "one_device_never_idle": True,
                                  "skew_below_busy": True, "first_device_skew_zero": True}, ms
    assert out == {"oracle_equal": True, "one": True, "init_devices_1": True, "three_contexts": True, "eight_contexts": True,
                   "three_contexts_set_shard_refused": True, "eight_contexts_set_shard_refused": True, "instanced_three_contexts": True}, out


def test_rccl_calls_of_the_n_device_path_run_on_one_device(art):
    """ART_FORCE_RCCL=1: art_init_devices(1) builds the communicator (ncclCommInitAll) and the framebuffer goes through the grouped
    ncclReduce into device 0's reduce buffer -- the collective of SURVEY 8(e) with one rank.  Same image as without it."""
    script = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import __graft_entry__ as ge
art = ge.load_package()
from ada_ray_tracer_amd import scenes
sd = scenes.synthetic_scene(2000, 3)
p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=9)
out = []; info = []
for force in ("0", "1"):
    os.environ["ART_FORCE_RCCL"] = force
    be = art.Backend(devices=[0])
    be.upload_scene(sd); be.resize(64, 48)
    accum, screen, spp = be.render_pass(p, 0, True, True)
    be.reduce(); be.synchronize()
    ri = be.reduce_info()
    info.append({"devices": ri.devices, "rccl_ranks": ri.rccl_ranks, "path": ri.path, "reduces_at_least_one": ri.reduces >= 1,
                 "reduce_ms_positive": ri.reduce_ms > 0.0, "device0_pass_ms_positive": ri.device_pass_ms[0] > 0.0,
                 "passes": ri.passes, "overlapped": ri.passes_overlapped, "busy_positive": ri.device_busy_ms[0] > 0.0, "idle": ri.device_idle_ms[0], "skew": ri.device_start_skew_ms[0]})
    out.append((accum.copy(), screen.copy()))
    be.shutdown()
print(json.dumps({"same": bool(np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and np.array_equal(out[0][1], out[1][1])), "nonzero": bool(out[1][0].any()), "info": info}))
'''
    r = subprocess.run([sys.executable, "-c", script, art.ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["same"] and got["nonzero"]
    # art_get_reduce_info: without the communicator nothing is reduced (one device); with it the reduce ran on a 1-rank RCCL communicator,
    # took GPU time, and the device's passes were timed -- the fields bench.py --gpus N puts on its line (multi_gpu)
    assert got["info"][0] == {"devices": 1, "rccl_ranks": 0, "path": 0, "reduces_at_least_one": False, "reduce_ms_positive": False, "device0_pass_ms_positive": True,
                              "passes": 1, "overlapped": 1, "busy_positive": True, "idle": 0.0, "skew": 0.0}
    assert got["info"][1] == {"devices": 1, "rccl_ranks": 1, "path": 1, "reduces_at_least_one": True, "reduce_ms_positive": True, "device0_pass_ms_positive": True,
                              "passes": 1, "overlapped": 1, "busy_positive": True, "idle": 0.0, "skew": 0.0}


def test_bench_under_torchrun_with_one_rank(art):
    """The driver's launch line for N > 1 (python -m torch.distributed.run ... bench.py --gpus N) with one rank: torch.distributed over
    RCCL, device_id passed to init_process_group, shard + bound accum + dist.reduce."""
    env = dict(os.environ, ART_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29517",
                        os.path.join(art.ROOT, "bench.py"), "--gpus", "1", "--scene", "c3", "--width", "256", "--height", "144", "--steps", "1", "--warmup", "1",
                        "--vthreads", "1", "--no-cpu"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["metric"] == "Mrays/s" and line["value"] > 0 and line["n_gpus"] == 1
    m = line["multi_gpu"]       # the run shows for itself which collective it used, on how many ranks, and what it cost
    assert m["backend"] == "nccl" and m["rccl_ranks"] == 1 and m["reduce_ms"] >= 0.0 and len(m["per_device_ms_per_step"]) == 1 and m["per_device_ms_per_step"][0] > 0
    assert line["stages"]["batches"] >= 1 and line["stages"]["shade"]["ms_per_batch"] > 0

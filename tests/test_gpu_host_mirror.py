"""GPU: the C++ replay of test.adb (host/test_main.cpp) writes the BMP the oracle predicts; bench.py keeps its contract."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def test_test_adb_replay_writes_identical_bmp(art, tmp_path):
    exe = os.path.join(art.PKG_DIR, "art_test")
    out = str(tmp_path / "ART_render.bmp")
    # width height passes Threads_Num render_type(PT_MIS) aa
    r = subprocess.run([exe, orc.PYRAMID_VSGF, out, "96", "64", "2", "3", "4", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "spp =  24" in r.stdout.replace("spp = 24", "spp =  24")
    cs = orc.CornellScene()
    acc, spp, _ = orc.render(cs.scene, orc.make_params(96, 64, orc.PT_MIS, True, 8, 3, seed=1), passes=2)
    assert open(out, "rb").read() == orc.bmp_bytes(orc.resolve(acc, spp))
    r = subprocess.run([exe, orc.PYRAMID_VSGF, out, "64", "64", "5", "28", "0", "1"], capture_output=True, text=True, timeout=300)   # RT_DEBUG finishes after one pass
    assert r.returncode == 0 and r.stdout.count("pass ") == 1
    oacc, _, _, _ = orc.debug_pass(cs.scene, orc.make_params(64, 64, orc.RT_DEBUG, False))
    assert open(out, "rb").read() == orc.bmp_bytes(orc.resolve(oacc, 1))


def test_test_adb_replay_over_a_hydra_scene_library_writes_the_flattened_scenes_bmp(art, tmp_path):
    """the reference's SCN = "external_cpp" build of the same driver (art.gpr:6-14): Scene.Init reads a Hydra scene library; here its meshes and
    <instance>s go through art_upload_scene as instances (host/hydra_scene.cpp Build_Render_Desc) and the BMP is the one the oracle predicts
    for the explicitly flattened scene"""
    import conv
    from test_hydra_scene import SCENE_DIR, _hydra_render_scene
    exe = os.path.join(art.PKG_DIR, "art_test")
    out = str(tmp_path / "ART_hydra.bmp")
    r = subprocess.run([exe, orc.PYRAMID_VSGF, out, "96", "64", "2", "3", "4", "1", SCENE_DIR], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    _, flat = _hydra_render_scene(art)
    acc, spp, _ = orc.render(conv.OracleScene(flat).scene, orc.make_params(96, 64, orc.PT_MIS, True, 8, 3, seed=1), passes=2)
    assert spp == 24 and open(out, "rb").read() == orc.bmp_bytes(orc.resolve(acc, spp))


def test_python_mirror_of_ray_tracer_package(art, backend):
    """test.adb:32-69 through the Python mirror of package Ray_Tracer: Ada (x,y) buffers, screen copy, SaveBMP."""
    import conv
    cs = orc.CornellScene()
    rt = art.RayTracer(backend, conv.desc_from_oracle(art, cs))
    rt.Threads_Num = 2
    rt.Init_Render(art.PT_MIS)
    rt.Resize_Viewport(72, 48)
    for _ in range(2):
        rt.Render_Pass()
    assert rt.GetSPP() == 16 and not rt.Finished()
    assert rt.screen_buffer.shape == (72, 48) and rt.g_accBuff.shape == (72, 48, 3)
    image = np.zeros((48, 72), np.uint32)
    for y in range(48):
        for x in range(72):
            image[y, x] = rt.screen_buffer[x, y]                       # test.adb:63-67
    acc, spp, _ = orc.render(cs.scene, orc.make_params(72, 48, orc.PT_MIS, True, 8, 2, seed=1), passes=2)
    assert art.save_bmp(None, image) == orc.bmp_bytes(orc.resolve(acc, spp))
    rt.Init_Render(art.RT_DEBUG)
    rt.Render_Pass()
    assert rt.Finished()


def test_bench_contract(art):
    r = subprocess.run([sys.executable, os.path.join(art.ROOT, "bench.py"), "--scene", "c3", "--width", "256", "--height", "144", "--steps", "1",
                        "--warmup", "1", "--vthreads", "1", "--cpu-seconds", "1", "--cpu-width", "64", "--cpu-height", "36"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in line
    assert line["metric"] == "Mrays/s" and line["value"] > 0 and line["vs_baseline"] is None and "workload" in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["frac"] <= 1.0 and rf["traffic"] is None             # a reduced frame never matches the committed profile's fingerprint
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["mode_a"] is None


def test_bench_c2_comes_from_the_product_and_reports_mode_a(art):
    """--scene c2: Scene.Init by the product's host layer (no oracle in the workload), and the reference-faithful CPU organisation
    (brute-force mesh, Threads_Num = 28 whole-frame tasks) next to the pixel-parallel port."""
    r = subprocess.run([sys.executable, os.path.join(art.ROOT, "bench.py"), "--scene", "c2", "--width", "128", "--height", "128", "--steps", "1",
                        "--warmup", "1", "--vthreads", "1", "--cpu-seconds", "1", "--cpu-width", "48", "--cpu-height", "48"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert "Scene.Init by the product" in line["config"]["workload"] and line["value"] > 0
    a = line["cpu_baseline"]["mode_a"]
    assert a["threads_num"] == 28 and a["value"] > 0 and "brute-force" in a["sample"]


def test_bench_one_process_many_contexts(art):
    """bench.py --contexts 4: the one-process N-device path of the library (art_init_devices) rehearsed on this box's single GPU."""
    r = subprocess.run([sys.executable, os.path.join(art.ROOT, "bench.py"), "--scene", "c3", "--width", "256", "--height", "144", "--steps", "1",
                        "--warmup", "1", "--vthreads", "1", "--contexts", "4", "--no-cpu"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    # four contexts on ONE physical GPU are reported as that: n_gpus 1, contexts_on_one_gpu 4; the roofline object is
    # there in this mode too (device 0's launches), not null
    assert line["n_gpus"] == 1 and line["config"]["contexts_on_one_gpu"] == 4 and line["value"] > 0 and "art_init_devices" in line["config"]["parallelism"]
    assert line["roofline"] is not None and line["roofline"]["device"].startswith("device 0 of 4") and line["config"]["end_to_end_s"] > 0

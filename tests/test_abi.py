"""The C-ABI library loads and exports every symbol include/art_hip.h declares; without a GPU it fails loudly."""
import ctypes as C
import os
import re

import pytest


def test_library_exports_every_declared_symbol(art):
    hdr = open(os.path.join(art.ROOT, "include", "art_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:art|gcore)_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(art.EXPORTED_SYMBOLS)
    L = art.load_library()
    for name in sorted(declared):
        assert getattr(L, name) is not None, name


def test_struct_layouts_match_the_header(art):
    assert C.sizeof(art.HitCpp) == 36                       # embree_connect.cpp:186-194
    assert C.sizeof(art.ArtMaterial) == 40 and C.sizeof(art.ArtSphere) == 20 and C.sizeof(art.ArtLight) == 76
    assert C.sizeof(art.ArtPassParams) == 48 and art.ArtPassParams.seed.offset == 32
    assert C.sizeof(art.ArtHit) == 44
    assert C.sizeof(art.ArtStats) == 14 * 8
    assert C.sizeof(art.ArtInstance) == 52 and art.ArtSceneDesc.n_instances.offset == art.ArtSceneDesc.cam_matrix.offset + 64
    assert C.sizeof(art.ArtStageStats) == 3 * 8 + 2 * 8 + 32 * 8 and C.sizeof(art.ArtReduceInfo) == 16 + 8 + 4 * 64 + 8


def test_no_cpu_fallback(art):
    """In the build container there is no GPU: the product must refuse to run rather than fall back to a CPU path."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as g; a = g.load_package()\n"
            "import ctypes\n"
            "try:\n    a.Backend(0); print('HAS_GPU')\nexcept a.ArtError as e:\n    print('ERR', e)\n") % art.ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
    if "HAS_GPU" in out:
        pytest.skip("a GPU is present")
    assert "no HIP device" in out and "no CPU path" in out


def test_host_mirror_scene_init_equals_oracle(art):
    """C++ mirror of Scene.Init (host/art_host.cpp, scene.adb:89-217) builds the same numbers as the oracle."""
    import numpy as np
    import orc
    so = art.HOST_LIB_PATH
    L = C.CDLL(so)
    sph = np.zeros((3, 5), np.float32); light = np.zeros(16, np.float32); mats = np.zeros((11, 10), np.float32)
    pos = np.zeros((64, 3), np.float32); bbox = np.zeros(6, np.float32); counts = (C.c_int * 4)()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    assert L.art_host_cornell_scene(orc.PYRAMID_VSGF.encode(), fp(sph), fp(light), fp(mats), fp(pos), fp(bbox), counts) == 0
    cs = orc.CornellScene(); s = cs.scene; a = cs.mesh_arrays()
    assert list(counts) == [3, 11, 17, 8]
    for i in range(3):
        assert list(sph[i, :3]) == list(s.spheres[i].pos) and sph[i, 3] == s.spheres[i].r and sph[i, 4] == s.spheres[i].mat
    assert list(light[2:5]) == list(s.lights[0].center) and light[5] == s.lights[0].radius
    assert list(light[6:9]) == list(s.lights[0].intensity) and light[9] == s.lights[0].surfaceArea
    for i in range(11):
        assert mats[i, 0] == s.materials[i].type and list(mats[i, 2:10]) == list(s.materials[i].p)
    assert np.array_equal(pos[:17].view(np.uint32), a["pos"].view(np.uint32))       # RotationMatrix(-Pi/6) etc. bit for bit
    assert np.array_equal(bbox[:3], a["bbmin"]) and np.array_equal(bbox[3:], a["bbmax"])


def test_scene_validation_errors_are_reported_without_a_gpu(art):
    """art_upload_scene validates the description before it touches the device: every malformed scene is refused with a message."""
    from ada_ray_tracer_amd import scenes
    L = art.load_library()

    def err_of(sd):
        rc = L.art_upload_scene(C.byref(sd.desc))
        assert rc != 0
        return L.art_last_error().decode()

    mats = scenes.cornell_materials()
    light = [scenes.sphere_light(0.0, 4)]
    assert "no light" in err_of(art.SceneDesc([], [], mats, [], None))
    assert "sphere.mat out of range" in err_of(art.SceneDesc([((0, 0, 0), 1.0, 99)], light, mats, [], None))
    bad_light_mat = [dict(type=art.MAT_LIGHT, light=3)]
    assert "missing light" in err_of(art.SceneDesc([], light, bad_light_mat, [], None))
    tri = dict(mode=art.MESH_CLOSEST, pos=[[0, 0, 0], [1, 0, 0], [0, 1, 0]], nrm=[[0, 0, 1]] * 3, idx=[[0, 1, 7]], matid=[1])
    assert "index out of range" in err_of(art.SceneDesc([], light, mats, [tri], None))
    tri["idx"] = [[0, 1, 2]]; tri["matid"] = [55]
    assert "material id out of range" in err_of(art.SceneDesc([], light, mats, [tri], None))
    tri["matid"] = [1]; tri["pos"] = [[0, 0, 0], [float("nan"), 0, 0], [0, 1, 0]]
    assert "non-finite" in err_of(art.SceneDesc([], light, mats, [tri], None))
    tri["pos"] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    two = [dict(tri), dict(tri)]
    assert "at most one CLOSEST mesh" in err_of(art.SceneDesc([], light, mats, two, None))
    assert "Cornell box material" in err_of(art.SceneDesc([], light, mats, [], dict(min=(0, 0, 0), max=(1, 1, 1), mat=(2, 3, 1, 1, 80, 1), nrm=scenes.CORNELL_BOX["nrm"])))
    p = art.Backend.pass_params()
    assert L.art_render_pass(C.byref(p), None, None, None) != 0 and "no scene" in L.art_last_error().decode()
    assert L.art_set_option(b"no_such_option", 1) != 0 and L.art_set_shard(3, 2, 32) != 0


def test_product_host_layer_builds_the_reference_scene(art):
    """scenes.reference_scene(): Scene.Init (scene.adb:24-27, 89-217) by host/art_host.cpp -- what bench.py --scene c2 and the
    reference-picture GPU test upload -- equals the oracle's construction byte for byte (spheres, light, materials, transformed mesh)."""
    import ctypes as C

    import conv
    import orc
    from ada_ray_tracer_amd import scenes
    d = scenes.reference_scene().desc
    od = conv.desc_from_oracle(art, orc.CornellScene()).desc

    def same(a, b, n):
        return C.string_at(a, n) == C.string_at(b, n)
    assert (d.n_spheres, d.n_lights, d.n_materials, d.n_meshes, d.has_cornell) == (3, 1, 11, 1, 1)
    assert same(d.spheres, od.spheres, C.sizeof(art.ArtSphere) * 3) and same(d.lights, od.lights, C.sizeof(art.ArtLight))
    assert same(d.materials, od.materials, C.sizeof(art.ArtMaterial) * 11)
    m, om = d.meshes[0], od.meshes[0]
    assert (m.mode, m.nverts, m.ntris) == (om.mode, om.nverts, om.ntris) == (art.MESH_REFERENCE_BF, 17, 8)
    assert same(m.pos, om.pos, 17 * 12) and same(m.nrm, om.nrm, 17 * 12) and same(m.idx, om.idx, 24 * 4)
    assert list(m.bbmin) == list(om.bbmin) and list(m.bbmax) == list(om.bbmax)
    assert list(d.cam_pos) == list(od.cam_pos) and list(d.cam_matrix) == list(od.cam_matrix)
    assert list(d.cb_min) == list(od.cb_min) and list(d.cb_mat) == list(od.cb_mat)
    assert open(scenes.PYRAMID_VSGF, "rb").read() == open(orc.PYRAMID_VSGF, "rb").read()


def test_c1_eight_sphere_scene_golden(art):
    """BASELINE configs[0]: the named 8-sphere scene; the oracle's RT_DEBUG ids match the committed golden and show all 8 spheres."""
    import numpy as np

    import conv
    import orc
    from ada_ray_tracer_amd import scenes
    sd = scenes.eight_sphere_scene()
    assert sd.desc.n_spheres == 8 and sd.desc.n_meshes == 0
    _, prim, mat, ptype = orc.debug_pass(conv.OracleScene(sd).scene, orc.make_params(64, 64, orc.RT_DEBUG, False))
    g = np.load(orc.GOLDEN + "/c1_debug_64.npz")
    assert np.array_equal(prim, g["prim"]) and np.array_equal(mat, g["mat"]) and np.array_equal(ptype, g["ptype"])
    assert set(np.unique(prim[ptype == 1])) == set(range(8))

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
    assert C.sizeof(art.ArtStats) == 13 * 8


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
    so = os.path.join(art.PKG_DIR, "libart_host.so")
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

"""Oracle pinned against every fixture that exists for this path: Philox KAT vectors, ART-M1 vs mpmath,
the reference's data/pyramid2.vsgf decode, Ada rounding in the resolve, BMP layout, and regression fixtures."""
import json
import struct

import numpy as np

import orc


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors (Salmon et al., SC'11)
    assert orc.philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_rng_uniform_is_24_bit_grid_in_unit_interval():
    L = orc.lib()
    v = np.array([L.orc_rng_uniform(7, p, s, b, k) for p in range(20) for s in range(3) for b in range(3) for k in range(5)], np.float64)
    assert (v >= 0).all() and (v < 1).all()
    assert np.array_equal(v * 2 ** 24, np.round(v * 2 ** 24))
    assert abs(v.mean() - 0.5) < 0.05


def test_art_m1_is_correctly_rounded_on_samples():
    import mpmath as mp
    mp.mp.prec = 120
    L = orc.lib()
    rng = np.random.default_rng(0)
    xs = (rng.random(3000) * 2 * np.pi).astype(np.float32)
    for x in xs:
        assert np.float32(L.orc_sinf(float(x))) == np.float32(float(mp.sin(mp.mpf(float(x)))))
        assert np.float32(L.orc_cosf(float(x))) == np.float32(float(mp.cos(mp.mpf(float(x)))))
    for x in rng.random(1500).astype(np.float32):
        if x == 0:
            continue
        for y in (80.0, 2.0 / 81.0, 1.0 / 81.0):
            y = np.float32(y)
            assert np.float32(L.orc_powf(float(x), float(y))) == np.float32(float(mp.power(mp.mpf(float(x)), mp.mpf(float(y)))))


def test_ada_power_special_cases():
    L = orc.lib()
    assert L.orc_powf(0.0, 2.5) == 0.0 and L.orc_powf(3.0, 0.0) == 1.0 and L.orc_powf(1.0, 77.0) == 1.0
    assert L.orc_powf(0.3, 1.0) == np.float32(0.3) and L.orc_powf(3.0, 2.0) == 9.0
    assert L.orc_powf(2.0, 0.5) == np.float32(np.sqrt(np.float32(2.0)))
    assert np.isnan(L.orc_powf(-1.0, 2.5)) and np.isnan(L.orc_powf(0.0, 0.0))
    assert L.orc_tanf(np.float32(np.pi / 4)) == 1.0          # camera z = -width / tan(fov/2), ray_tracer.adb:67


def test_vsgf_decode_matches_independent_decoder():
    g = json.load(open(orc.GOLDEN + "/vsgf_decode.json"))
    assert (g["fileSizeInBytes"], g["verticesNum"], g["indicesNum"], g["flags"]) == (1104, 17, 24, 1)
    L = orc.lib()
    ident = np.eye(4, dtype=np.float32).ravel()
    m = orc.Mesh()
    assert L.orc_load_vsgf(orc.PYRAMID_VSGF.encode(), orc.fp(ident), m) == 0
    assert (m.nverts, m.ntris) == (17, 8)
    pos = np.ctypeslib.as_array(m.pos, (17, 3)); nrm = np.ctypeslib.as_array(m.nrm, (17, 3))
    assert np.array_equal(pos, np.array(g["positions"], np.float32))
    assert np.array_equal(nrm, np.array(g["normals"], np.float32))
    assert np.ctypeslib.as_array(m.idx, (24,)).tolist() == g["indices"]
    assert np.ctypeslib.as_array(m.matid, (8,)).tolist() == g["material_ids"] == [2] * 8
    assert not np.ctypeslib.as_array(m.uv, (17, 2)).any()          # geometry.adb:565-566
    assert g["indices"] == list(range(15)) + [14, 13, 15, 15, 13, 16, 16, 13, 12]


def test_cornell_scene_constants():
    cs = orc.CornellScene()
    s = cs.scene
    assert s.n_spheres == 3 and s.n_lights == 1 and s.n_materials == 11
    assert list(s.spheres[2].pos) == [0.0, 4.5, 1.0] and s.spheres[2].r == 0.5 and s.spheres[2].mat == 4
    assert list(s.lights[0].intensity) == [10.0, 10.0, 10.0]
    assert s.lights[0].surfaceArea == np.float32(np.pi)
    assert list(s.cb_mat) == [2, 3, 1, 1, 8, 1]
    assert [s.materials[i].type for i in range(11)] == [4, 2, 2, 2, 1, 3, 0, 0, 5, 2, 2]
    a = cs.mesh_arrays()
    assert np.allclose(a["bbmin"], [-1.3645577, 0.1, 2.4611044]) and np.allclose(a["bbmax"], [-0.1354422, 0.73832464, 3.7388954])


def test_resolve_rounds_like_ada_and_bmp_layout():
    acc = np.zeros((2, 4, 3), np.float32)
    acc[0, 0] = (0.25, 1.0, 4.0); acc[0, 1] = (0.0, 1.0 / 16.0, 9.0 / 64.0)
    img = orc.resolve(acc, 1)
    assert img[0, 0] == (128 | (255 << 8) | (255 << 16))         # sqrt(.25)*255 = 127.5 -> 128 (ties away), clamp
    assert img[0, 1] == 0 | (64 << 8) | (96 << 16)               # 63.75 -> 64, 95.625 -> 96
    b = orc.bmp_bytes(img)
    assert len(b) == 54 + 2 * 4 * 3
    assert b[:2] == b"BM" and struct.unpack_from("<IHHI", b, 2) == (54 + 24, 0, 0, 54)
    assert struct.unpack_from("<IIIHHIIIIII", b, 14) == (40, 4, 2, 1, 24, 0, 0, 0, 0, 0, 0)
    assert b[54:57] == bytes([255, 255, 128])                    # bytes = bits 16-23, 8-15, 0-7 (bitmap.adb:75-79)


def test_regression_fixtures():
    cs = orc.CornellScene()
    g = np.load(orc.GOLDEN + "/cornell_debug_64.npz")
    _, prim, mat, ptype = orc.debug_pass(cs.scene, orc.make_params(64, 64, orc.RT_DEBUG, False))
    assert np.array_equal(prim, g["prim"]) and np.array_equal(mat, g["mat"]) and np.array_equal(ptype, g["ptype"])
    assert set(np.unique(ptype)) == {-1, 0, 1, 2}
    g = np.load(orc.GOLDEN + "/cornell_mis_32.npz")
    acc, spp, cnt = orc.render(cs.scene, orc.make_params(32, 32, orc.PT_MIS, True, 8, 1, seed=1), passes=2)
    assert spp == int(g["spp"]) and cnt.rays == int(g["rays"])
    assert np.array_equal(acc.view(np.uint32), g["accum_bits"])


def test_intersect_triangle_against_numpy_float32_transcription():
    """geometry.adb:231-263 transcribed independently in numpy float32 (same operation order)."""
    f = np.float32
    A, B, C = np.array([0, 0, -3], f), np.array([1, 0, -3], f), np.array([0, 1, -3], f)
    o, d = np.array([0.2, 0.3, 0.0], f), np.array([0.05, -0.02, -1.0], f)
    d = (d * (f(1) / np.sqrt(f(d[0] * d[0] + d[1] * d[1]) + f(d[2] * d[2])))).astype(f)
    e1, e2 = B - A, C - A
    cross = lambda a, b: np.array([a[1] * b[2] - b[1] * a[2], a[2] * b[0] - b[2] * a[0], a[0] * b[1] - b[0] * a[1]], f)
    dot = lambda a, b: f(f(f(a[0] * b[0]) + f(a[1] * b[1])) + f(a[2] * b[2]))
    pv, tv = cross(d, e2), o - A
    qv = cross(tv, e1)
    inv = f(1) / max(dot(e1, pv), f(1e-25))
    v, u, t = dot(tv, pv) * inv, dot(qv, d) * inv, dot(e2, qv) * inv
    import conv  # noqa: F401
    mats = (orc.Material * 3)(); mats[2].type = orc.MAT_LAMBERT
    light = (orc.Light * 1)(); light[0].shape = orc.LIGHT_SPHERE; light[0].mat = 0
    pos = np.stack([A, B, C]).astype(f); nrm = np.tile(np.array([0, 0, 1], f), (3, 1)); uv = np.zeros((3, 2), f)
    idx = np.array([0, 1, 2], np.int32); mid = np.array([2], np.int32)
    m = (orc.Mesh * 1)()
    m[0].mode = orc.MESH_CLOSEST; m[0].nverts = 3; m[0].ntris = 1
    m[0].pos = orc.fp(pos); m[0].nrm = orc.fp(nrm); m[0].uv = orc.fp(uv); m[0].idx = orc.ip(idx); m[0].matid = orc.ip(mid)
    s = orc.Scene(); s.n_lights = 1; s.lights = light; s.n_materials = 3; s.materials = mats; s.n_meshes = 1; s.meshes = m
    h = orc.closest_hits(s, o[None], d[None])[0]
    assert h.is_hit == 1 and h.prim_type == 2 and h.prim_index == 0
    assert np.float32(h.t) == t and np.float32(h.tx) == 0 and v > 0 and u > 0
    back = orc.closest_hits(s, (o + np.array([0, 0, -6], f))[None], (d * np.array([1, 1, -1], f))[None])[0]
    assert back.is_hit == 0        # max(det, 1e-25) rejects back faces (geometry.adb:243)


def test_task_pool_organisation_gives_the_same_bits():
    """orc_render_pass_tasks (Threads_Num tasks, each a whole-frame DoPass into a private frame: the reference's organisation,
    ray_tracer.adb:142-194) == orc_render_pass (pixel-parallel), bit for bit, for any number of OS threads."""
    cs = orc.CornellScene()
    for aa, nthreads in ((True, 3), (False, 0)):
        a, spp_a, ca = orc.render(cs.scene, orc.make_params(40, 32, orc.PT_MIS, aa, 8, 5, seed=4))
        b, spp_b, cb = orc.render_tasks(cs.scene, orc.make_params(40, 32, orc.PT_MIS, aa, 8, 5, seed=4, nthreads=nthreads))
        assert spp_a == spp_b and ca.rays == cb.rays and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_pixel_list_pass_is_the_frame_pass():
    """orc_render_pixels (the pass for a list of pixels, used where a whole BASELINE frame is out of reach: tests/test_gpu_stated_spp.py)
    == the same pixels of orc_render_pass, also cumulatively over two passes (spp0 = 12 for the second)."""
    cs = orc.CornellScene()
    prm = orc.make_params(40, 32, orc.PT_MIS, True, 8, 3, seed=4)
    frame, spp, cnt = orc.render(cs.scene, prm, passes=2)
    assert spp == 24
    ys, xs = np.mgrid[0:32, 0:40]
    xs = xs.ravel()[::7]; ys = ys.ravel()[::7]
    part, c1 = orc.render_pixels(cs.scene, prm, xs, ys, 0)
    part, c2 = orc.render_pixels(cs.scene, prm, xs, ys, 12, part)
    assert c1.samples == len(xs) * 12
    assert np.array_equal(part.view(np.uint32), frame[ys, xs].view(np.uint32))

"""GPU parity tests: the HIP path (through the C ABI of libart_hip.so) against the CPU oracle on the same
seeded inputs.  Bit-exact for hit indices AND for radiance (the arithmetic contract makes the float path
reproducible), which is stronger than BASELINE's 1e-4 per-channel tolerance; the tolerance form is also
asserted so the stated bar is visible:  max |gpu - cpu| / spp <= 1e-4 per channel."""
import ctypes as C

import numpy as np
import pytest

import conv
import orc

pytestmark = pytest.mark.gpu

TOL = 1.0e-4   # BASELINE.json north_star: per-channel radiance within 1e-4 of the CPU reference at equal spp


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_radiance_equal(gpu, cpu, spp):
    finite = np.isfinite(cpu)
    assert np.array_equal(np.isfinite(gpu), finite)
    err = np.abs(gpu[finite] - cpu[finite]).max() / spp if finite.any() else 0.0
    assert err <= TOL, "per-channel radiance error %g > %g" % (err, TOL)
    assert np.array_equal(bits(gpu), bits(cpu)), "radiance not bit-identical (max err %g)" % err


@pytest.fixture(scope="module")
def cornell(art):
    cs = orc.CornellScene()
    return cs, conv.desc_from_oracle(art, cs)


@pytest.mark.parametrize("rt", ["PT_MIS", "PT_SHADOW", "PT_STUPID"])
@pytest.mark.parametrize("aa", [True, False])
def test_cornell_render_pass_bit_exact(art, backend, cornell, rt, aa):
    """C1/C2 scene (scene.adb:89-217 incl. pyramid2.vsgf through the reference brute-force mesh path)."""
    cs, sd = cornell
    backend.upload_scene(sd)
    backend.resize(96, 80)
    p = art.Backend.pass_params(getattr(art, rt), aa, 8, 2, seed=11)
    accum, _, spp = backend.render_pass(p, 0)
    ref, rspp, cnt = orc.render(cs.scene, orc.make_params(96, 80, getattr(orc, rt), aa, 8, 2, seed=11))
    assert spp == rspp
    assert_radiance_equal(accum, ref, spp)
    assert backend.stats().rays == cnt.rays


def test_c1_eight_sphere_scene(art, backend):
    """BASELINE configs[0] at its own size: 8 spheres, 256x256, 1 spp (AA off), PT_MIS -- plumbing; radiance bit-exact vs the oracle,
    RT_DEBUG ids equal to the committed golden."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.eight_sphere_scene()
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    backend.resize(256, 256)
    accum, _, spp = backend.render_pass(art.Backend.pass_params(art.PT_MIS, False, 8, 1, seed=1), 0)
    ref, rspp, cnt = orc.render(osc.scene, orc.make_params(256, 256, orc.PT_MIS, False, 8, 1, seed=1))
    assert spp == rspp == 1
    assert_radiance_equal(accum, ref, spp)
    assert backend.stats().rays == cnt.rays
    backend.resize(64, 64)
    _, _, prim, mat, ptype = backend.debug_hit_pass(art.Backend.pass_params(art.RT_DEBUG, False, 8, 1))
    g = np.load(orc.GOLDEN + "/c1_debug_64.npz")
    assert np.array_equal(prim, g["prim"]) and np.array_equal(mat, g["mat"]) and np.array_equal(ptype, g["ptype"])


def _tilted_camera_scene(art):
    """Internal scene seen through a rotated + translated camera matrix (cam.matrix * dir adds the translation column,
    vector_math.adb:137-144, before the normalize of integrators.adb:46)."""
    cs = orc.CornellScene()
    c, s = np.float32(np.cos(0.2)), np.float32(np.sin(0.2))
    m = np.array([[c, 0, s, 0.05], [0, 1, 0, -0.02], [-s, 0, c, 0.01], [0, 0, 0, 1]], np.float32)
    cs.scene.cam_matrix = (C.c_float * 16)(*[float(v) for v in m.ravel()])
    cs.scene.cam_pos = (C.c_float * 3)(0.4, 2.4, 11.0)
    return cs, conv.desc_from_oracle(art, cs)


@pytest.mark.parametrize("depth,aa,bg,seed", [(3, True, (0.1, 0.2, 0.3), 1 << 40 | 5), (12, False, (0.0, 0.05, 0.0), 77), (1, True, (0.0, 0.0, 0.0), 3)])
def test_camera_matrix_background_depth_and_wide_seed(art, backend, depth, aa, bg, seed):
    cs, sd = _tilted_camera_scene(art)
    backend.upload_scene(sd)
    backend.resize(72, 56)
    p = art.Backend.pass_params(art.PT_MIS, aa, depth, 2, seed=seed, background=bg)
    accum, _, spp = backend.render_pass(p, 0)
    ref, rspp, cnt = orc.render(cs.scene, orc.make_params(72, 56, orc.PT_MIS, aa, depth, 2, seed=seed, background=bg))
    assert spp == rspp
    assert_radiance_equal(accum, ref, spp)
    assert backend.stats().rays == cnt.rays


def test_two_passes_accumulate_like_reference(art, backend, cornell):
    cs, sd = cornell
    backend.upload_scene(sd)
    backend.resize(64, 64)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 3, seed=2)
    _, _, spp = backend.render_pass(p, 0, want_accum=False)
    accum, screen, spp = backend.render_pass(p, spp, True, True)
    ref, rspp, _ = orc.render(cs.scene, orc.make_params(64, 64, orc.PT_MIS, True, 8, 3, seed=2), passes=2)
    assert spp == rspp == 24
    assert_radiance_equal(accum, ref, spp)
    assert np.array_equal(screen, orc.resolve(ref, rspp))


def test_small_batches_do_not_change_the_image(art, backend, cornell):
    cs, sd = cornell
    backend.upload_scene(sd)
    backend.resize(64, 48)
    backend.set_option("batch_paths", 4096)
    try:
        p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=9)
        accum, _, spp = backend.render_pass(p, 0)
    finally:
        backend.set_option("batch_paths", 32 << 20)
    ref, _, _ = orc.render(cs.scene, orc.make_params(64, 48, orc.PT_MIS, True, 8, 2, seed=9))
    assert_radiance_equal(accum, ref, spp)


def test_debug_hit_pass_and_bmp_bytes(art, backend, cornell):
    """RT_DEBUG (ray_tracer.adb:208-261): hit indices bit-exact, LDR image and BMP file bytes identical."""
    cs, sd = cornell
    backend.upload_scene(sd)
    backend.resize(256, 256)
    accum, screen, prim, mat, ptype = backend.debug_hit_pass(art.Backend.pass_params(art.RT_DEBUG, False, 8, 1))
    oacc, oprim, omat, optype = orc.debug_pass(cs.scene, orc.make_params(256, 256, orc.RT_DEBUG, False))
    assert np.array_equal(prim, oprim) and np.array_equal(mat, omat) and np.array_equal(ptype, optype)
    assert np.array_equal(bits(accum), bits(oacc))
    oscreen = orc.resolve(oacc, 1)
    assert np.array_equal(screen, oscreen)
    assert art.save_bmp(None, screen) == orc.bmp_bytes(oscreen)
    golden = np.load(orc.GOLDEN + "/cornell_debug_64.npz")
    backend.resize(64, 64)
    _, _, prim, mat, ptype = backend.debug_hit_pass(art.Backend.pass_params(art.RT_DEBUG, False, 8, 1))
    assert np.array_equal(prim, golden["prim"]) and np.array_equal(mat, golden["mat"]) and np.array_equal(ptype, golden["ptype"])


def test_ada_xy_layout_is_the_transpose(art, backend, cornell):
    """AccumBuff(x,y) / ScreenBufferData(x,y): element (x,y) at x*height + y (ray_tracer.ads:35,54)."""
    cs, sd = cornell
    backend.upload_scene(sd)
    backend.resize(80, 48)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=4, layout=art.LAYOUT_ADA_XY)
    accum_xy, screen_xy, spp = backend.render_pass(p, 0, True, True)
    assert accum_xy.shape == (80, 48, 3) and screen_xy.shape == (80, 48)
    ref, _, _ = orc.render(cs.scene, orc.make_params(80, 48, orc.PT_MIS, True, 8, 1, seed=4))
    assert np.array_equal(bits(accum_xy), bits(ref.transpose(1, 0, 2)))
    assert np.array_equal(screen_xy, orc.resolve(ref, spp).T)


def _random_rays(n, seed):
    rng = np.random.default_rng(seed)
    o = (rng.random((n, 3)) * [4.6, 4.4, 4.6] + [-2.3, 0.3, 0.2]).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


def _assert_hits_equal(gh, oh):
    a, b = conv.hits_to_arrays(gh), conv.hits_to_arrays(oh)
    hit = b[1] == 1
    assert np.array_equal(a[1], b[1])
    assert np.array_equal(a[2], b[2])
    for k in (0, 3, 4, 5):
        x, y = np.asarray(a[k])[hit], np.asarray(b[k])[hit]
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert np.array_equal(x, y), "hit field %d differs" % k


@pytest.mark.parametrize("kernel", ["TRACE_COOP", "TRACE_SIMPLE"])
def test_find_closest_hit_reference_scene(art, backend, cornell, kernel):
    cs, sd = cornell
    backend.upload_scene(sd)
    o, d = _random_rays(50000, 5)
    _assert_hits_equal(backend.trace_rays(o, d, kernel=getattr(art, kernel)), orc.closest_hits(cs.scene, o, d))


@pytest.mark.parametrize("kernel", ["TRACE_COOP", "TRACE_SIMPLE"])
@pytest.mark.parametrize("ntris", [1, 7, 300, 20000])
def test_find_closest_hit_bvh_vs_brute_force(art, backend, kernel, ntris):
    """BVH traversal (any order) == brute-force minimum over all triangles (lowest index on ties)."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(ntris, 3)
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    o, d = _random_rays(30000, ntris)
    _assert_hits_equal(backend.trace_rays(o, d, kernel=getattr(art, kernel)), orc.closest_hits(osc.scene, o, d))


def test_traversal_counters_match_oracle_walk(art, backend):
    """box tests B and triangle tests T per ray, counted by the kernel == the oracle's walk of the
    exported BVH in the published order (mesh-only scene so every ray starts unbounded)."""
    from ada_ray_tracer_amd import scenes
    mesh = scenes.random_triangles(20000, 77)
    mats = scenes.cornell_materials()
    lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    sd = art.SceneDesc([], lights, mats, [mesh], None, scenes.REFERENCE_CAMERA)   # mesh only: no analytic primitive to hit
    backend.upload_scene(sd)
    nodes, tris, info = backend.export_bvh()
    o, d = _random_rays(40000, 8)
    d[:100, 0] = 0.0   # axis-parallel directions: 1/0 = inf in the slab test
    t, prim, cnt = orc.bvh_walk(nodes, tris, o, d, width=info.node_width)
    for kernel in (art.TRACE_COOP, art.TRACE_SIMPLE):
        hits, st = backend.trace_rays(o, d, kernel=kernel, want_stats=True)
        gprim = np.array([h.prim_index if h.is_hit else -1 for h in hits], np.int32)
        gt = np.array([h.t for h in hits], np.float32)
        assert np.array_equal(gprim, prim)
        assert np.array_equal(gt[prim >= 0].view(np.uint32), t[prim >= 0].view(np.uint32))
        assert (st.box_tests, st.tri_tests, st.node_visits, st.leaf_visits, st.traced_rays) == \
               (cnt.box_tests, cnt.tri_tests, cnt.node_visits, cnt.leaf_visits, cnt.rays)


@pytest.mark.parametrize("ntris", [2000])
def test_synthetic_multi_light_scene_bit_exact(art, backend, ntris):
    """C3-style scene at a size the brute-force oracle finishes in seconds: Cornell walls + random triangles +
    3 rect lights (uniform light choice), PT_MIS."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(ntris, 3)
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    backend.resize(64, 64)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=3)
    accum, _, spp = backend.render_pass(p, 0)
    ref, _, cnt = orc.render(osc.scene, orc.make_params(64, 64, orc.PT_MIS, True, 8, 1, seed=3))
    assert_radiance_equal(accum, ref, spp)
    assert backend.stats().rays == cnt.rays


def test_rect_light_mis_nan_pattern_matches(art, backend):
    """Rect AreaLights + PT_MIS overflow to NaN on the ceiling in the reference's arithmetic (see scenes.synthetic_scene);
    the GPU reproduces the very same NaN pixels.  With PT_SHADOW (no MIS weight) the same scene is finite."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(800, 3, rect_lights=True)
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    for rt, ort in ((art.PT_MIS, orc.PT_MIS), (art.PT_SHADOW, orc.PT_SHADOW)):
        backend.resize(48, 48)
        accum, _, spp = backend.render_pass(art.Backend.pass_params(rt, True, 8, 1, seed=3), 0)
        ref, _, _ = orc.render(osc.scene, orc.make_params(48, 48, ort, True, 8, 1, seed=3))
        assert np.array_equal(bits(accum), bits(ref))
        assert np.isnan(ref).any() == (rt == art.PT_MIS)


def test_mixed_scene_bit_exact(art, backend):
    """C5-style scene: spheres (Phong, glass, diffuse), emissive sphere light and a mesh."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.mixed_scene(1500, 5)
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    backend.resize(64, 64)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=6)
    accum, _, spp = backend.render_pass(p, 0)
    ref, _, _ = orc.render(osc.scene, orc.make_params(64, 64, orc.PT_MIS, True, 8, 1, seed=6))
    assert_radiance_equal(accum, ref, spp)


@pytest.mark.parametrize("rt", ["PT_MIS", "PT_SHADOW", "PT_STUPID"])
def test_mirror_material_on_sphere_and_mesh_bit_exact(art, backend, rt):
    """MaterialMirror (materials.adb:232-264) on a sphere and on mesh triangles, next to every other material, all three integrators."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.mirror_scene()
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    backend.resize(80, 64)
    p = art.Backend.pass_params(getattr(art, rt), True, 8, 2, seed=12)
    accum, _, spp = backend.render_pass(p, 0)
    ref, rspp, cnt = orc.render(osc.scene, orc.make_params(80, 64, getattr(orc, rt), True, 8, 2, seed=12))
    assert spp == rspp
    assert_radiance_equal(accum, ref, spp)
    assert backend.stats().rays == cnt.rays
    # the mirror really is in the picture: primary hits on material 5 (sphere: matId 0 / mat 5; triangles: matId 5)
    _, _, prim, mat, ptype = backend.debug_hit_pass(art.Backend.pass_params(art.RT_DEBUG, False, 8, 1))
    hits = backend.trace_rays(*_camera_rays(80, 64))
    mats = np.array([h.mat for h in hits])
    assert (mats == 5).sum() > 200 and ((mats == 5) & (np.array([h.prim_type for h in hits]) == 2)).sum() > 5


def _camera_rays(w, h):
    """primary rays of the reference camera (ray_tracer.adb:61-69: z' = -w / tan(pi/4), pixel centres)"""
    x, y = np.meshgrid(np.arange(w, dtype=np.float32), np.arange(h, dtype=np.float32))
    d = np.stack([x + 0.5 - w / 2.0, y + 0.5 - h / 2.0, np.full_like(x, -float(w))], -1).reshape(-1, 3)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    o = np.tile(np.array([0.0, 2.55, 12.5], np.float32), (d.shape[0], 1))
    return o, d


@pytest.mark.parametrize("config,rt", [("c3", "PT_MIS"), ("c4", "PT_MIS"), ("c5", "PT_MIS"), ("c5", "PT_SHADOW"), ("c5", "PT_STUPID"), ("c3", "PT_SHADOW")])
def test_baseline_scenes_at_full_triangle_count(art, backend, config, rt):
    """BASELINE configs C3 (100k triangles), C4 (1M) and C5 (mixed 20k) at their real scene size, reduced frame: the oracle's
    mesh search walks the exported BVH (proven equal to its brute-force scan in test_host_sim_parity / the <=20k cases above),
    everything else is the oracle's own recursion.  Radiance bit-exact, ray counts equal."""
    from ada_ray_tracer_amd import scenes
    sd = {"c3": lambda: scenes.synthetic_scene(100000, 3), "c4": lambda: scenes.synthetic_scene(1000000, 4),
          "c5": lambda: scenes.mixed_scene(20000, 5)}[config]()
    backend.upload_scene(sd)
    osc = conv.OracleScene(sd)
    nodes, tris, info = backend.export_bvh()
    osc.attach_bvh(nodes, tris, info.node_width)
    W, H = (96, 54)
    backend.resize(W, H)
    p = art.Backend.pass_params(getattr(art, rt), True, 8, 2, seed=1)
    accum, _, spp = backend.render_pass(p, 0)
    ref, rspp, cnt = orc.render(osc.scene, orc.make_params(W, H, getattr(orc, rt), True, 8, 2, seed=1))
    assert spp == rspp == 8
    assert_radiance_equal(accum, ref, spp)
    assert backend.stats().rays == cnt.rays and np.isfinite(ref).all()


@pytest.mark.parametrize("config", ["c3", "c4"])
def test_bench_scale_tree_is_sound_and_hits_equal_brute_force(art, backend, config):
    """Independent of the product's tree (the test above lets the oracle walk it): the uploaded C3 / C4 tree is checked structurally
    (tests/bvh_check.py) and the trace kernel's hits are compared with the oracle's O(N) scan over all 100 k / 1 M triangles."""
    import bvh_check
    from ada_ray_tracer_amd import scenes
    ntris = 100000 if config == "c3" else 1000000
    sd = scenes.synthetic_scene(ntris, 3 if config == "c3" else 4)
    backend.upload_scene(sd)
    nodes, tris, info = backend.export_bvh()
    pos, _, idx, _, _ = [a for a, m in zip(sd._mesh_arrays, sd.meshes) if m.mode == art.MESH_CLOSEST][0]
    r = bvh_check.check_tree(nodes, tris, info.n_nodes, info.max_stack, info.node_width, pos, idx)
    assert r["records"] == ntris
    o, d = _random_rays(6000 if config == "c3" else 2500, ntris)
    _assert_hits_equal(backend.trace_rays(o, d), orc.closest_hits(conv.OracleScene(sd).scene, o, d))


def test_pixel_tile_shards_sum_to_the_full_frame(art, backend, cornell):
    """8(e): interleaved pixel tiles, one owner per pixel -> the sum over ranks is bit-identical to 1 GPU."""
    cs, sd = cornell
    backend.upload_scene(sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=5)
    backend.set_shard(0, 1, 32)
    backend.resize(96, 64)
    full, _, _ = backend.render_pass(p, 0)
    total = np.zeros_like(full)
    for r in range(3):
        backend.set_shard(r, 3, 16)
        backend.resize(96, 64)
        part, _, _ = backend.render_pass(p, 0)
        assert np.count_nonzero(part.any(-1)) <= part.shape[0] * part.shape[1]
        total += part
    backend.set_shard(0, 1, 32)
    assert np.array_equal(bits(total), bits(full))


def test_property_full_size_synthetic(art, backend):
    """Size-independent properties at bench scale (100k triangles, no brute-force oracle possible):
    the two trace kernels agree bit-for-bit, hits are self-consistent, and the image is reproducible."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(100000, 3)
    backend.upload_scene(sd)
    o, d = _random_rays(200000, 21)
    a = conv.hits_to_arrays(backend.trace_rays(o, d, kernel=art.TRACE_COOP))
    b = conv.hits_to_arrays(backend.trace_rays(o, d, kernel=art.TRACE_SIMPLE))
    for x, y in zip(a, b):
        x = x.view(np.uint32) if x.dtype == np.float32 else x
        y = y.view(np.uint32) if y.dtype == np.float32 else y
        assert np.array_equal(x, y)
    assert (a[0][a[1] == 1] > 0).all()
    backend.resize(128, 128)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=1)
    img1, _, _ = backend.render_pass(p, 0)
    backend.resize(128, 128)
    backend.set_option("trace_kernel", art.TRACE_SIMPLE)
    try:
        img2, _, _ = backend.render_pass(p, 0)
    finally:
        backend.set_option("trace_kernel", art.TRACE_COOP)
    assert np.array_equal(bits(img1), bits(img2))


def test_gcore_seam(art, backend):
    """Legacy embree_connect.cpp symbols: add mesh -> instance (3x4 row-major) -> commit -> closest hit."""
    L = backend.lib
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32)
    idx = np.array([0, 1, 2, 2, 1, 3], np.int32)
    L.gcore_init_and_clear()
    mid = L.gcore_add_mesh_3f(verts.ctypes.data_as(art.f32p), 4, idx.ctypes.data_as(art.i32p), 6)
    assert mid == 0
    m = np.eye(4, dtype=np.float32); m[2, 3] = -5.0          # translate z by -5
    L.gcore_instance_meshes(mid, m.ctypes.data_as(art.f32p), 1)
    L.gcore_commit_scene()
    hit = art.HitCpp(); hit.primIndex = -1
    pos = (C.c_float * 3)(0.25, 0.25, 0.0); dirn = (C.c_float * 3)(0.0, 0.0, -1.0)
    assert L.gcore_closest_hit(pos, dirn, 0.0, 100000.0, C.byref(hit))
    assert hit.primIndex == 0 and hit.instIndex == 0 and hit.geomIndex == 0
    assert abs(hit.t - 5.0) < 1e-6
    assert list(hit.normal) == [0.0, 0.0, 1.0]              # Ng = cross(v1-v0, v2-v0), unnormalised
    assert abs(hit.texCoord[0] - 0.25) < 1e-6 and abs(hit.texCoord[1] - 0.25) < 1e-6
    dirn2 = (C.c_float * 3)(0.0, 0.0, 1.0)
    assert not L.gcore_closest_hit(pos, dirn2, 0.0, 100000.0, C.byref(hit))
    pos3 = (C.c_float * 3)(0.75, 0.75, -10.0)               # from behind: Embree is two-sided
    assert L.gcore_closest_hit(pos3, dirn2, 0.0, 100000.0, C.byref(hit))
    assert hit.primIndex == 1 and abs(hit.t - 5.0) < 1e-6
    assert not L.gcore_closest_hit(pos3, dirn2, 0.0, 4.0, C.byref(hit))   # beyond t_far
    L.gcore_destroy()


def test_gcore_closest_hit_is_thread_safe(art, backend):
    """The reference calls gcore_closest_hit concurrently from up to 28 tasks (scene_hydra_embree.adb:426-446)."""
    import threading
    L = backend.lib
    rng = np.random.default_rng(3)
    verts = (rng.random((300, 3)) * 4 - 2).astype(np.float32)
    idx = rng.integers(0, 300, 600).astype(np.int32)
    L.gcore_init_and_clear()
    mid = L.gcore_add_mesh_3f(verts.ctypes.data_as(art.f32p), 300, idx.ctypes.data_as(art.i32p), 600)
    m = np.eye(4, dtype=np.float32)
    L.gcore_instance_meshes(mid, m.ctypes.data_as(art.f32p), 1)
    L.gcore_commit_scene()
    n = 64
    o = (rng.random((n, 3)) * 2 - 1).astype(np.float32); o[:, 2] = 6.0
    d = np.tile(np.array([0, 0, -1], np.float32), (n, 1))

    def query(i):
        h = art.HitCpp(); h.primIndex = -1
        ok = L.gcore_closest_hit(o[i].ctypes.data_as(art.f32p), d[i].ctypes.data_as(art.f32p), 0.0, 100000.0, C.byref(h))
        return (bool(ok), h.primIndex, h.t) if ok else (False, -1, 0.0)
    serial = [query(i) for i in range(n)]
    out = [None] * n

    def worker(k):
        for i in range(k, n, 8):
            out[i] = query(i)
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert out == serial and any(s[0] for s in serial)
    L.gcore_destroy()


def test_gcore_queries_combine_and_batch(art, backend):
    """28 concurrent callers (Threads_Num, ray_tracer.ads:23) are combined into shared launches, and gcore_closest_hit_n answers a whole
    batch with one: same answers as one call per ray, and the rates say so (printed; round 1 served ~1 query per full persistent-grid
    launch under a global mutex)."""
    import threading
    import time
    L = backend.lib
    rng = np.random.default_rng(4)
    verts = (rng.random((3000, 3)) * 4 - 2).astype(np.float32)
    idx = rng.integers(0, 3000, 9000).astype(np.int32)
    L.gcore_init_and_clear()
    mid = L.gcore_add_mesh_3f(verts.ctypes.data_as(art.f32p), 3000, idx.ctypes.data_as(art.i32p), 9000)
    m = np.eye(4, dtype=np.float32)
    L.gcore_instance_meshes(mid, m.ctypes.data_as(art.f32p), 1)
    L.gcore_commit_scene()
    n = 28 * 40
    o = (rng.random((n, 3)) * 2 - 1).astype(np.float32); o[:, 2] = 6.0
    d = np.tile(np.array([0, 0, -1], np.float32), (n, 1))

    def query(i):
        h = art.HitCpp(); h.primIndex = -1
        ok = L.gcore_closest_hit(o[i].ctypes.data_as(art.f32p), d[i].ctypes.data_as(art.f32p), 0.0, 100000.0, C.byref(h))
        return (True, h.primIndex, h.instIndex, h.t, tuple(h.normal), tuple(h.texCoord)) if ok else (False,)
    t0 = time.perf_counter(); serial = [query(i) for i in range(200)]; t_serial = (time.perf_counter() - t0) / 200
    out = [None] * n

    def worker(k):
        for i in range(k, n, 28):
            out[i] = query(i)
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(28)]
    t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]; t_threads = (time.perf_counter() - t0) / n
    assert out[:200] == serial and sum(1 for s in out if s[0]) > n // 4
    hits = (art.HitCpp * n)(); found = (C.c_ubyte * n)()
    t0 = time.perf_counter()
    nh = L.gcore_closest_hit_n(n, o.ctypes.data_as(art.f32p), d.ctypes.data_as(art.f32p), None, None, hits, found)
    t_batch = (time.perf_counter() - t0) / n
    batch = [(True, hits[i].primIndex, hits[i].instIndex, hits[i].t, tuple(hits[i].normal), tuple(hits[i].texCoord)) if found[i] else (False,) for i in range(n)]
    assert batch == out and nh == sum(1 for s in out if s[0])            # the host walk of a single ray == the GPU batch, every field, bit for bit
    print("gcore queries/s through ctypes: serial %.0f, 28 threads %.0f, batch of %d %.0f" % (1 / t_serial, 1 / t_threads, n, 1 / t_batch))
    # the flat-combined GPU launches of rounds 2-3 (gcore_set_single_ray_on_gpu): the same answers again, and far slower per ray than a batch
    L.gcore_set_single_ray_on_gpu(1)
    try:
        t0 = time.perf_counter(); on_gpu = [query(i) for i in range(100)]; t_gpu1 = (time.perf_counter() - t0) / 100
        out2 = [None] * n

        def worker2(k):
            for i in range(k, n, 28):
                out2[i] = query(i)
        ts = [threading.Thread(target=worker2, args=(k,)) for k in range(28)]
        [t.start() for t in ts]; [t.join() for t in ts]
    finally:
        L.gcore_set_single_ray_on_gpu(0)
    assert on_gpu == serial[:100] and out2 == out
    # wall-clock ratios through ctypes depend on the host (cores, load, the GIL): reported above; only the orders of magnitude
    # are held -- a batch amortises the launch, and the host walk is not slower than a launch per ray
    assert t_batch * 5 < t_gpu1, "a batch must cost far less per ray than one launch per ray"
    assert t_serial < t_gpu1 * 2, "the host walk must not lose to a launch per ray"
    L.gcore_destroy()


@pytest.mark.parametrize("instances", [1, 64])
def test_gcore_single_ray_calls_serve_the_reference_s_call_pattern(art, backend, instances):
    """scene_hydra_embree.adb:426-446 calls gcore_closest_hit once per ray from up to 28 tasks; Embree answers on the caller's core
    (embree_connect.cpp:218).  host/gcore_bench.cpp does the same natively (ctypes would measure the GIL): 28 threads x one ray per call
    on a 200k-triangle soup (flattened upload) and on 64 instances of a 20k-triangle mesh (two-level scene), against ONE GPU batch of the
    same rays: every HitCpp the same bytes, and at least a million queries per second (round 3: 18 k/s)."""
    import json
    import os
    import subprocess
    backend.shutdown()                                   # the harness is its own process with its own backend
    exe = os.path.join(art.PKG_DIR, "gcore_bench")
    args = [exe, "200000", "400000", "28", "1"] if instances == 1 else [exe, "20000", "400000", "28", "64"]
    try:
        res = subprocess.run(args, capture_output=True, text=True, timeout=280)
    finally:
        backend.__init__(0)
    assert res.returncode == 0, res.stdout + res.stderr
    d = json.loads(res.stdout.strip().splitlines()[-1])
    print(d)
    assert d["different_from_gpu_batch"] == 0 and d["hits"] == d["batch_hits"] and d["hits"] > d["rays"] // 10
    # the rate depends on the host's cores and load (28 threads are asked for); the byte-equality above is the test, the rate is
    # only held to the round-3 floor (18 k/s from one thread through the GPU) unless the host really has the threads
    if d.get("hardware_threads", 0) >= 28:
        assert d["threads_queries_per_s"] >= 5.0e5
    else:
        assert d["threads_queries_per_s"] >= 5.0e4


def test_gcore_two_level_instancing_matches_the_flattened_scene(art, backend):
    """embree_connect.cpp:147-184: one tree per mesh + a tree over the instances (art_instanced.h).  1,000 instances of the 8-triangle
    pyramid (rotated, scaled, translated): hits against the oracle's brute-force scan of the explicitly transformed two-sided triangles,
    and against this library's own flattened upload of the same scene.  The triangle test runs in OBJECT space here and in world space
    there, so t agrees to rounding (1e-5 relative), ids exactly -- except for the handful of rays that graze an edge."""
    L = backend.lib
    F = np.float32
    ident = np.eye(4, dtype=F).ravel(); om = orc.Mesh()
    assert orc.lib().orc_load_vsgf(orc.PYRAMID_VSGF.encode(), orc.fp(ident), C.byref(om)) == 0
    pos = np.ctypeslib.as_array(om.pos, (om.nverts, 3)).copy(); idx = np.ctypeslib.as_array(om.idx, (om.ntris, 3)).copy().astype(np.int32)
    rng = np.random.default_rng(11)
    n_inst = 1000
    mats = np.zeros((n_inst, 16), F)
    for k in range(n_inst):
        a = rng.random() * 2 * np.pi; s = 0.5 + rng.random()
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) * s
        m = np.eye(4); m[:3, :3] = R; m[:3, 3] = rng.random(3) * 40 - 20
        mats[k] = m.astype(F).ravel()

    def commit(mode):
        L.gcore_set_two_level(mode)
        L.gcore_init_and_clear()
        mid = L.gcore_add_mesh_3f(np.ascontiguousarray(pos).ctypes.data_as(art.f32p), pos.shape[0], np.ascontiguousarray(idx).ctypes.data_as(art.i32p), idx.size)
        L.gcore_instance_meshes(mid, mats.ctypes.data_as(art.f32p), n_inst)
        L.gcore_commit_scene()

    n = 6000
    tgt_inst = rng.integers(0, n_inst, n)
    centres = mats.reshape(n_inst, 4, 4)[tgt_inst, :3, 3] + np.array([0, 0.15, 0], F)
    o = (centres + rng.normal(size=(n, 3)) * 6).astype(F)
    tgt = (centres + (rng.random((n, 3)) - 0.5) * 0.5).astype(F)
    d = tgt - o; d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(F)

    def query():
        hits = (art.HitCpp * n)(); found = (C.c_ubyte * n)()
        L.gcore_closest_hit_n(n, o.ctypes.data_as(art.f32p), d.ctypes.data_as(art.f32p), None, None, hits, found)
        return (np.array([found[i] for i in range(n)], bool), np.array([hits[i].instIndex for i in range(n)]), np.array([hits[i].primIndex for i in range(n)]),
                np.array([hits[i].t for i in range(n)], np.float64), np.array([list(hits[i].normal) for i in range(n)]), np.array([list(hits[i].texCoord) for i in range(n)]))
    try:
        commit(-1)                      # automatic: 1000 instances -> two-level
        two = query()
        commit(0)                       # the same scene flattened: 16,000 world-space triangles
        flat = query()
    finally:
        L.gcore_set_two_level(-1)
        L.gcore_destroy()
    # oracle on the flattened mesh (as tests/test_hydra_scene.py builds it)
    wpos, widx = [], []
    for k in range(n_inst):
        m = mats[k].reshape(4, 4)
        w = np.stack([((m[r, 0] * pos[:, 0] + m[r, 1] * pos[:, 1]).astype(F) + m[r, 2] * pos[:, 2]).astype(F) + m[r, 3] for r in range(3)], 1).astype(F)
        base = k * pos.shape[0]
        wpos.append(w)
        widx.append(np.stack([np.stack([base + idx[:, 0], base + idx[:, 1], base + idx[:, 2]], 1), np.stack([base + idx[:, 0], base + idx[:, 2], base + idx[:, 1]], 1)], 1).reshape(-1, 3))
    wpos = np.concatenate(wpos); widx = np.concatenate(widx).astype(np.int32)
    from ada_ray_tracer_amd import scenes
    mesh = dict(mode=art.MESH_CLOSEST, pos=wpos, nrm=np.zeros_like(wpos), idx=widx, matid=np.ones(widx.shape[0], np.int32))
    light = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 400.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    want = orc.closest_hits(conv.OracleScene(art.SceneDesc([], light, scenes.cornell_materials(), [mesh], None, scenes.REFERENCE_CAMERA)).scene, o, d)
    w_hit = np.array([bool(h.is_hit) and h.t < 100000.0 for h in want]); w_k = np.array([h.prim_index >> 1 for h in want]); w_t = np.array([h.t for h in want], np.float64)
    # flattened upload == oracle exactly (ids, t bits)
    assert np.array_equal(flat[0], w_hit) and np.array_equal(flat[1][w_hit], w_k[w_hit] // 8) and np.array_equal(flat[2][w_hit], w_k[w_hit] % 8)
    assert np.array_equal(flat[3][w_hit].astype(np.float32).view(np.uint32), w_t[w_hit].astype(np.float32).view(np.uint32))
    # two-level == flattened up to object-space rounding
    both = two[0] & flat[0]
    assert (two[0] != flat[0]).sum() <= n // 500, "hit / miss differs on %d rays" % (two[0] != flat[0]).sum()
    same_id = (two[1] == flat[1]) & (two[2] == flat[2])
    assert (both & ~same_id).sum() <= n // 500, "another triangle on %d rays" % (both & ~same_id).sum()
    ok = both & same_id
    assert ok.sum() > n // 3
    assert np.abs(two[3][ok] - flat[3][ok]).max() <= 2.0e-5 * np.abs(flat[3][ok]).max()
    assert np.abs(two[5][ok] - flat[5][ok]).max() < 1.0e-3                                   # barycentrics
    nn = lambda v: v / np.linalg.norm(v, axis=1, keepdims=True)
    assert np.abs(nn(two[4][ok]) - nn(flat[4][ok])).max() < 1.0e-4                            # world-space Ng


def test_structured_mesh_scene_bit_exact(art, backend):
    """scenes.structured_scene (bench.py --scene s4: a tessellated torus over a regular grid -- shared vertices, coplanar neighbours, slivers,
    interpolated vertex normals that differ per vertex) at a size the oracle's brute-force scan can follow: the accum buffer bit for bit."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.structured_scene(6000)
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    backend.resize(80, 64)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=9)
    accum, _, spp = backend.render_pass(p, 0)
    ref, _, cnt = orc.render(osc.scene, orc.make_params(80, 64, orc.PT_MIS, True, 8, 2, seed=9))
    assert_radiance_equal(accum, ref, spp)
    assert backend.stats().rays == cnt.rays


def test_gcore_single_ray_host_walk_equals_the_gpu_batch_on_edge_rays(art, backend):
    """The host walk of a single-ray gcore_closest_hit (round 4) against the GPU batch on the rays that stress the window and the slab
    arithmetic: axis-parallel directions (zero components: 1/d is replaced by +-1e30), t_near > 0 cutting the first hit off, a t_far in
    front of every triangle, an empty and an inverted window, unnormalised directions, rays starting on a triangle's plane, NaN / inf
    components.  found flags and every byte of the HitCpp must agree (two-level and flattened scene)."""
    L = backend.lib
    F = np.float32
    rng = np.random.default_rng(21)
    verts = (rng.random((900, 3)) * 4 - 2).astype(F)
    idx = rng.integers(0, 900, 2400).astype(np.int32)
    n = 4000
    o = (rng.random((n, 3)) * 6 - 3).astype(F)
    d = rng.normal(size=(n, 3)).astype(F)
    d[:600] = np.eye(3, dtype=F)[rng.integers(0, 3, 600)] * rng.choice([-1.0, 1.0], 600)[:, None]     # axis-parallel
    d[600:900, rng.integers(0, 3)] = 0.0                                                                    # one zero component
    d[900:1200] *= 7.5                                                                                      # unnormalised
    tn = np.zeros(n, F); tf = np.full(n, 1.0e5, F)
    tn[1200:1800] = rng.random(600).astype(F) * 3.0                                                         # near cut
    tf[1800:2400] = rng.random(600).astype(F) * 0.5                                                         # far cut
    tn[2400:2500] = 2.0; tf[2400:2500] = 2.0                                                                # empty window
    tn[2500:2600] = 3.0; tf[2500:2600] = 1.0                                                                # inverted window
    tri = idx[:300].reshape(-1, 3)
    o[2600:2700] = verts[tri[:, 0]] * F(0.4) + verts[tri[:, 1]] * F(0.3) + verts[tri[:, 2]] * F(0.3)       # origin on a triangle
    d[2700:2710, 0] = np.nan; d[2710:2720, 1] = np.inf; o[2720:2730, 2] = np.nan
    for two_level, mats in ((0, np.eye(4, dtype=F)[None]), (1, np.stack([np.eye(4, dtype=F) + np.array([[0, 0, 0, 5.0 * k], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0]], F) for k in range(3)]))):
        L.gcore_set_two_level(two_level)
        try:
            L.gcore_init_and_clear()
            mid = L.gcore_add_mesh_3f(verts.ctypes.data_as(art.f32p), 900, idx.ctypes.data_as(art.i32p), 2400)
            mm = np.ascontiguousarray(mats.reshape(-1, 16))
            L.gcore_instance_meshes(mid, mm.ctypes.data_as(art.f32p), mm.shape[0])
            L.gcore_commit_scene()
            hits = (art.HitCpp * n)(); found = (C.c_ubyte * n)()
            L.gcore_closest_hit_n(n, o.ctypes.data_as(art.f32p), d.ctypes.data_as(art.f32p), tn.ctypes.data_as(art.f32p), tf.ctypes.data_as(art.f32p), hits, found)
            batch = np.frombuffer(hits, np.uint8).reshape(n, C.sizeof(art.HitCpp)).copy()
            one = np.zeros_like(batch); f1 = np.zeros(n, np.uint8)
            for i in range(n):
                h = art.HitCpp()
                f1[i] = 1 if L.gcore_closest_hit(o[i].ctypes.data_as(art.f32p), d[i].ctypes.data_as(art.f32p), float(tn[i]), float(tf[i]), C.byref(h)) else 0
                if f1[i]:
                    one[i] = np.frombuffer(h, np.uint8)
            fb = np.array([found[i] for i in range(n)], np.uint8)
            assert np.array_equal(f1, fb), "found differs on %d rays" % (f1 != fb).sum()
            assert np.array_equal(one[f1 == 1], batch[f1 == 1])
            assert f1[:1200].sum() > 100 and f1[2400:2600].sum() == 0 and f1[2700:2730].sum() == 0
        finally:
            L.gcore_set_two_level(-1)
            L.gcore_destroy()


def test_more_spheres_and_lights_than_the_stages_keep_in_lds(art, backend):
    """The stages keep up to 64 spheres and 8 (shade) / 16 (raygen) lights in LDS and read the scene's own tables otherwise.  A scene with 70
    spheres and 10 sphere lights + a BVH mesh takes the global-table path in every stage: the accum buffer bit for bit, for MIS and the
    shadow-ray integrator."""
    from ada_ray_tracer_amd import scenes
    rng = np.random.default_rng(5)
    mats = scenes.cornell_materials()
    lights, spheres = [], []
    for k in range(10):
        mats.append(dict(type=art.MAT_LIGHT, light=k)) if k else mats.__setitem__(4, dict(type=art.MAT_LIGHT, light=0))
        m = 4 if k == 0 else len(mats) - 1
        l = scenes.sphere_light(-2.0 + 0.44 * k, m, cy=4.4, cz=1.0 + 0.3 * k, radius=0.12)
        lights.append(l); spheres.append((l["center"], l["radius"], m))
    for k in range(60):
        p = (float(-2.1 + 4.2 * rng.random()), float(0.3 + 3.2 * rng.random()), float(0.4 + 4.0 * rng.random()))
        spheres.append((p, 0.12, (0, 1, 2, 3, 8)[k % 5]))
    mesh = scenes.random_triangles(1500, 0xADA5EED0 + 33)
    sd = art.SceneDesc(spheres=spheres, lights=lights, materials=mats, meshes=[mesh], cornell=scenes.CORNELL_BOX, cam_pos=scenes.REFERENCE_CAMERA)
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    for rt in ("PT_MIS", "PT_SHADOW"):
        backend.resize(72, 56)
        accum, _, spp = backend.render_pass(art.Backend.pass_params(getattr(art, rt), True, 6, 2, seed=13), 0)
        ref, _, cnt = orc.render(osc.scene, orc.make_params(72, 56, getattr(orc, rt), True, 6, 2, seed=13))
        assert_radiance_equal(accum, ref, spp)
        assert backend.stats().lost_paths == 0


@pytest.mark.parametrize("scene", ["c3_small", "mixed", "instanced"])
def test_skip_null_shadow_option_keeps_the_picture_and_traces_fewer_rays(art, backend, scene):
    """Option skip_null_shadow (off by default: the reference calls Compute_Shadow for every surface hit, integrators.adb:270): shadow rays
    whose explicit colour is exactly zero under either verdict are not traced -- same bits in the frame, fewer rays in the counter"""
    from ada_ray_tracer_amd import scenes
    sd = {"c3_small": lambda: scenes.synthetic_scene(20000, 3), "mixed": lambda: scenes.mixed_scene(5000, 5), "instanced": lambda: scenes.instanced_scene(12, 300)}[scene]()
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=13)
    backend.upload_scene(sd); backend.resize(160, 120)
    r0 = backend.stats().rays
    ref, _, _ = backend.render_pass(p, 0)
    rays_ref = backend.stats().rays - r0
    backend.set_option("skip_null_shadow", 1)
    try:
        backend.resize(160, 120)
        r0 = backend.stats().rays
        got, _, _ = backend.render_pass(p, 0)
        rays = backend.stats().rays - r0
    finally:
        backend.set_option("skip_null_shadow", 0)
    assert np.array_equal(bits(got), bits(ref)) and backend.stats().lost_paths == 0
    assert 0 < rays < rays_ref


def test_a_lost_path_in_the_first_batch_fails_the_call_that_waits(art, backend):
    """the self-check counter (ArtStats::lost_paths) is compared with a baseline that only the synchronising call advances.  Until
    round 5 the items-per-thread trial of the shade stage read the counter in the first two batches after every upload / resize and absorbed
    a loss there -- i.e. in the only batches a small render has.  Option inject_lost bumps the counter after bounce 0 of the next pass' first
    batch, as a stage that lost a path would; the pass must fail, once, and the backend must keep working."""
    from ada_ray_tracer_amd import scenes
    backend.upload_scene(scenes.synthetic_scene(2000, 3)); backend.resize(64, 48)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=3)
    backend.set_option("inject_lost", 1)
    with pytest.raises(art.ArtError, match="1 path.s. lost"):
        backend.render_pass(p, 0)
    ref, _, spp = backend.render_pass(p, 4)                      # the same loss is not reported twice; the next pass is fine
    assert spp == 8 and backend.stats().lost_paths == 1
    backend.set_option("inject_lost", 1)                         # ... and in a later batch of a render, after passes that synchronised
    with pytest.raises(art.ArtError, match="1 path.s. lost"):
        backend.render_pass_device(p, spp)
    backend.resize(64, 48)                                       # (zeroes the counters for the tests that follow)
    assert backend.stats().lost_paths == 0


@pytest.mark.parametrize("option,value", [("paths_spread", 2), ("paths_spread", 64), ("paths_spread", 0), ("paths_spread_holes", 1), ("paths_contiguous", 1), ("hot_pad", 1088), ("hot_pad", 64)])
def test_where_the_path_state_lives_does_not_change_the_picture(art, backend, option, value):
    """Round 6 (profiles/r6_bimodal): the path state as one address range over separately created physical chunks (paths_spread = chunk MB;
    the HIP virtual-memory calls -- the default for path states of a gigabyte or more, forced here on a small one; 0 = plain hipMalloc), with
    spacer chunks between them (paths_spread_holes), as physically contiguous memory (paths_contiguous), with a pad between the fields of a
    bank's block (hot_pad, items): placement options -- the same bits in the frame, the same rays, and the backend goes back to its default
    afterwards."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(20000, 3)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=11)
    backend.upload_scene(sd); backend.resize(200, 150)
    r0 = backend.stats().rays
    ref, _, _ = backend.render_pass(p, 0)
    rays_ref = backend.stats().rays - r0
    if option == "paths_spread_holes":
        backend.set_option("paths_spread", 2)
    backend.set_option(option, value)
    try:
        backend.resize(200, 150)
        r0 = backend.stats().rays
        got, _, spp = backend.render_pass(p, 0)
        rays = backend.stats().rays - r0
        got2, _, _ = backend.render_pass(p, spp)                     # (a second pass re-uses the allocation)
    finally:
        backend.set_option(option, -1 if option == "paths_spread" else 0)
        backend.set_option("paths_spread", -1)
    assert spp == 8 and rays == rays_ref and backend.stats().lost_paths == 0
    assert np.array_equal(bits(got), bits(ref))
    backend.resize(200, 150)
    again, _, _ = backend.render_pass(p, 0)
    assert np.array_equal(bits(again), bits(ref))


def test_a_failing_chunk_of_the_path_state_is_undone_and_hipmalloc_takes_over(art):
    """The default backing of a large path state is a reserved address range over separately created chunks (art_api.cpp alloc_spread).  When
    the device cannot give a chunk -- test option spread_fail_at: chunk 3 of a path state forced into 2 MB chunks -- everything created so
    far must be released, the range given back, and the pass must run from a plain hipMalloc with the same picture.  In a child process:
    which backing a batch got is only visible on the library's debug line (ART_DEBUG_ADDR)."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as ge
art = ge.load_package()
from ada_ray_tracer_amd import scenes
be = art.Backend(0)
be.upload_scene(scenes.synthetic_scene(20000, 3)); be.resize(320, 200)
p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=11)
out = {}
for name, opts in (("hipmalloc", {"paths_spread": 0}), ("chunks", {"paths_spread": 2}), ("chunk_3_fails", {"paths_spread": 2, "spread_fail_at": 3}), ("chunk_0_fails", {"paths_spread": 2, "spread_fail_at": 0})):
    for k, v in opts.items():
        be.set_option(k, v)
    be.resize(320, 200)
    print("CASE", name, file=sys.stderr, flush=True)
    acc, _, spp = be.render_pass(p, 0)
    out[name] = [int(np.ascontiguousarray(acc).view(np.uint32).sum(dtype=np.uint64)), int(be.stats().rays), int(be.stats().lost_paths)]
    be.set_option("spread_fail_at", -1)
be.shutdown()
print(json.dumps(out))
'''
    r = subprocess.run([sys.executable, "-c", code, art.ROOT], capture_output=True, text=True, timeout=600, env=dict(os.environ, ART_DEBUG_ADDR="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["hipmalloc"] == out["chunks"] == out["chunk_3_fails"] == out["chunk_0_fails"] and out["hipmalloc"][2] == 0 and out["hipmalloc"][1] > 0
    got = {}
    case = None
    for line in r.stderr.splitlines():
        if line.startswith("CASE "):
            case = line.split()[1]
        elif line.startswith("ART_DEBUG_ADDR spread ") and case:
            got[case] = int(line.split()[2])
    assert got == {"hipmalloc": 0, "chunks": 1, "chunk_3_fails": 0, "chunk_0_fails": 0}, (got, r.stderr[-1500:])

"""Instanced scenes through the C ABI (round 5; SURVEY 8(f) rank 2, embree_connect.cpp:147-184): art_upload_scene takes meshes + 3x4
instance transforms, art_render_pass walks the two-level tree (k_trace_inst) without flattening, and the picture is the FLATTENED
scene's, bit for bit.  The reference: the product's own render of the explicitly flattened mesh (itself oracle-checked at this size
class by tests/test_gpu_stated_spp.py) AND the oracle on sampled pixels, its mesh search walking the flattened scene's exported tree."""
import numpy as np
import pytest

import conv
import hostsim
import orc

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("rt", ["PT_MIS", "PT_SHADOW", "PT_STUPID"])
@pytest.mark.parametrize("kernel", ["coop", "coop_stack_cap_3", "one_ray_per_lane"])
def test_small_instanced_scene_equals_the_oracle_on_the_flattened_scene(art, backend, rt, kernel):
    """12 instances of two ~300-triangle meshes: the oracle's O(N) scan of the flattened mesh, whole frame, every integrator -- through the
    cooperative kernel crossing the instance boundary (k_trace_coop<.., INST>), the same with its LDS stack capped at 3 entries (rays that
    do not fit -- the "leave" marker included -- finish in k_trace_overflow's two-level search), and through k_trace_inst (option inst_coop = 0)."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(12, 300)
    flat = hostsim.flattened_copy(art, sd)
    p = art.Backend.pass_params(getattr(art, rt), True, 8, 2, seed=21)
    backend.set_option("inst_coop", 0 if kernel == "one_ray_per_lane" else 1)
    backend.set_option("lds_stack_cap", 3 if kernel == "coop_stack_cap_3" else 0)
    try:
        backend.upload_scene(sd); backend.resize(96, 80)
        accum, _, spp = backend.render_pass(p, 0)
    finally:
        backend.set_option("inst_coop", 1); backend.set_option("lds_stack_cap", 0)
    rays = backend.stats().rays
    ref, _, cnt = orc.render(conv.OracleScene(flat).scene, orc.make_params(96, 80, getattr(orc, rt), True, 8, 2, seed=21))
    assert spp == 8 and rays == cnt.rays and backend.stats().lost_paths == 0
    assert np.array_equal(bits(accum), bits(ref))


@pytest.mark.parametrize("inst_open", [0, 1, 8, 1000])
@pytest.mark.parametrize("kernel", ["coop", "coop_stack_cap_3", "one_ray_per_lane"])
def test_mirrored_sheared_coincident_tiny_and_huge_instances(art, backend, kernel, inst_open):
    """hostsim.awkward_instances (a mirror image, one transform twice -- equal t, the lower hit index wins --, a shear, scales 1e-3 and 2.2,
    interpenetrating instances; glass, mirror and Phong triangles): the oracle's O(N) scan of the flattened mesh, whole frame, and the
    hit indices of the debug pass against the flattened upload's.  inst_open: the instance tree ends at whole instances (1; 0, the default, chooses: 1 here), at about 8
    subtrees per instance (the default), at every leaf of the meshes' trees (1000 asked for: an entry point may then BE a leaf)."""
    from ada_ray_tracer_amd import scenes
    tr = hostsim.awkward_instances()
    sd = scenes.instanced_scene(0, 260, transforms=tr, all_materials=True)
    flat = hostsim.flattened_copy(art, sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=77)
    backend.set_option("inst_coop", 0 if kernel == "one_ray_per_lane" else 1)
    backend.set_option("lds_stack_cap", 3 if kernel == "coop_stack_cap_3" else 0)
    backend.set_option("inst_open", inst_open)
    try:
        backend.upload_scene(sd); backend.resize(128, 96)
        accum, _, spp = backend.render_pass(p, 0)
        rays = backend.stats().rays
        dbg = backend.debug_hit_pass(art.Backend.pass_params(art.RT_DEBUG, False, 8, 1))
    finally:
        backend.set_option("inst_coop", 1); backend.set_option("lds_stack_cap", 0); backend.set_option("inst_open", 0)
    ref, _, cnt = orc.render(conv.OracleScene(flat).scene, orc.make_params(128, 96, orc.PT_MIS, True, 8, 2, seed=77))
    assert spp == 8 and rays == cnt.rays and backend.stats().lost_paths == 0
    assert np.array_equal(bits(accum), bits(ref))
    backend.upload_scene(flat); backend.resize(128, 96)
    ref_dbg = backend.debug_hit_pass(art.Backend.pass_params(art.RT_DEBUG, False, 8, 1))
    ntris = [sd.desc.meshes[mi].ntris for mi, _ in tr]
    offs = np.concatenate([[0], np.cumsum(ntris)])
    shift = int(np.ceil(np.log2(max(ntris))))
    prim, mat, ptype = dbg[2], dbg[3], dbg[4]; rprim = ref_dbg[2]
    assert np.array_equal(bits(dbg[0]), bits(ref_dbg[0])) and np.array_equal(mat, ref_dbg[3]) and np.array_equal(ptype, ref_dbg[4])
    on_mesh = (ptype == 2)                                     # mesh triangles: instance << shift | triangle of the mesh here, the position in the flattened list there
    assert np.array_equal(prim[~on_mesh], rprim[~on_mesh]) and on_mesh.sum() > 500
    inst = prim[on_mesh] >> shift
    assert np.array_equal(offs[inst] + (prim[on_mesh] & ((1 << shift) - 1)), rprim[on_mesh])
    hit_insts = set(np.unique(inst).tolist())
    assert 3 not in hit_insts and 2 in hit_insts              # the coincident pair: always the first of the two
    assert {0, 1, 4, 6}.issubset(hit_insts)


def test_interpenetrating_cluster_is_opened_by_the_build_and_renders_the_flattened_picture(art, backend):
    """24 instances of two ~2 k-triangle meshes pulled into one cluster: the build's own rule opens them (more instance-tree nodes than with
    inst_open = 1); the picture is the one of whole instances and of the flattened upload, whole frame, and the oracle's on sampled pixels"""
    from ada_ray_tracer_amd import scenes
    W, H = 320, 180
    sd = scenes.instanced_cluster(24, 2000)
    flat = hostsim.flattened_copy(art, sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=9)
    backend.upload_scene(sd); backend.resize(W, H)
    n_auto = backend.bvh_info().n_nodes
    auto, _, spp = backend.render_pass(p, 0)
    rays = backend.stats().rays
    assert spp == 8 and backend.stats().lost_paths == 0
    backend.set_option("inst_open", 1)
    try:
        backend.upload_scene(sd); backend.resize(W, H)
        n_whole = backend.bvh_info().n_nodes
        whole, _, _ = backend.render_pass(p, 0)
    finally:
        backend.set_option("inst_open", 0)
    assert n_auto > n_whole + 24                               # the instance tree grew: entry points were added
    assert np.array_equal(bits(auto), bits(whole))
    backend.upload_scene(flat); backend.resize(W, H)
    ref, _, _ = backend.render_pass(p, 0)
    assert backend.stats().rays == rays and np.array_equal(bits(auto), bits(ref))
    osc = conv.OracleScene(flat)
    nodes, tris, info = backend.export_bvh()
    osc.attach_bvh(nodes, tris, info.node_width)
    rng = np.random.default_rng(24)
    xs = rng.integers(0, W, 200); ys = rng.integers(0, H, 200)
    oref, _ = orc.render_pixels(osc.scene, orc.make_params(W, H, orc.PT_MIS, True, 8, 2, seed=9), xs, ys)
    assert np.array_equal(bits(auto[ys, xs]), bits(oref))


def test_four_thousand_small_instances(art, backend):
    """4096 instances of two ~300-triangle meshes crowding the box (opened into entry points by the build's own rule; a 12-bit instance index
    over a 9-bit triangle index in the hit key): the cooperative kernel's picture == the flattened 1.2 M-triangle upload's, whole frame"""
    from ada_ray_tracer_amd import scenes
    W, H = 256, 144
    sd = scenes.instanced_scene(4096, 300)
    flat = hostsim.flattened_copy(art, sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=9)
    backend.upload_scene(sd); backend.resize(W, H)
    n_auto = backend.bvh_info().n_nodes
    inst, _, spp = backend.render_pass(p, 0)
    rays = backend.stats().rays
    assert spp == 4 and backend.stats().lost_paths == 0
    backend.set_option("inst_open", 1)
    try:
        backend.upload_scene(sd)
        assert backend.bvh_info().n_nodes < n_auto           # (the rule did open them)
    finally:
        backend.set_option("inst_open", 0)
    backend.upload_scene(flat); backend.resize(W, H)
    ref, _, _ = backend.render_pass(p, 0)
    assert backend.stats().rays == rays and np.array_equal(bits(inst), bits(ref))


def test_64_instances_of_20k_triangles_at_64_spp(art, backend):
    """The review's case: 64 instances x ~20 k triangles (1.28 M triangles flattened), 640x360, PT_MIS depth 8, 64 spp.  Instanced render ==
    render of the flattened upload (whole frame, bits), and == the oracle on 300 sampled pixels (its search walks the flattened tree)."""
    from ada_ray_tracer_amd import scenes
    W, H, T = 640, 360, 16
    sd = scenes.instanced_scene(64, 20000)
    flat = hostsim.flattened_copy(art, sd)
    assert flat.desc.meshes[0].ntris > 1200000
    p = art.Backend.pass_params(art.PT_MIS, True, 8, T, seed=5)
    backend.upload_scene(sd); backend.resize(W, H)
    inst, _, spp = backend.render_pass(p, 0)
    rays_inst = backend.stats().rays
    assert spp == 64 and backend.stats().lost_paths == 0
    backend.set_option("inst_coop", 0)                        # the one-ray-per-lane two-level kernel: the same picture
    try:
        backend.resize(W, H)
        slow, _, _ = backend.render_pass(p, 0)
    finally:
        backend.set_option("inst_coop", 1)
    assert np.array_equal(bits(inst), bits(slow))
    backend.upload_scene(flat); backend.resize(W, H)
    ref, _, _ = backend.render_pass(p, 0)
    assert backend.stats().rays == rays_inst
    assert np.array_equal(bits(inst), bits(ref))
    osc = conv.OracleScene(flat)
    nodes, tris, info = backend.export_bvh()
    osc.attach_bvh(nodes, tris, info.node_width)
    rng = np.random.default_rng(64)
    xs = rng.integers(0, W, 300); ys = rng.integers(0, H, 300)
    oref, _ = orc.render_pixels(osc.scene, orc.make_params(W, H, orc.PT_MIS, True, 8, T, seed=5), xs, ys)
    assert np.array_equal(bits(inst[ys, xs]), bits(oref))
    assert (oref.sum(1) > 0).mean() > 0.3


def test_instanced_scene_needs_the_record_schedule(art, backend):
    """the one-ray-per-lane cross-check kernel walks a single tree: an instanced scene is refused there, with a message"""
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(4, 100)
    backend.upload_scene(sd); backend.resize(32, 32)
    backend.set_option("trace_kernel", art.TRACE_SIMPLE)
    try:
        with pytest.raises(art.ArtError, match="instanced"):
            backend.render_pass(art.Backend.pass_params(art.PT_MIS, True, 4, 1, seed=1), 0)
    finally:
        backend.set_option("trace_kernel", art.TRACE_COOP)


def test_ray_queries_and_the_debug_pass_see_the_flattened_scene(art, backend):
    """art_trace_rays (both trace kernels) and Debug_Ray_Tracing over an instanced scene: the hits of the flattened upload -- t, u, v, the
    material and the world-space shading normal the same bits; the triangle index is instance << shift | triangle of the mesh there and
    the position in the flattened list here."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(10, 200)
    flat = hostsim.flattened_copy(art, sd)
    rng = np.random.default_rng(3)
    n = 20000
    o = np.tile(np.array([0.0, 2.55, 12.5], np.float32), (n, 1)) + rng.normal(0, 0.05, (n, 3)).astype(np.float32)
    centres = np.array([[sd.desc.instances[i].m[3], sd.desc.instances[i].m[7], sd.desc.instances[i].m[11]] for i in range(10)], np.float32)
    d = (centres[rng.integers(0, 10, n)] + rng.normal(0, 0.25, (n, 3)).astype(np.float32) - o).astype(np.float32)      # towards the instances
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    ntris = [sd.desc.meshes[sd.desc.instances[i].mesh].ntris for i in range(10)]
    offs = np.concatenate([[0], np.cumsum(ntris)])
    shift = int(np.ceil(np.log2(max(ntris))))
    dbg = art.Backend.pass_params(art.RT_DEBUG, False, 8, 1)
    backend.upload_scene(flat); backend.resize(80, 60)
    ref = {k: backend.trace_rays(o, d, kernel=k) for k in (art.TRACE_COOP, art.TRACE_SIMPLE)}
    ref_dbg = backend.debug_hit_pass(dbg)
    backend.upload_scene(sd); backend.resize(80, 60)
    got_dbg = backend.debug_hit_pass(dbg)
    for k in (art.TRACE_COOP, art.TRACE_SIMPLE):
        got = backend.trace_rays(o, d, kernel=k)
        tri_hits = 0
        for i in range(n):
            a, b = got[i], ref[k][i]
            assert (a.is_hit, a.prim_type, a.mat_id, a.mat) == (b.is_hit, b.prim_type, b.mat_id, b.mat)
            if not a.is_hit:
                continue
            assert bits([a.t, a.u, a.v] + list(a.normal)).tolist() == bits([b.t, b.u, b.v] + list(b.normal)).tolist()
            if a.prim_type == 2:
                tri_hits += 1
                assert offs[a.prim_index >> shift] + (a.prim_index & ((1 << shift) - 1)) == b.prim_index
            else:
                assert a.prim_index == b.prim_index
        assert tri_hits > n // 10
    assert np.array_equal(bits(got_dbg[0]), bits(ref_dbg[0])) and np.array_equal(got_dbg[3], ref_dbg[3]) and np.array_equal(got_dbg[4], ref_dbg[4])


@pytest.mark.parametrize("kernel", ["coop", "coop_stack_cap_3", "one_ray_per_lane"])
@pytest.mark.parametrize("view", [(2.0, 1.0), (1.5, 0.1)])
def test_a_speck_far_from_the_origin_at_full_resolution(art, backend, kernel, view):
    """ADVICE r5 (tests/test_instanced_host_sim.py has the CPU twin and the arithmetic): one instance of scale 5e-4 far from the origin, seen
    from two speck sizes away at 1024 x 1024 -- grazing rays along every triangle edge and box face.  The ray taken to object space is 6e-4
    object units off in binary32; the meshes' box pad follows the instances since round 6, so all three trace kernels still give the
    flattened upload's picture bit for bit (until round 5: holes)."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.speck_scene(view=view)
    flat = hostsim.flattened_copy(art, sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 6, 1, seed=3)
    backend.upload_scene(flat); backend.resize(1024, 1024)
    r0 = backend.stats().rays
    ref, _, _ = backend.render_pass(p, 0)
    rays_ref = backend.stats().rays - r0
    backend.set_option("inst_coop", 0 if kernel == "one_ray_per_lane" else 1)
    backend.set_option("lds_stack_cap", 3 if kernel == "coop_stack_cap_3" else 0)
    try:
        backend.upload_scene(sd); backend.resize(1024, 1024)
        r0 = backend.stats().rays
        accum, _, spp = backend.render_pass(p, 0)
        rays = backend.stats().rays - r0
    finally:
        backend.set_option("inst_coop", 1); backend.set_option("lds_stack_cap", 0)
    assert spp == 4 and rays == rays_ref and backend.stats().lost_paths == 0
    assert np.array_equal(bits(accum), bits(ref)) and (accum > 0).mean() > 0.5
    # ... and sampled pixels against the oracle's O(N) scan of the flattened mesh
    rng = np.random.default_rng(5)
    xs = rng.integers(0, 1024, 300); ys = rng.integers(0, 1024, 300)
    want, _ = orc.render_pixels(conv.OracleScene(flat).scene, orc.make_params(1024, 1024, orc.PT_MIS, True, 6, 1, seed=3), xs, ys)
    assert np.array_equal(bits(accum[ys, xs]), bits(want))

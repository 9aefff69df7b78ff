"""Both node widths of the cooperative trace kernel -- 4 lanes per ray with 4-wide nodes (the default: fewer lane-steps per ray)
and 8 lanes per ray with 8-wide nodes -- and the capped LDS stack with its overflow path (k_trace_overflow), against the CPU oracle."""
import numpy as np
import pytest

import conv
import orc
from test_gpu_parity import _assert_hits_equal, _random_rays, assert_radiance_equal, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[4, 8])
def wbackend(backend, request):
    backend.set_option("bvh_width", request.param)
    yield backend
    backend.set_option("bvh_width", 4)
    backend.set_option("lds_stack_cap", 0)


@pytest.mark.parametrize("ntris", [1, 3, 7, 300, 20000])
def test_hits_match_brute_force(art, wbackend, ntris):
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(ntris, 3)
    osc = conv.OracleScene(sd)
    wbackend.upload_scene(sd)
    o, d = _random_rays(30000, ntris + 2)
    want = orc.closest_hits(osc.scene, o, d)
    for kernel in (art.TRACE_COOP, art.TRACE_SIMPLE):
        _assert_hits_equal(wbackend.trace_rays(o, d, kernel=kernel), want)


def test_counters_match_oracle_walk(art, wbackend):
    from ada_ray_tracer_amd import scenes
    mesh = scenes.random_triangles(20000, 77)
    lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    sd = art.SceneDesc([], lights, scenes.cornell_materials(), [mesh], None, scenes.REFERENCE_CAMERA)
    wbackend.upload_scene(sd)
    nodes, tris, info = wbackend.export_bvh()
    assert nodes.size == info.n_nodes * 8 * info.node_width
    o, d = _random_rays(40000, 8)
    d[:100, 0] = 0.0
    t, prim, cnt = orc.bvh_walk(nodes, tris, o, d, width=info.node_width)
    for kernel in (art.TRACE_COOP, art.TRACE_SIMPLE):
        hits, st = wbackend.trace_rays(o, d, kernel=kernel, want_stats=True)
        gprim = np.array([h.prim_index if h.is_hit else -1 for h in hits], np.int32)
        assert np.array_equal(gprim, prim)
        assert (st.box_tests, st.tri_tests, st.node_visits, st.leaf_visits, st.traced_rays) == \
               (cnt.box_tests, cnt.tri_tests, cnt.node_visits, cnt.leaf_visits, cnt.rays)


@pytest.mark.parametrize("rt", ["PT_MIS", "PT_SHADOW", "PT_STUPID"])
def test_render_bit_exact(art, wbackend, rt):
    from ada_ray_tracer_amd import scenes
    for sd, seed in ((scenes.synthetic_scene(2000, 3), 3), (scenes.mixed_scene(1500, 5), 6)):
        osc = conv.OracleScene(sd)
        wbackend.upload_scene(sd)
        wbackend.resize(64, 64)
        accum, _, spp = wbackend.render_pass(art.Backend.pass_params(getattr(art, rt), True, 8, 1, seed=seed), 0)
        ref, _, cnt = orc.render(osc.scene, orc.make_params(64, 64, getattr(orc, rt), True, 8, 1, seed=seed))
        assert_radiance_equal(accum, ref, spp)
        assert wbackend.stats().rays == cnt.rays


@pytest.mark.parametrize("cap", [2, 5, 9])
def test_capped_lds_stack_overflow_path(art, wbackend, cap):
    """A tiny LDS stack sends most rays through the overflow queue and k_trace_overflow: same hits, same image."""
    from ada_ray_tracer_amd import scenes
    wbackend.set_option("lds_stack_cap", cap)
    sd = scenes.synthetic_scene(20000, 3)
    osc = conv.OracleScene(sd)
    wbackend.upload_scene(sd)
    o, d = _random_rays(30000, 78)
    _assert_hits_equal(wbackend.trace_rays(o, d), orc.closest_hits(osc.scene, o, d))
    sd = scenes.synthetic_scene(2000, 3)
    osc = conv.OracleScene(sd)
    wbackend.upload_scene(sd)
    wbackend.resize(64, 64)
    accum, _, spp = wbackend.render_pass(art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=3), 0)
    ref, _, cnt = orc.render(osc.scene, orc.make_params(64, 64, orc.PT_MIS, True, 8, 1, seed=3))
    assert_radiance_equal(accum, ref, spp)
    assert wbackend.stats().rays == cnt.rays


def test_full_size_scene_same_image_for_both_widths(art, backend):
    """C4 (1M triangles), reduced frame: the image does not depend on the node width (nor on the overflow path, which the
    width-4 tree of this scene uses: its stack bound exceeds the LDS stack)."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(1000000, 4)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 4, seed=5)
    out = []
    try:
        for w in (4, 8):
            backend.set_option("bvh_width", w)
            backend.upload_scene(sd)
            backend.resize(320, 180)
            accum, _, spp = backend.render_pass(p, 0)
            out.append((accum.copy(), backend.stats().rays))
    finally:
        backend.set_option("bvh_width", 4)
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and out[0][1] == out[1][1]


def test_refill_and_chunk_options_do_not_change_results(art, backend):
    """Round 2: rays reach the trace kernel as 64-byte trace records claimed in chunks (ray_chunk, a multiple of 16, all prefetched) and
    idle ray groups refill when refill_min of them are idle.  Neither may change a bit of the image, the ray count or the traversal
    counters; a chunk size the prefetch cannot cover is refused."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(5000, 3)
    backend.upload_scene(sd)
    backend.resize(64, 48)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=11)
    r0 = backend.stats().rays
    ref, _, _ = backend.render_pass(p, 0)
    rays = backend.stats().rays - r0
    with pytest.raises(art.ArtError):
        backend.set_option("ray_chunk", 24)
    with pytest.raises(art.ArtError):
        backend.set_option("refill_min", 0)
    try:
        for chunk, refill in [(16, 1), (32, 3), (4096, 8), (48, 16 // 2)]:
            backend.set_option("ray_chunk", chunk)
            backend.set_option("refill_min", refill)
            backend.resize(64, 48)
            r0 = backend.stats().rays
            img, _, _ = backend.render_pass(p, 0)
            assert np.array_equal(bits(img), bits(ref)), (chunk, refill)
            assert backend.stats().rays - r0 == rays
    finally:
        backend.set_option("ray_chunk", 48)
        backend.set_option("refill_min", 2)


def test_node_min_and_refill_gate_cannot_stall_a_wave(art, backend):
    """with node_min = 8 (every group of a wave must want a node) or, at width 8, refill_min >= 6, a wave could sit with
    too few groups for the node phase, too few idle groups for the refill gate and rays still queued: it spun forever.  The node
    phase now only yields to a refill that will really happen.  Both settings, both widths, bit-equal to the defaults.  (A hang
    would be killed by the test's own timeout, not by the box.)"""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(3000, 3)
    p = art.Backend.pass_params(art.PT_MIS, True, 6, 1, seed=5)
    ref = None
    try:
        for width in (4, 8):
            backend.set_option("bvh_width", width)
            backend.upload_scene(sd)
            for node_min, refill in [(4, 2), (8, 2), (4, 8), (8, 8), (8, 6)]:
                backend.set_option("node_min", node_min)
                backend.set_option("refill_min", refill)
                backend.resize(40, 32)
                img, _, _ = backend.render_pass(p, 0)
                if ref is None:
                    ref = img
                assert np.array_equal(bits(img), bits(ref)), (width, node_min, refill)
    finally:
        backend.set_option("bvh_width", 4)
        backend.set_option("node_min", 0)
        backend.set_option("refill_min", 2)


def test_shadow_rays_as_full_closest_hit_searches_give_the_same_image(art, backend):
    """Option shadow_anyhit = 0: shadow rays carry no 10*eps in their trace records and run as full closest-hit searches
    (Compute_Shadow's own formulation, ray_tracer.adb:100-132) instead of under the visibility rule.  Same decision, so the same image and
    ray count -- through the record-writing stages, both widths, and a capped LDS stack (k_trace_overflow restarts from the record)."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.mixed_scene(4000, 5)
    osc = conv.OracleScene(sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=9)
    ref, _, cnt = orc.render(osc.scene, orc.make_params(48, 40, orc.PT_MIS, True, 8, 1, seed=9))
    try:
        for width in (4, 8):
            backend.set_option("bvh_width", width)
            backend.upload_scene(sd)
            for anyhit, cap in [(1, 0), (0, 0), (0, 5), (1, 5)]:
                backend.set_option("shadow_anyhit", anyhit)
                backend.set_option("lds_stack_cap", cap)
                backend.resize(48, 40)
                r0 = backend.stats().rays
                img, _, _ = backend.render_pass(p, 0)
                assert np.array_equal(bits(img), bits(ref)), (width, anyhit, cap)
                assert backend.stats().rays - r0 == cnt.rays
    finally:
        backend.set_option("bvh_width", 4)
        backend.set_option("shadow_anyhit", 1)
        backend.set_option("lds_stack_cap", 0)


def test_trace_kernel_keeps_its_exec_contract(art, backend, tmp_path):
    """the exec-masked inline asm of k_trace_coop's node step (pop, node load, LDS push, sort key: `s_mov exec, mask ... s_mov
    exec, -1`) is only right when it is entered with all 64 lanes enabled.  libart_hip_check.so is the same library with -DART_CHECK_EXEC:
    the kernel traps if EXEC != -1 at the head of a node step.  A render of the mixed scene (both widths, a capped LDS stack with the
    overflow kernel, shadow rays with and without the visibility rule) in a process of its own must finish and give the product's bits."""
    import os
    import subprocess
    import sys
    lib = os.path.join(art.PKG_DIR, "libart_hip_check.so")
    assert os.path.exists(lib), "make -C ada-ray-tracer_amd builds it (__graft_entry__.build)"
    from ada_ray_tracer_amd import scenes
    sd = scenes.mixed_scene(4000, 5)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=9)
    want = {}
    try:
        for width, cap in ((4, 0), (4, 5), (8, 0)):
            backend.set_option("bvh_width", width); backend.set_option("lds_stack_cap", cap)
            backend.upload_scene(sd); backend.resize(96, 64)
            want[(width, cap)], _, _ = backend.render_pass(p, 0)
    finally:
        backend.set_option("bvh_width", 4); backend.set_option("lds_stack_cap", 0)
    out = str(tmp_path / "check.npz")
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import __graft_entry__ as ge\n"
        "art = ge.load_package()\n"
        "from ada_ray_tracer_amd import scenes\n"
        "be = art.Backend(0)\n"
        "sd = scenes.mixed_scene(4000, 5)\n"
        "p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=9)\n"
        "res = {}\n"
        "for width, cap in ((4, 0), (4, 5), (8, 0)):\n"
        "    be.set_option('bvh_width', width); be.set_option('lds_stack_cap', cap)\n"
        "    be.upload_scene(sd); be.resize(96, 64)\n"
        "    res['w%%d_c%%d' %% (width, cap)], _, _ = be.render_pass(p, 0)\n"
        "np.savez(%r, **res)\n"
        "be.shutdown()\n" % (art.ROOT, out))
    env = dict(os.environ, ART_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.load(out)
    for (width, cap), img in want.items():
        assert np.array_equal(bits(got["w%d_c%d" % (width, cap)]), bits(img)), (width, cap)

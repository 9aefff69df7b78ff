"""TRACE_COOP2 -- the cooperative kernel with two rays per 8-lane group (csrc/art_kernels.hip k_trace_coop2) and its
overflow path (k_trace_overflow).  Same bar as every other kernel: hits and radiance bit-identical to the CPU oracle."""
import numpy as np
import pytest

import conv
import orc
from test_gpu_parity import _assert_hits_equal, _random_rays, assert_radiance_equal, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def width8(backend):
    """k_trace_coop2 is written for 8-lane groups: these tests build width-8 trees"""
    backend.set_option("bvh_width", 8)
    yield
    backend.set_option("bvh_width", 4)


@pytest.fixture()
def coop2(art, backend):
    backend.set_option("trace_kernel", art.TRACE_COOP2)
    yield backend
    backend.set_option("trace_kernel", art.TRACE_COOP)
    backend.set_option("stack_cap", 0)


@pytest.mark.parametrize("ntris", [1, 7, 300, 20000])
def test_coop2_hits_match_brute_force(art, backend, ntris):
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(ntris, 3)
    osc = conv.OracleScene(sd)
    backend.upload_scene(sd)
    o, d = _random_rays(30000, ntris)
    _assert_hits_equal(backend.trace_rays(o, d, kernel=art.TRACE_COOP2), orc.closest_hits(osc.scene, o, d))


@pytest.mark.parametrize("cap", [1, 3, 8])
def test_coop2_overflow_path(art, coop2, cap):
    """A tiny LDS stack cap pushes most rays through the overflow queue and k_trace_overflow: same hits, same image."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(20000, 3)
    osc = conv.OracleScene(sd)
    coop2.set_option("stack_cap", cap)
    coop2.upload_scene(sd)
    o, d = _random_rays(30000, 77)
    _assert_hits_equal(coop2.trace_rays(o, d, kernel=art.TRACE_COOP2), orc.closest_hits(osc.scene, o, d))
    sd = scenes.synthetic_scene(2000, 3)
    osc = conv.OracleScene(sd)
    coop2.upload_scene(sd)
    coop2.resize(64, 64)
    accum, _, spp = coop2.render_pass(art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=3), 0)
    ref, _, cnt = orc.render(osc.scene, orc.make_params(64, 64, orc.PT_MIS, True, 8, 1, seed=3))
    assert_radiance_equal(accum, ref, spp)
    assert coop2.stats().rays == cnt.rays


def test_coop2_counters_match_oracle_walk(art, backend):
    from ada_ray_tracer_amd import scenes
    mesh = scenes.random_triangles(20000, 77)
    lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    sd = art.SceneDesc([], lights, scenes.cornell_materials(), [mesh], None, scenes.REFERENCE_CAMERA)
    backend.upload_scene(sd)
    nodes, tris, info = backend.export_bvh()
    o, d = _random_rays(40000, 8)
    t, prim, cnt = orc.bvh_walk(nodes, tris, o, d, width=info.node_width)
    hits, st = backend.trace_rays(o, d, kernel=art.TRACE_COOP2, want_stats=True)
    gprim = np.array([h.prim_index if h.is_hit else -1 for h in hits], np.int32)
    assert np.array_equal(gprim, prim)
    assert (st.box_tests, st.tri_tests, st.node_visits, st.leaf_visits, st.traced_rays) == \
           (cnt.box_tests, cnt.tri_tests, cnt.node_visits, cnt.leaf_visits, cnt.rays)


@pytest.mark.parametrize("rt", ["PT_MIS", "PT_SHADOW", "PT_STUPID"])
def test_coop2_render_bit_exact(art, coop2, rt):
    from ada_ray_tracer_amd import scenes
    for sd, seed in ((scenes.synthetic_scene(2000, 3), 3), (scenes.mixed_scene(1500, 5), 6)):
        osc = conv.OracleScene(sd)
        coop2.upload_scene(sd)
        coop2.resize(64, 64)
        accum, _, spp = coop2.render_pass(art.Backend.pass_params(getattr(art, rt), True, 8, 1, seed=seed), 0)
        ref, _, cnt = orc.render(osc.scene, orc.make_params(64, 64, getattr(orc, rt), True, 8, 1, seed=seed))
        assert_radiance_equal(accum, ref, spp)
        assert coop2.stats().rays == cnt.rays


def test_coop2_cornell_reference_scene(art, coop2):
    cs = orc.CornellScene()
    coop2.upload_scene(conv.desc_from_oracle(art, cs))
    coop2.resize(96, 80)
    accum, _, spp = coop2.render_pass(art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=11), 0)
    ref, rspp, cnt = orc.render(cs.scene, orc.make_params(96, 80, orc.PT_MIS, True, 8, 2, seed=11))
    assert_radiance_equal(accum, ref, spp)


def test_coop2_full_size_scene_same_image_as_coop(art, backend):
    """C4 (1M triangles) at a reduced frame: both cooperative kernels produce the same bits and count the same rays."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(1000000, 4)
    backend.upload_scene(sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 4, seed=5)
    out = []
    try:
        for k in (art.TRACE_COOP, art.TRACE_COOP2):
            backend.set_option("trace_kernel", k)
            backend.resize(320, 180)
            accum, _, spp = backend.render_pass(p, 0)
            out.append((accum.copy(), backend.stats().rays))
    finally:
        backend.set_option("trace_kernel", art.TRACE_COOP)
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and out[0][1] == out[1][1]

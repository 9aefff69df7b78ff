"""Instanced scenes (round 5; embree_connect.cpp:147-184): the product's two-level search (a tree over the instances, one tree per mesh walked
with the ray in object space, triangles tested in world space) must give the picture of the explicitly FLATTENED scene, bit for bit.
CPU side: the product's device functions compiled for the host (tests/host_sim) against the oracle's render of the flattened scene; the
-m gpu twin (tests/test_gpu_instanced.py) repeats it through the C ABI."""
import numpy as np
import pytest

import conv
import hostsim
import orc


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("rt", ["PT_MIS", "PT_SHADOW", "PT_STUPID"])
def test_instanced_scene_equals_the_oracle_on_the_flattened_scene(art, rt):
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(12, 300)                       # 12 instances of two ~300-triangle meshes: rotated, non-uniformly scaled, translated
    flat = hostsim.flattened_copy(art, sd)
    assert flat.desc.n_instances == 0 and flat.desc.meshes[0].ntris == sum(sd.desc.meshes[sd.desc.instances[i].mesh].ntris for i in range(12))
    p = art.Backend.pass_params(getattr(art, rt), True, 6, 2, seed=21)
    acc, rays = hostsim.render(art, sd, p, 56, 44)             # the two-level search
    osc = conv.OracleScene(flat)
    ref, _, cnt = orc.render(osc.scene, orc.make_params(56, 44, getattr(orc, rt), True, 6, 2, seed=21))      # the oracle's O(N) scan of the flattened mesh
    assert rays == cnt.rays
    assert np.array_equal(bits(acc), bits(ref))
    acc2, rays2 = hostsim.render(art, flat, p, 56, 44)        # and the product's own render of the flattened mesh
    assert rays2 == rays and np.array_equal(bits(acc2), bits(ref))
    assert float(np.abs(acc).sum()) > 0.0


@pytest.mark.parametrize("inst_open", [0, 1, 8, 1000])
def test_mirrored_sheared_coincident_tiny_and_huge_instances(art, inst_open):
    """hostsim.awkward_instances: a mirror image, one transform twice, a shear, scales 1e-3 and 2.2, interpenetrating instances; glass, mirror
    and Phong triangles on the torus -- still the flattened scene's picture, bit for bit.  inst_open: the instance tree ends at whole
    instances (1; 0, the default, chooses: 1 here), at about 8 subtrees per instance, at every LEAF of the meshes' trees (1000 per instance asked for)."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(0, 260, transforms=hostsim.awkward_instances(), all_materials=True)
    flat = hostsim.flattened_copy(art, sd)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=77)
    hostsim.set_bvh_param(art, "inst_open", inst_open)
    try:
        acc, rays = hostsim.render(art, sd, p, 64, 48)
    finally:
        hostsim.set_bvh_param(art, "inst_open", 0)
    ref, _, cnt = orc.render(conv.OracleScene(flat).scene, orc.make_params(64, 48, orc.PT_MIS, True, 8, 2, seed=77))
    assert rays == cnt.rays
    assert np.array_equal(bits(acc), bits(ref))
    assert np.isfinite(acc).mean() > 0.99 and (acc > 0).mean() > 0.3      # (the reference's 1 / max(cos, 1e-20) gives a few pixels of 1e16 .. 1e38 on the mirror and glass triangles: the same bits in both)


@pytest.mark.parametrize("view", [(2.0, 1.0), (1.5, 0.1)])
def test_a_speck_far_from_the_origin(art, view):
    """a tiny instance (scale 5e-4) far from the origin seen from two speck sizes away.  Its ray in object space carries a binary32
    error of |minv| * |o| * 2^-24 = 6e-4 object units, more than the 1.1e-4 the meshes' boxes were padded by until round 5: the box test
    culled triangles the world-space triangle test accepts (these two views: 2 and 1 pixels of 9216 differed from the flattened scene).  The
    pad now follows the instances (art_instanced_build.cpp): the flattened scene's picture, bit for bit, grazing rays included."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.speck_scene(view=view)
    flat = hostsim.flattened_copy(art, sd)
    assert tuple(flat.desc.cam_pos) == tuple(sd.desc.cam_pos)
    p = art.Backend.pass_params(art.PT_MIS, True, 4, 1, seed=3)
    acc, rays = hostsim.render(art, sd, p, 96, 96)
    ref, _, cnt = orc.render(conv.OracleScene(flat).scene, orc.make_params(96, 96, orc.PT_MIS, True, 4, 1, seed=3))      # the oracle's O(N) scan of the flattened mesh
    assert rays == cnt.rays and np.array_equal(bits(acc), bits(ref))
    assert (acc > 0).mean() > 0.5


def test_four_thousand_small_instances(art):
    """4096 instances of two ~300-triangle meshes crowding the box (every ray inside dozens of instance boxes: the build opens them into
    entry points by its own rule): the two-level search == the product's own search of the flattened 1.2 M-triangle mesh, bits and ray count"""
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(4096, 300)
    flat = hostsim.flattened_copy(art, sd)
    assert flat.desc.meshes[0].ntris > 1200000
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 1, seed=9)
    acc, rays = hostsim.render(art, sd, p, 32, 24)
    acc2, rays2 = hostsim.render(art, flat, p, 32, 24)
    assert rays == rays2 and np.array_equal(bits(acc), bits(acc2)) and (acc > 0).mean() > 0.05


def test_instanced_scene_rejects_what_it_cannot_hold(art):
    from ada_ray_tracer_amd import scenes
    sd = scenes.instanced_scene(4, 100)
    sd.instances[1].m = (np.ctypeslib.ctypes.c_float * 12)(*([0.0] * 12))                                   # a singular matrix
    acc = np.zeros((8, 8, 3), np.float32)
    import ctypes as C
    rays = C.c_uint64(0)
    p = art.Backend.pass_params(art.PT_MIS, True, 2, 1, seed=1)
    rc = hostsim.lib(art).hs_render(C.byref(sd.desc), C.byref(p), 8, 8, 0, acc.ctypes.data_as(art.f32p), C.byref(rays))
    assert rc != 0 and b"singular" in hostsim.lib(art).hs_last_error()

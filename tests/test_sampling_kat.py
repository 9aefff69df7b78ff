"""Known-answer tests of light sampling and the five BSDFs against an INDEPENDENT numpy-float32 transcription of the Ada text
(tests/ada_transcription.py: lights.adb:42-255, materials.adb:18-410, vector_math.adb:64-82,175-326).  Both the oracle
(oracle/art_oracle.c, orc_kat_*) and the product's device code (csrc/art_shade.h compiled for the host by tests/host_sim, hs_kat_*)
must reproduce the transcription BIT FOR BIT on seeded random inputs -- a misreading of the Ada source shared by oracle and product
would have to be made a third time, in a third language, to pass."""
import ctypes as C

import numpy as np
import pytest

import ada_transcription as ada
import hostsim
import orc

f = np.float32
SEED = 0x5EED


def bits(x):
    return np.asarray(x, np.float32).view(np.uint32)


def unit(rng):
    v = rng.normal(size=3)
    return tuple(f(c) for c in ada.normalize(tuple(f(c) for c in v)))


def c3(v):
    return (C.c_float * 3)(*[float(c) for c in v])


@pytest.fixture(scope="module")
def libs(art):
    L = orc.lib()
    L.orc_kat_light_sample.argtypes = [C.POINTER(orc.Light), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, orc.f32p, orc.f32p]
    L.orc_kat_light_eval_pdf.argtypes = [C.POINTER(orc.Light), orc.f32p, orc.f32p, C.c_float]; L.orc_kat_light_eval_pdf.restype = C.c_float
    L.orc_kat_mat_sample.argtypes = [C.POINTER(orc.Material), C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, orc.f32p, orc.f32p, orc.f32p]
    L.orc_kat_mat_eval.argtypes = [C.POINTER(orc.Material), orc.f32p, orc.f32p, orc.f32p, orc.f32p]
    H = hostsim.lib(art)
    H.hs_kat_light_sample.argtypes = [C.POINTER(art.ArtLight), C.c_float, C.c_float, orc.f32p, orc.f32p]
    H.hs_kat_light_eval_pdf.argtypes = [C.POINTER(art.ArtLight), orc.f32p, orc.f32p, C.c_float]; H.hs_kat_light_eval_pdf.restype = C.c_float
    H.hs_kat_mat_sample.argtypes = [C.POINTER(art.ArtMaterial), C.c_float, C.c_float, orc.f32p, orc.f32p, orc.f32p]
    H.hs_kat_mat_eval.argtypes = [C.POINTER(art.ArtMaterial), orc.f32p, orc.f32p, orc.f32p, orc.f32p]
    return L, H


def uniforms(L, k, bounce=0):
    """the four uniforms the render draws at (seed, pixel k, sample 0, bounce): light x2, BSDF x2"""
    return [f(L.orc_rng_uniform(SEED, k, 0, bounce, s)) for s in range(4)]


def both_lights(art, l):
    a, b = orc.Light(), art.ArtLight()
    for x in (a, b):
        x.shape = l["shape"]; x.mat = 4
        for key in ("boxMin", "boxMax", "normal", "center", "intensity"):
            setattr(x, key, c3(l[key]))
        x.radius = float(l["radius"]); x.surfaceArea = float(l["surfaceArea"])
    return a, b


def both_mats(art, mtype, p):
    a, b = orc.Material(), art.ArtMaterial()
    for x in (a, b):
        x.type = mtype; x.light = 0
        x.p = (C.c_float * 8)(*([float(v) for v in p] + [0.0] * (8 - len(p))))
    return a, b


SPHERE_LIGHT = dict(shape=1, boxMin=ada.V(0, 0, 0), boxMax=ada.V(0, 0, 0), normal=ada.V(0, -1, 0), center=ada.V(0.0, 4.5, 1.0), radius=f(0.5),
                    intensity=ada.V(10, 10, 10), surfaceArea=f(f(4.0) * f(np.pi) * f(0.5) * f(0.5)))                       # scene.adb:104-122
RECT_LIGHT = dict(shape=0, boxMin=ada.V(-0.75, 4.98, 1.25), boxMax=ada.V(0.75, 4.98, 3.25), normal=ada.V(0, -1, 0), center=ada.V(0, 0, 0), radius=f(0),
                  intensity=ada.V(20, 20, 20), surfaceArea=f(f(1.5) * f(2.0)))


@pytest.mark.parametrize("light", [SPHERE_LIGHT, RECT_LIGHT], ids=["SphereLight", "AreaLight"])
def test_light_sample_and_eval_pdf(art, libs, light):
    L, H = libs
    lo, lp = both_lights(art, light)
    rng = np.random.default_rng(1)
    n_inside = 0
    for k in range(400):
        p = tuple(f(v) for v in (rng.random(3) * [5.0, 4.9, 5.0] + [-2.5, 0.0, 0.0]))
        if k % 25 == 0 and light is SPHERE_LIGHT:                      # points on / inside the light sphere: the uniform-sphere branch
            d = unit(rng)
            p = ada.add(light["center"], ada.scale(f(0.5 * (k % 50 == 0) + 0.25), d))
            n_inside += 1
        u = uniforms(L, k)
        want = ada.sphere_light_sample(light, u[0], u[1], p) if light is SPHERE_LIGHT else ada.area_light_sample(light, u[0], u[1], p)
        want10 = np.array(list(want["pos"]) + list(want["dir"]) + list(want["intensity"]) + [want["pdf"]], np.float32)
        o10 = np.zeros(10, np.float32); h10 = np.zeros(10, np.float32)
        L.orc_kat_light_sample(C.byref(lo), SEED, k, 0, 0, orc.fp(np.array(p, np.float32)), orc.fp(o10))
        H.hs_kat_light_sample(C.byref(lp), float(u[0]), float(u[1]), orc.fp(np.array(p, np.float32)), orc.fp(h10))
        assert np.array_equal(bits(o10), bits(want10)), ("oracle", k, o10, want10)
        assert np.array_equal(bits(h10), bits(want10)), ("product", k, h10, want10)
        rd = unit(rng); dist = f(rng.random() * 6 + 0.1)
        wpdf = ada.sphere_light_eval_pdf(light, p) if light is SPHERE_LIGHT else ada.area_light_eval_pdf(light, p, rd, dist)
        pa, ra = np.array(p, np.float32), np.array(rd, np.float32)
        assert bits(L.orc_kat_light_eval_pdf(C.byref(lo), orc.fp(pa), orc.fp(ra), float(dist))) == bits(wpdf)
        assert bits(H.hs_kat_light_eval_pdf(C.byref(lp), orc.fp(pa), orc.fp(ra), float(dist))) == bits(wpdf)
    assert light is not SPHERE_LIGHT or n_inside >= 10


MATERIALS = {   # scene.adb:155-180
    "Lambert": (orc.MAT_LAMBERT, [0.25, 0.5, 0.0]),
    "Mirror": (orc.MAT_MIRROR, [0.75, 0.75, 0.75]),
    "FresnelDielectric": (orc.MAT_GLASS, [0.75, 0.75, 0.75, 0.85, 0.85, 0.85, 1.75]),
    "Phong": (orc.MAT_PHONG, [0.75, 0.75, 0.75, 80.0]),
    "PhongWide": (orc.MAT_PHONG, [0.5, 0.6, 0.7, 3.0]),
}


@pytest.mark.parametrize("name", list(MATERIALS))
def test_bsdf_sample_and_eval(art, libs, name):
    L, H = libs
    mtype, p = MATERIALS[name]
    mo, mp = both_mats(art, mtype, p)
    pf = [f(v) for v in p]
    rng = np.random.default_rng(2)
    n_spec_branches = set()
    for k in range(500):
        n = unit(rng); d = unit(rng)
        if k % 3 and ada.dot(d, n) > 0:                                 # mostly rays arriving from outside; every third left as drawn (inside / grazing)
            d = ada.scale(f(-1), d)
        if k % 17 == 0:                                                 # grazing incidence
            t = ada.normalize(ada.cross(n, unit(rng)))
            d = ada.normalize(ada.add(t, ada.scale(f(-1e-3 * (k % 5)), n)))
        u = uniforms(L, k)
        if name == "Lambert":
            want = ada.lambert_sample(tuple(pf[:3]), u[2], u[3], d, n)
        elif name == "Mirror":
            want = ada.mirror_sample(tuple(pf[:3]), d, n)
        elif name == "FresnelDielectric":
            want = ada.glass_sample(tuple(pf[:3]), tuple(pf[3:6]), pf[6], u[2], d, n)
            n_spec_branches.add((bool(ada.dot(d, n) < 0), bool(ada.dot(want["dir"], n) * ada.dot(d, n) > 0)))
        else:
            want = ada.phong_sample(tuple(pf[:3]), pf[3], u[2], u[3], d, n)
        want8 = np.array(list(want["color"]) + list(want["dir"]) + [want["pdf"], 1.0 if want["specular"] else 0.0], np.float32)
        o8 = np.zeros(8, np.float32); h8 = np.zeros(8, np.float32)
        da, na = np.array(d, np.float32), np.array(n, np.float32)
        L.orc_kat_mat_sample(C.byref(mo), SEED, k, 0, 0, orc.fp(da), orc.fp(na), orc.fp(o8))
        H.hs_kat_mat_sample(C.byref(mp), float(u[2]), float(u[3]), orc.fp(da), orc.fp(na), orc.fp(h8))
        assert np.array_equal(bits(o8), bits(want8)), ("oracle", name, k, o8, want8)
        assert np.array_equal(bits(h8), bits(want8)), ("product", name, k, h8, want8)
        # EvalBxDF / EvalPDF for a light direction l and view direction v = -d
        l = unit(rng); v = ada.scale(f(-1), d)
        if name == "Lambert":
            wb, wp = ada.lambert_eval(tuple(pf[:3]), l, v, n)
        elif name.startswith("Phong"):
            wb, wp = ada.phong_eval(tuple(pf[:3]), pf[3], l, v, n)
        else:
            wb, wp = ada.V(0, 0, 0), f(1)                               # materials.adb:256-264, 333-341: no direct sampling of specular materials
        want4 = np.array(list(wb) + [wp], np.float32)
        o4 = np.zeros(4, np.float32); h4 = np.zeros(4, np.float32)
        la, va = np.array(l, np.float32), np.array(v, np.float32)
        L.orc_kat_mat_eval(C.byref(mo), orc.fp(la), orc.fp(va), orc.fp(na), orc.fp(o4))
        H.hs_kat_mat_eval(C.byref(mp), orc.fp(la), orc.fp(va), orc.fp(na), orc.fp(h4))
        assert np.array_equal(bits(o4), bits(want4)), ("oracle eval", name, k, o4, want4)
        assert np.array_equal(bits(h4), bits(want4)), ("product eval", name, k, h4, want4)
    if name == "FresnelDielectric":
        assert len(n_spec_branches) == 4, "reflection and refraction, entering and leaving, must all occur: %s" % n_spec_branches


# ------------------------------------------------------------------------------------------------ whole paths
def _transcribed_scene(rect_light=False):
    """the reference's internal scene (scene.adb:89-217) as the transcription's dict; the transformed pyramid comes from the VSGF loader,
    which tests/test_oracle_kat.py pins separately"""
    cs = orc.CornellScene(use_rect_light=rect_light)
    m = cs.mesh_arrays()
    V = ada.V
    mats = [dict(type="glass", refl=V(0.75, 0.75, 0.75), trans=V(0.85, 0.85, 0.85), ior=f(1.75)),
            dict(type="lambert", kd=V(0.5, 0.5, 0.5)), dict(type="lambert", kd=V(0.25, 0.5, 0.0)), dict(type="lambert", kd=V(0.5, 0.0, 0.0)),
            dict(type="light"), dict(type="mirror", refl=V(0.75, 0.75, 0.75)), None, None,
            dict(type="phong", refl=V(0.75, 0.75, 0.75), pw=f(80.0)), dict(type="lambert", kd=V(0.5, 0.5, 0.5)), dict(type="lambert", kd=V(0.5, 0.5, 0.5))]
    spheres = [(V(-1.5, 1.0, 1.5), f(1.0), 8), (V(1.4, 1.0, 3.0), f(1.0), 0)]
    if rect_light:
        lo, hi = V(-0.75, 4.98, 1.25), V(0.75, 4.98, 3.25)                         # scene.adb:104-107
        light = dict(shape=0, boxMin=lo, boxMax=hi, normal=V(0, -1, 0), center=V(0, 0, 0), radius=f(0), intensity=V(20, 20, 20),
                     surfaceArea=(hi[0] - lo[0]) * (hi[2] - lo[2]))
    else:
        light = dict(SPHERE_LIGHT)
        spheres.append((V(0.0, 4.5, 1.0), f(0.5), 4))
    scn = dict(spheres=spheres, light=light, materials=mats,
               cornell=dict(min=V(-2.5, 0, 0), max=V(2.5, 5, 5), mat=(2, 3, 1, 1, 8, 1),
                            nrm=(V(1, 0, 0), V(-1, 0, 0), V(0, 1, 0), V(0, -1, 0), V(0, 0, 1), V(0, 0, -1))),
               mesh=dict(pos=[tuple(f(c) for c in p) for p in m["pos"]], nrm=[tuple(f(c) for c in p) for p in m["nrm"]],
                         idx=[tuple(int(i) for i in t) for t in m["idx"]], bbmin=tuple(f(c) for c in m["bbmin"]), bbmax=tuple(f(c) for c in m["bbmax"])))
    return cs, scn


@pytest.mark.parametrize("kind,rt", [("mis", orc.PT_MIS), ("shadow", orc.PT_SHADOW), ("stupid", orc.PT_STUPID)])
def test_whole_paths_against_the_transcribed_integrators(kind, rt):
    """PathTrace x 3 (ray_tracer-integrators.adb:82-301), Find_Closest_Hit (scene.adb:56-86), the five intersectors (geometry.adb:48-323),
    Compute_Shadow and the camera rays (ray_tracer.adb:61-132), transcribed a third time in numpy float32 (tests/ada_transcription.py):
    the oracle's radiance of single camera samples must equal the transcription's BIT FOR BIT -- glass, Phong, Lambert, the light, the
    pyramid's first-hit-wins mesh scan, shadow rays, the MIS weights and the recursion's association, paths up to depth 8."""
    cs, scn = _transcribed_scene()
    L = orc.lib()
    W = H = 24; depth = 8; seed = 31
    prm = orc.make_params(W, H, rt, True, depth, 1, seed=seed)
    rng = np.random.default_rng(5)
    # pixels spread over the picture + pixels aimed at the glass sphere, the Phong sphere, the pyramid and the light
    pixels = [(int(x), int(y)) for x, y in zip(rng.integers(4, 20, 14), rng.integers(4, 20, 14))] + [(15, 8), (16, 9), (8, 8), (9, 9), (10, 7), (11, 7), (12, 16), (12, 15)]
    third = (f(1.0 / 3.0), f(2.0 / 3.0))
    n_nonzero = 0; deepest = 0
    for (x, y) in pixels:
        for sample in range(4):
            ox, oy = third[(sample >> 1) & 1], third[sample & 1]                 # Generate4RayDirections order
            d0 = ada.eye_ray_direction(x, y, ox, oy, W, H)
            cm = cs.scene.cam_matrix
            d = ada.normalize(tuple(f(cm[4 * r]) * d0[0] + f(cm[4 * r + 1]) * d0[1] + f(cm[4 * r + 2]) * d0[2] + f(cm[4 * r + 3]) for r in range(3)))
            o = tuple(f(v) for v in cs.scene.cam_pos)
            pixel = y * W + x
            seen = []

            def uniforms(bounce, pixel=pixel, sample=sample, seen=seen):
                seen.append(bounce)
                return [f(L.orc_rng_uniform(seed, pixel, sample, bounce, s)) for s in range(4)]
            want = ada.path_trace(scn, kind, o, d, dict(pdf=f(1), specular=True), depth, uniforms, depth)
            got = np.zeros(3, np.float32)
            L.orc_sample_radiance(C.byref(cs.scene), C.byref(prm), x, y, sample, orc.fp(got))
            assert np.array_equal(bits(got), bits(np.array(want, np.float32))), (kind, x, y, sample, got, want)
            n_nonzero += bool(np.any(got != 0)); deepest = max(deepest, max(seen) + 1 if seen else 0)
    assert n_nonzero > (20 if kind != "stupid" else 2) and deepest == depth


def test_whole_paths_with_the_rect_light_overflow():
    """the reference's own overflow (AreaLight pdf = d^2 / (A * 1e-20) above the light plane -> MIS weight inf / inf = NaN,
    lights.adb:42-45, integrators.adb:277-279) comes out of the transcription as well: same NaNs, same bits elsewhere"""
    cs, scn = _transcribed_scene(rect_light=True)
    L = orc.lib()
    W = H = 16; depth = 4; seed = 3
    prm = orc.make_params(W, H, orc.PT_MIS, False, depth, 1, seed=seed)
    n_nan = 0
    for (x, y) in [(8, 12), (6, 11), (10, 12), (8, 11), (7, 12), (8, 3), (3, 8), (12, 8), (8, 8)]:       # rows 11-12 look at the ceiling
        d = ada.normalize(ada.eye_ray_direction(x, y, f(0.5), f(0.5), W, H))
        o = tuple(f(v) for v in cs.scene.cam_pos)
        uniforms = lambda bounce, pixel=y * W + x: [f(L.orc_rng_uniform(seed, pixel, 0, bounce, s)) for s in range(4)]
        want = np.array(ada.path_trace(scn, "mis", o, d, dict(pdf=f(1), specular=True), depth, uniforms, depth), np.float32)
        got = np.zeros(3, np.float32)
        L.orc_sample_radiance(C.byref(cs.scene), C.byref(prm), x, y, 0, orc.fp(got))
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(bits(got)[~np.isnan(got)], bits(want)[~np.isnan(want)]), (x, y, got, want)
        n_nan += bool(np.isnan(got).any())
    assert n_nan >= 1


def test_pyramid_transform_chain_against_the_transcription(art):
    """scene.adb:194-206 + geometry.adb:593-607: RotationMatrix(-PI/6), mtans * mrot * mscale, positions transformed, bounding box -- the
    transcription, the oracle's loader and the product's host layer (host/art_host.cpp) give the same bits"""
    import json
    from ada_ray_tracer_amd import scenes
    T = ada.cornell_mesh_transform()
    To = np.zeros(16, np.float32); orc.lib().orc_cornell_mesh_transform(orc.fp(To))
    assert np.array_equal(bits(To), bits(np.array(T, np.float32).ravel()))
    g = json.load(open(orc.GOLDEN + "/vsgf_decode.json"))
    want = np.array([ada.mat_mul_point(T, tuple(f(c) for c in p)) for p in g["positions"]], np.float32)
    m = orc.CornellScene().mesh_arrays()
    assert np.array_equal(bits(m["pos"]), bits(want))
    assert np.array_equal(bits(m["bbmin"]), bits(want.min(0))) and np.array_equal(bits(m["bbmax"]), bits(want.max(0)))
    d = scenes.reference_scene().desc
    assert np.array_equal(bits(np.ctypeslib.as_array(d.meshes[0].pos, (17, 3))), bits(want))

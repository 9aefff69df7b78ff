"""Scene inputs for the backend: the material / light tables of the reference's internal scene
(scene.adb:100-217) and the synthetic N-triangle generator of SURVEY 8(d) used by the benchmark configs
C3 (100k triangles), C4 (1M) and C5 (mixed).  Pure numpy; everything is deterministic (SplitMix64)."""
import numpy as np

from . import (LIGHT_RECT, LIGHT_SPHERE, MAT_GLASS, MAT_LAMBERT, MAT_LIGHT, MAT_MIRROR, MAT_NULL, MAT_PHONG,
               MESH_CLOSEST, SceneDesc)      # noqa: F401

F = np.float32
PYRAMID_VSGF = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "data", "pyramid2.vsgf")

CORNELL_BOX = dict(min=(-2.5, 0.0, 0.0), max=(2.5, 5.0, 5.0), mat=(2, 3, 1, 1, 8, 1),            # scene.ads:75-80
                   nrm=((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)))
REFERENCE_CAMERA = (0.0, 2.55, 12.5)                                                               # scene.adb:212


def cornell_materials():
    """materials(0..10) of Init_Cornell_Box (scene.adb:155-180); 6 and 7 stay null."""
    white = dict(type=MAT_LAMBERT, p=(0.5, 0.5, 0.5))
    return [dict(type=MAT_GLASS, p=(0.75, 0.75, 0.75, 0.85, 0.85, 0.85, 1.75)),
            white,
            dict(type=MAT_LAMBERT, p=(0.25, 0.5, 0.0)),
            dict(type=MAT_LAMBERT, p=(0.5, 0.0, 0.0)),
            dict(type=MAT_LIGHT, light=0),
            dict(type=MAT_MIRROR, p=(0.75, 0.75, 0.75)),
            dict(type=MAT_NULL), dict(type=MAT_NULL),
            dict(type=MAT_PHONG, p=(0.75, 0.75, 0.75, 80.0)),
            white, white]


def splitmix64(seed, n):
    """n 64-bit outputs of SplitMix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    return ((splitmix64(seed, n) >> np.uint64(40)).astype(np.float64) * 2.0 ** -24).astype(F)


def random_triangles(n_tris, seed):
    """De-indexed triangle soup inside the Cornell box: centre ~U([-2.3,2.3]x[0.2,4.6]x[0.2,4.8]), two edge vectors
    ~U([-s,s]^3), s = 2.5 N^(-1/3); per-vertex normals = flat normal of the front face (cross(B-A, C-A))."""
    u = uniform01(seed, 9 * n_tris).reshape(n_tris, 9)
    lo = np.array([-2.3, 0.2, 0.2], F); hi = np.array([2.3, 4.6, 4.8], F)
    c = lo + u[:, 0:3] * (hi - lo)
    s = F(2.5 * n_tris ** (-1.0 / 3.0))
    e1 = (u[:, 3:6] * F(2.0) - F(1.0)) * s
    e2 = (u[:, 6:9] * F(2.0) - F(1.0)) * s
    A = c; B = (c + e1).astype(F); Cc = (c + e2).astype(F)
    n = np.cross(e1, e2).astype(F)
    ln = np.sqrt((n * n).sum(1, keepdims=True)).astype(F)
    n = np.where(ln > 0, n / np.maximum(ln, F(1e-30)), np.array([0, 1, 0], F)).astype(F)
    pos = np.stack([A, B, Cc], 1).reshape(-1, 3).astype(F)
    nrm = np.repeat(n, 3, axis=0).astype(F)
    idx = np.arange(3 * n_tris, dtype=np.int32).reshape(-1, 3)
    matid = (1 + (np.arange(n_tris) % 3)).astype(np.int32)          # white, green, red
    return dict(mode=MESH_CLOSEST, pos=pos, nrm=nrm, idx=idx, matid=matid)


def grid_mesh(m, y0=1.0, tilt=0.35):
    """Regular tessellation: m x m quads = 2 m^2 equal triangles on a tilted plane inside the Cornell box, front faces towards
    the camera.  With m a power of two the Morton-sorted radix tree is perfectly balanced -- the worst case for the node count of a
    wide tree with 1-triangle leaves (about 2n/3 nodes), which random soups never reach."""
    g = np.linspace(0.0, 1.0, m + 1, dtype=np.float64)
    X, Z = np.meshgrid(-2.0 + 4.0 * g, 0.5 + 4.0 * g, indexing="ij")
    Y = y0 + tilt * (Z - 0.5)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(F)
    i, j = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    a = (i * (m + 1) + j).ravel(); b = a + 1; c = a + (m + 1); d = c + 1
    idx = np.stack([np.stack([a, b, c], 1), np.stack([b, d, c], 1)], 1).reshape(-1, 3).astype(np.int32)   # normal = +y-ish
    e1 = pos[idx[:, 1]] - pos[idx[:, 0]]; e2 = pos[idx[:, 2]] - pos[idx[:, 0]]
    fn = np.cross(e1, e2); fn /= np.linalg.norm(fn, axis=1, keepdims=True)
    nrm = np.zeros_like(pos); nrm[idx[:, 0]] = fn; nrm[idx[:, 1]] = fn; nrm[idx[:, 2]] = fn
    matid = (1 + (np.arange(idx.shape[0]) % 3)).astype(np.int32)
    return dict(mode=MESH_CLOSEST, pos=pos, nrm=nrm.astype(F), idx=idx, matid=matid)


def torus_mesh(nu=256, nv=96, R=1.3, r=0.45, centre=(0.0, 2.2, 2.4)):
    """A closed, connected surface (2 nu nv triangles of very different sizes: small on the inside of the ring, large outside), tilted in
    the Cornell box -- the structured counterpart of the random soups: shared vertices, coplanar neighbours, slivers at the seams."""
    u = np.linspace(0.0, 2.0 * np.pi, nu, endpoint=False); v = np.linspace(0.0, 2.0 * np.pi, nv, endpoint=False)
    U, V = np.meshgrid(u, v, indexing="ij")
    x = (R + r * np.cos(V)) * np.cos(U); y = r * np.sin(V); z = (R + r * np.cos(V)) * np.sin(U)
    c, s_ = np.cos(0.6), np.sin(0.6)
    pos = np.stack([x, c * y - s_ * z, s_ * y + c * z], -1).reshape(-1, 3) + np.asarray(centre)
    n = np.stack([np.cos(V) * np.cos(U), np.sin(V), np.cos(V) * np.sin(U)], -1).reshape(-1, 3)
    nrm = np.stack([n[:, 0], c * n[:, 1] - s_ * n[:, 2], s_ * n[:, 1] + c * n[:, 2]], -1)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a_ = (i * nv + j).ravel(); b_ = (((i + 1) % nu) * nv + j).ravel(); c_ = (i * nv + (j + 1) % nv).ravel(); d_ = (((i + 1) % nu) * nv + (j + 1) % nv).ravel()
    idx = np.stack([np.stack([a_, c_, b_], 1), np.stack([b_, c_, d_], 1)], 1).reshape(-1, 3).astype(np.int32)      # outward-facing
    matid = (1 + (np.arange(idx.shape[0]) % 3)).astype(np.int32)
    return dict(mode=MESH_CLOSEST, pos=pos.astype(F), nrm=nrm.astype(F), idx=idx, matid=matid)


def rect_light(cx, mat, intensity=(20.0, 20.0, 20.0), y=4.98, half_x=0.25, cz=2.25, half_z=0.33):
    bmin = (F(cx - half_x), F(y), F(cz - half_z)); bmax = (F(cx + half_x), F(y), F(cz + half_z))
    area = F(F(bmax[0] - bmin[0]) * F(bmax[2] - bmin[2]))
    return dict(shape=LIGHT_RECT, mat=mat, boxMin=[float(v) for v in bmin], boxMax=[float(v) for v in bmax],
                normal=(0.0, -1.0, 0.0), intensity=intensity, surfaceArea=float(area))


def sphere_light(cx, mat, cy=4.5, cz=2.25, radius=0.25, intensity=(10.0, 10.0, 10.0)):
    area = float(F(4.0) * F(np.pi) * F(radius) * F(radius))          # scene.adb:122
    return dict(shape=LIGHT_SPHERE, mat=mat, center=(cx, cy, cz), radius=radius, intensity=intensity, surfaceArea=area)


def synthetic_scene(n_tris, config_id=3, rect_lights=False):
    """C3 / C4: Cornell walls + n_tris random triangles + 3 area lights, light choice uniform per bounce (an extension:
    the reference has a single light, scene.adb:45-48).

    Default lights are three SphereLights (lights.adb:101-266), radius 0.25 at y = 4.5, x in {-1.5, 0, 1.5}.
    SURVEY 8(d) suggested rect AreaLights at y = 4.98; with PT_MIS those overflow in the reference's own arithmetic:
    for any surface point at or above the light plane (the whole ceiling) AreaLight.Sample returns
    pdf = d^2 / (A * 1e-20) (lights.adb:42-45,72-74), its square is +inf in binary32 and the MIS weight
    inf/(inf+x) is NaN (integrators.adb:277-279) -- the accumulated pixel never recovers.  The backend reproduces that
    bit-for-bit (tests/test_gpu_parity.py::test_rect_light_mis_nan_pattern_matches), but a benchmark image made of NaNs
    is useless, so the rect variant is kept for parity tests only (rect_lights=True)."""
    mats = cornell_materials()
    mats[4] = dict(type=MAT_LIGHT, light=0)
    mats.append(dict(type=MAT_LIGHT, light=1))     # 11
    mats.append(dict(type=MAT_LIGHT, light=2))     # 12
    mesh = random_triangles(n_tris, 0xADA5EED0 + config_id)
    if rect_lights:
        lights = [rect_light(-1.5, 4), rect_light(0.0, 11), rect_light(1.5, 12)]
        spheres = []
    else:
        lights = [sphere_light(-1.5, 4), sphere_light(0.0, 11), sphere_light(1.5, 12)]
        spheres = [(l["center"], l["radius"], l["mat"]) for l in lights]
    return SceneDesc(spheres=spheres, lights=lights, materials=mats, meshes=[mesh], cornell=CORNELL_BOX, cam_pos=REFERENCE_CAMERA)


def structured_scene(n_tris=1000000):
    """The structured counterpart of C4 (perf evidence beyond random soups, bench.py --scene s4): Cornell walls + the three sphere lights of
    synthetic_scene + two closed, connected surfaces of about n_tris triangles in total -- a finely tessellated torus (triangles of very
    different sizes, slivers at the seams, coplanar neighbours) over a tilted regular grid (equal triangles: the radix tree's balanced
    worst case).  Not a BASELINE configuration."""
    mats = cornell_materials()
    mats[4] = dict(type=MAT_LIGHT, light=0)
    mats.append(dict(type=MAT_LIGHT, light=1)); mats.append(dict(type=MAT_LIGHT, light=2))
    lights = [sphere_light(-1.5, 4), sphere_light(0.0, 11), sphere_light(1.5, 12)]
    spheres = [(l["center"], l["radius"], l["mat"]) for l in lights]
    nv = max(8, int(round((0.75 * n_tris / 5.0) ** 0.5))); nu = int(round(2.5 * nv))             # 2 nu nv = 5 nv^2 = 3/4 of the triangles ...
    m = max(4, int(round((n_tris / 4 / 2) ** 0.5)))                                              # ... 2 m^2 = the rest
    tor = torus_mesh(nu, nv); grd = grid_mesh(m, y0=0.4, tilt=0.15)
    pos = np.concatenate([tor["pos"], grd["pos"]]); nrm = np.concatenate([tor["nrm"], grd["nrm"]])
    idx = np.concatenate([tor["idx"], grd["idx"] + tor["pos"].shape[0]]).astype(np.int32)
    matid = (1 + (np.arange(idx.shape[0]) % 3)).astype(np.int32)
    mesh = dict(mode=MESH_CLOSEST, pos=pos, nrm=nrm, idx=idx, matid=matid)
    return SceneDesc(spheres=spheres, lights=lights, materials=mats, meshes=[mesh], cornell=CORNELL_BOX, cam_pos=REFERENCE_CAMERA)


def instanced_scene(n_instances=64, tris_per_mesh=20000, seed=0xADA5EED0 + 64, transforms=None, all_materials=False):
    """Round 5 (embree_connect.cpp:147-184): Cornell walls + the three sphere lights of synthetic_scene + n_instances
    instances of TWO prototype meshes of about tris_per_mesh triangles each (a torus and a bumpy grid, object space around the origin),
    every instance under its own rotation, non-uniform scale and translation.  Rendered through the two-level tree without flattening;
    tests/host_sim's hs_flatten_instances gives the explicit world-space mesh the picture must equal bit for bit.
    transforms: [(mesh index 0 | 1, 3x4 object -> world)] instead of the n_instances seeded ones (the tests' mirrored, sheared, coincident,
    tiny and huge instances); all_materials: the torus also carries mirror (5) and Phong (8) triangles next to its glass (0) ones."""
    mats = cornell_materials()
    mats[4] = dict(type=MAT_LIGHT, light=0)
    mats.append(dict(type=MAT_LIGHT, light=1)); mats.append(dict(type=MAT_LIGHT, light=2))
    lights = [sphere_light(-1.5, 4), sphere_light(0.0, 11), sphere_light(1.5, 12)]
    spheres = [(l["center"], l["radius"], l["mat"]) for l in lights]
    nv = max(4, int(round((tris_per_mesh / 5.0) ** 0.5))); nu = int(round(2.5 * nv))
    tor = torus_mesh(nu, nv, R=0.7, r=0.3, centre=(0.0, 0.0, 0.0))
    m = max(3, int(round((tris_per_mesh / 2) ** 0.5)))
    g = np.linspace(-1.0, 1.0, m + 1)
    X, Z = np.meshgrid(g, g, indexing="ij")
    Y = 0.15 * np.sin(3.0 * X) * np.cos(2.0 * Z)
    pos = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(F)
    dydx = 0.45 * np.cos(3.0 * X) * np.cos(2.0 * Z); dydz = -0.3 * np.sin(3.0 * X) * np.sin(2.0 * Z)
    nrm = np.stack([-dydx, np.ones_like(X), -dydz], -1).reshape(-1, 3); nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(F)
    i, j = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    a_ = (i * (m + 1) + j).ravel(); b_ = ((i + 1) * (m + 1) + j).ravel(); c_ = (i * (m + 1) + j + 1).ravel(); d_ = ((i + 1) * (m + 1) + j + 1).ravel()
    idx = np.stack([np.stack([a_, c_, b_], 1), np.stack([b_, c_, d_], 1)], 1).reshape(-1, 3).astype(np.int32)
    grid = dict(mode=MESH_CLOSEST, pos=pos, nrm=nrm, idx=idx, matid=(1 + (np.arange(idx.shape[0]) % 3)).astype(np.int32))
    tn = np.arange(tor["idx"].shape[0])
    tor["matid"] = np.where(tn % 7 == 0, 0, 1 + (tn % 3)).astype(np.int32)      # every seventh triangle glass (material 0)
    if all_materials:
        tor["matid"] = np.where(tn % 7 == 3, 5, np.where(tn % 7 == 5, 8, tor["matid"])).astype(np.int32)
    rng = np.random.default_rng(seed)
    insts = []
    for k in range(n_instances):
        ax, ay, az = rng.random(3) * 2.0 * np.pi
        Rx = np.array([[1, 0, 0], [0, np.cos(ax), -np.sin(ax)], [0, np.sin(ax), np.cos(ax)]])
        Ry = np.array([[np.cos(ay), 0, np.sin(ay)], [0, 1, 0], [-np.sin(ay), 0, np.cos(ay)]])
        Rz = np.array([[np.cos(az), -np.sin(az), 0], [np.sin(az), np.cos(az), 0], [0, 0, 1]])
        S = np.diag(0.25 + 0.35 * rng.random(3))
        M = np.zeros((3, 4)); M[:, :3] = Rz @ Ry @ Rx @ S
        M[:, 3] = [-2.0 + 4.0 * rng.random(), 0.5 + 3.6 * rng.random(), 0.3 + 4.0 * rng.random()]
        insts.append((k % 2, M.astype(F).ravel()))
    if transforms is not None:
        insts = [(int(mi), np.asarray(M, F).reshape(3, 4).ravel()) for mi, M in transforms]
    return SceneDesc(spheres=spheres, lights=lights, materials=mats, meshes=[tor, grid], cornell=CORNELL_BOX, cam_pos=REFERENCE_CAMERA, instances=insts)


def instanced_cluster(n_instances=64, tris_per_mesh=20000, shrink=0.25, grow=1.5):
    """instanced_scene's instances pulled together into one interpenetrating cluster (translations shrunk towards the centre of the box,
    scales x grow): the boxes of whole instances nearly coincide -- the case the build opens instances for (art_instanced_build.cpp)."""
    sd = instanced_scene(n_instances, tris_per_mesh)
    c = np.array([0.0, 2.3, 2.3])
    tr = []
    for i in range(n_instances):
        M = np.array(list(sd.instances[i].m), np.float64).reshape(3, 4)
        M[:, 3] = c + shrink * (M[:, 3] - c); M[:, :3] *= grow
        tr.append((int(sd.instances[i].mesh), M))
    return instanced_scene(n_instances, tris_per_mesh, transforms=tr)


def mixed_scene(n_tris=20000, config_id=5, extra_spheres=8):
    """C5: spheres (Phong, glass, extra glass/diffuse) + emissive sphere light + a triangle mesh inside the Cornell box."""
    mats = cornell_materials()
    area = float(F(4.0) * F(np.pi) * F(0.5) * F(0.5))
    lights = [dict(shape=LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=area)]
    spheres = [((-1.5, 1.0, 1.5), 1.0, 8), ((1.4, 1.0, 3.0), 1.0, 0), ((0.0, 4.5, 1.0), 0.5, 4)]
    u = uniform01(0xADA5EED0 + 100 + config_id, 3 * extra_spheres).reshape(-1, 3)
    for i in range(extra_spheres):
        p = (float(-2.0 + 4.0 * u[i, 0]), float(2.4 + 1.6 * u[i, 1]), float(0.6 + 3.6 * u[i, 2]))
        spheres.append((p, 0.4, (0, 1, 2, 3)[i % 4]))
    mesh = random_triangles(n_tris, 0xADA5EED0 + config_id)
    return SceneDesc(spheres=spheres, lights=lights, materials=mats, meshes=[mesh], cornell=CORNELL_BOX, cam_pos=REFERENCE_CAMERA)


def mirror_scene(n_tris=600, seed=0xADA5EED0 + 17):
    """Every material of materials.adb in one scene, MaterialMirror (scene.adb:168: materials(5), which no primitive of the reference's
    own scene carries) on a sphere AND on mesh triangles: a mirror sphere, a Phong sphere, a glass sphere, the sphere light, and a
    triangle soup whose material ids cycle white / green / red / mirror."""
    mats = cornell_materials()
    area = float(F(4.0) * F(np.pi) * F(0.5) * F(0.5))
    lights = [dict(shape=LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=area)]
    spheres = [((-1.3, 1.0, 1.8), 1.0, 5), ((1.4, 1.0, 3.0), 1.0, 0), ((0.1, 0.6, 3.9), 0.6, 8), ((0.0, 4.5, 1.0), 0.5, 4)]
    mesh = random_triangles(n_tris, seed)
    mesh["matid"] = np.array([(1, 2, 3, 5)[i % 4] for i in range(n_tris)], np.int32)
    return SceneDesc(spheres=spheres, lights=lights, materials=mats, meshes=[mesh], cornell=CORNELL_BOX, cam_pos=REFERENCE_CAMERA)


class HostSceneDesc:
    """ArtSceneDesc built by the product's own host layer (host/art_host.cpp, Scene.Init = scene.adb:24-27, 89-217): the reference's
    internal Cornell scene with data/pyramid2.vsgf.  `.desc` is what Backend.upload_scene takes."""

    def __init__(self, vsgf_path=None, cam_pos=None):
        import ctypes as C
        import os
        from . import HOST_LIB_PATH, ArtError, ArtSceneDesc
        lib = C.CDLL(HOST_LIB_PATH)
        lib.art_host_scene_init.restype = C.POINTER(ArtSceneDesc)
        lib.art_host_scene_init.argtypes = [C.c_char_p]
        p = lib.art_host_scene_init((vsgf_path or PYRAMID_VSGF).encode())
        if not p:
            raise ArtError("art_host_scene_init failed (is %s there?)" % (vsgf_path or PYRAMID_VSGF))
        self._lib = lib
        self.desc = ArtSceneDesc()
        C.memmove(C.byref(self.desc), p, C.sizeof(ArtSceneDesc))      # pointers stay those of the host layer's static scene
        if cam_pos is not None:
            self.desc.cam_pos = (C.c_float * 3)(*cam_pos)


class HydraSceneDesc:
    """A Hydra scene library (statex_00001.xml + its VSGF meshes; scene_hydra_embree.adb:303-390) as a render scene, built by the product's
    host layer (host/hydra_scene.cpp Build_Render_Desc): the library's meshes as prototypes, its <instance>s as ArtInstances, inside the internal
    scene's box / spheres / light / camera; materials = the internal table + one Lambert per <material>.  Rendered through the two-level tree."""

    def __init__(self, scene_dir, vsgf_path=None):
        import ctypes as C
        from . import HOST_LIB_PATH, ArtError, ArtSceneDesc
        lib = C.CDLL(HOST_LIB_PATH)
        lib.art_host_hydra_scene_create.restype = C.c_void_p
        lib.art_host_hydra_scene_create.argtypes = [C.c_char_p, C.c_char_p]
        lib.art_host_hydra_scene_desc.restype = C.POINTER(ArtSceneDesc)
        lib.art_host_hydra_scene_desc.argtypes = [C.c_void_p]
        lib.art_host_hydra_scene_clamped_ids.restype = C.c_longlong
        lib.art_host_hydra_scene_clamped_ids.argtypes = [C.c_void_p]
        lib.art_host_hydra_scene_destroy.argtypes = [C.c_void_p]
        self._lib = lib
        self._handle = lib.art_host_hydra_scene_create(str(scene_dir).encode(), (vsgf_path or PYRAMID_VSGF).encode())
        if not self._handle:
            raise ArtError("art_host_hydra_scene_create failed for %s" % scene_dir)
        self.clamped_material_ids = int(lib.art_host_hydra_scene_clamped_ids(self._handle))
        self.desc = ArtSceneDesc()
        C.memmove(C.byref(self.desc), lib.art_host_hydra_scene_desc(self._handle), C.sizeof(ArtSceneDesc))      # its pointers live as long as the handle this object owns

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            self._lib.art_host_hydra_scene_destroy(h)


def reference_scene(cam_pos=None):
    """C2 (and, with eight spheres, C1): Scene.Init through the product's host mirror."""
    return HostSceneDesc(cam_pos=cam_pos)


def eight_sphere_scene(seed=0xADA5EED0 + 1):
    """C1 'Cornell-box-style 8-sphere scene' (BASELINE.json configs[0]): the internal scene has 3 spheres (scene.adb:139-144,182-192:
    Phong, glass, the light); SURVEY 8(d) completes it to 8 with five spheres of radius 0.4 at seeded positions inside the box,
    materials cycling Lambert white / Phong / glass / Lambert green / Lambert red.  No mesh: plumbing only, as BASELINE says."""
    mats = cornell_materials()
    area = float(F(4.0) * F(np.pi) * F(0.5) * F(0.5))
    lights = [dict(shape=LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=area)]
    spheres = [((-1.5, 1.0, 1.5), 1.0, 8), ((1.4, 1.0, 3.0), 1.0, 0), ((0.0, 4.5, 1.0), 0.5, 4)]
    u = uniform01(seed, 15).reshape(5, 3)
    for i in range(5):
        p = (float(F(-2.0) + F(4.0) * u[i, 0]), float(F(2.3) + F(1.4) * u[i, 1]), float(F(0.6) + F(3.8) * u[i, 2]))
        spheres.append((p, 0.4, (1, 8, 0, 2, 3)[i]))
    return SceneDesc(spheres=spheres, lights=lights, materials=mats, meshes=[], cornell=CORNELL_BOX, cam_pos=REFERENCE_CAMERA)


def speck_scene(scale=5.0e-4, centre=(-2.1, 4.4, 0.7), view=(2.0, 1.0), tris=600):
    """ONE instance of instanced_scene's bumpy grid (object coordinates within +-1) shrunk to a speck of `scale` and placed far from the
    origin, the camera `view` = (dz, dy) speck sizes in front of / above it so that it fills the frame: the ray taken to object space in
    binary32 is then |minv| * |o| * 2^-24 = several 1e-4 object units off -- the case the meshes' absolute box pad has to follow
    (csrc/art_instanced_build.cpp; round 6).  Rendered instanced it must still be the flattened scene's picture, bit for bit."""
    m = np.zeros((3, 4)); m[:, :3] = np.eye(3) * scale; m[:, 3] = centre
    sd = instanced_scene(0, tris, transforms=[(1, m)])
    cam = (centre[0], centre[1] + view[1] * scale, centre[2] + view[0] * scale)
    for k in range(3):
        sd.desc.cam_pos[k] = cam[k]
    sd._kw["cam_pos"] = cam
    return sd

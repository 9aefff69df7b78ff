"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(_HERE, "golden")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)

MAT_NULL, MAT_LIGHT, MAT_LAMBERT, MAT_MIRROR, MAT_GLASS, MAT_PHONG = range(6)
LIGHT_RECT, LIGHT_SPHERE = 0, 1
MESH_REFERENCE_BF, MESH_CLOSEST = 0, 1
RT_DEBUG, RT_WHITTED, PT_STUPID, PT_SHADOW, PT_MIS = range(5)


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("light", C.c_int32), ("p", C.c_float * 8)]


class Light(C.Structure):
    _fields_ = [("shape", C.c_int32), ("mat", C.c_int32),
                ("boxMin", C.c_float * 3), ("boxMax", C.c_float * 3), ("normal", C.c_float * 3),
                ("center", C.c_float * 3), ("radius", C.c_float),
                ("intensity", C.c_float * 3), ("surfaceArea", C.c_float)]


class Sphere(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("r", C.c_float), ("mat", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("mode", C.c_int32), ("nverts", C.c_int32), ("ntris", C.c_int32),
                ("pos", f32p), ("nrm", f32p), ("uv", f32p), ("idx", i32p), ("matid", i32p),
                ("bbmin", C.c_float * 3), ("bbmax", C.c_float * 3),
                ("bvh_nodes", f32p), ("bvh_tris", f32p), ("bvh_width", C.c_int32)]


class Scene(C.Structure):
    _fields_ = [("n_spheres", C.c_int32), ("spheres", C.POINTER(Sphere)),
                ("has_cornell", C.c_int32),
                ("cb_min", C.c_float * 3), ("cb_max", C.c_float * 3),
                ("cb_mat", C.c_int32 * 6), ("cb_nrm", (C.c_float * 3) * 6),
                ("n_lights", C.c_int32), ("lights", C.POINTER(Light)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(Material)),
                ("n_meshes", C.c_int32), ("meshes", C.POINTER(Mesh)),
                ("cam_pos", C.c_float * 3), ("cam_matrix", C.c_float * 16)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("render_type", C.c_int32),
                ("aa_on", C.c_int32), ("max_depth", C.c_int32), ("vthreads", C.c_int32),
                ("background", C.c_float * 3), ("seed", C.c_uint64), ("nthreads", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("tri_tests", C.c_uint64)]


class CornellStorage(C.Structure):
    _fields_ = [("scene", Scene), ("spheres", Sphere * 3), ("lights", Light * 1),
                ("materials", Material * 11), ("meshes", Mesh * 1)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("is_hit", C.c_int32), ("prim_type", C.c_int32), ("prim_index", C.c_int32),
                ("mat_id", C.c_int32), ("mat", C.c_int32), ("normal", C.c_float * 3), ("tx", C.c_float), ("ty", C.c_float)]


class BvhCounters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("node_visits", C.c_uint64), ("leaf_visits", C.c_uint64)]


_lib = None


ASAN = os.environ.get("ART_ASAN", "") not in ("", "0")      # tests/run_sanitizers.sh: load the AddressSanitizer + UBSan builds


def build():
    """Compile oracle/liboracle.so (gcc, seconds)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR] + (["ASAN=1"] if ASAN else []))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "liboracle_asan.so" if ASAN else "liboracle.so")
    src = os.path.join(ORACLE_DIR, "art_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        build()
    L = C.CDLL(so)
    L.orc_sinf.restype = C.c_float; L.orc_sinf.argtypes = [C.c_float]
    L.orc_cosf.restype = C.c_float; L.orc_cosf.argtypes = [C.c_float]
    L.orc_tanf.restype = C.c_float; L.orc_tanf.argtypes = [C.c_float]
    L.orc_powf.restype = C.c_float; L.orc_powf.argtypes = [C.c_float, C.c_float]
    L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.orc_rng_uniform.restype = C.c_float
    L.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_load_vsgf.restype = C.c_int; L.orc_load_vsgf.argtypes = [C.c_char_p, f32p, C.POINTER(Mesh)]
    L.orc_free_mesh.argtypes = [C.POINTER(Mesh)]
    L.orc_cornell_mesh_transform.argtypes = [f32p]
    L.orc_build_cornell.argtypes = [C.POINTER(CornellStorage), C.POINTER(Mesh), C.c_int]
    L.orc_render_pass.argtypes = [C.POINTER(Scene), C.POINTER(Params), f32p, i32p, C.POINTER(Counters)]
    L.orc_render_pass_tasks.argtypes = [C.POINTER(Scene), C.POINTER(Params), f32p, i32p, C.POINTER(Counters)]
    L.orc_render_pass_tasks.restype = C.c_int
    L.orc_render_pixels.argtypes = [C.POINTER(Scene), C.POINTER(Params), i32p, i32p, C.c_int64, f32p, C.c_int32, C.POINTER(Counters)]
    L.orc_sample_radiance.argtypes = [C.POINTER(Scene), C.POINTER(Params), C.c_int32, C.c_int32, C.c_uint32, f32p]
    L.orc_debug_pass.argtypes = [C.POINTER(Scene), C.POINTER(Params), f32p, i32p, i32p, i32p]
    L.orc_resolve.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, u32p]
    L.orc_closest_hits.argtypes = [C.POINTER(Scene), f32p, f32p, C.c_int64, C.POINTER(Hit)]
    L.orc_save_bmp.restype = C.c_int; L.orc_save_bmp.argtypes = [C.c_char_p, u32p, C.c_int32, C.c_int32]
    L.orc_bmp_bytes.restype = C.c_int64
    L.orc_bmp_bytes.argtypes = [u32p, C.c_int32, C.c_int32, C.POINTER(C.c_uint8), C.c_int64]
    L.orc_bvh_walk_w.argtypes = [f32p, C.c_int32, f32p, C.c_int32, C.c_int32, f32p, f32p, f32p, C.c_int64, f32p, i32p,
                                 C.POINTER(BvhCounters)]
    _lib = L
    return L


def fp(a):
    return a.ctypes.data_as(f32p)


def ip(a):
    return a.ctypes.data_as(i32p)


def up(a):
    return a.ctypes.data_as(u32p)


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr); k = (C.c_uint32 * 2)(*key); o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


PYRAMID_VSGF = os.path.join(GOLDEN, "pyramid2.vsgf")


class CornellScene:
    """The reference's internal scene (scene.adb:89-217) built by the oracle."""

    def __init__(self, use_rect_light=False, vsgf_path=PYRAMID_VSGF):
        L = lib()
        self.T = np.zeros(16, np.float32)
        L.orc_cornell_mesh_transform(fp(self.T))
        self.mesh = Mesh()
        rc = L.orc_load_vsgf(vsgf_path.encode(), fp(self.T), C.byref(self.mesh))
        if rc != 0:
            raise RuntimeError("orc_load_vsgf failed: %d" % rc)
        self.storage = CornellStorage()
        L.orc_build_cornell(C.byref(self.storage), C.byref(self.mesh), int(use_rect_light))
        self.scene = self.storage.scene

    def mesh_arrays(self):
        m = self.mesh
        nv, nt = m.nverts, m.ntris
        return dict(pos=np.ctypeslib.as_array(m.pos, (nv, 3)).copy(), nrm=np.ctypeslib.as_array(m.nrm, (nv, 3)).copy(),
                    uv=np.ctypeslib.as_array(m.uv, (nv, 2)).copy(), idx=np.ctypeslib.as_array(m.idx, (nt, 3)).copy(),
                    matid=np.ctypeslib.as_array(m.matid, (nt,)).copy(),
                    bbmin=np.array(list(m.bbmin), np.float32), bbmax=np.array(list(m.bbmax), np.float32))


def make_params(width, height, render_type=PT_MIS, aa_on=True, max_depth=8, vthreads=1, seed=1, nthreads=0,
                background=(0.0, 0.0, 0.0)):
    p = Params()
    p.width, p.height, p.render_type, p.aa_on, p.max_depth, p.vthreads = width, height, render_type, int(aa_on), max_depth, vthreads
    p.background = (C.c_float * 3)(*background)
    p.seed = seed
    p.nthreads = nthreads
    return p


def render(scene, params, passes=1, accum=None, spp0=0):
    """Returns (accum[H,W,3] float32 row-major, spp, Counters)."""
    L = lib()
    if accum is None:
        accum = np.zeros((params.height, params.width, 3), np.float32)
    spp = C.c_int32(spp0)
    cnt = Counters()
    for _ in range(passes):
        L.orc_render_pass(C.byref(scene), C.byref(params), fp(accum), C.byref(spp), C.byref(cnt))
    return accum, spp.value, cnt


def render_pixels(scene, params, xs, ys, spp0=0, accum=None):
    """One Render_Pass for the listed pixels only (orc_render_pixels: the per-pixel body of orc_render_pass).  Returns (accum[n,3], Counters)."""
    xs = np.ascontiguousarray(xs, np.int32); ys = np.ascontiguousarray(ys, np.int32)
    if accum is None:
        accum = np.zeros((len(xs), 3), np.float32)
    cnt = Counters()
    lib().orc_render_pixels(C.byref(scene), C.byref(params), ip(xs), ip(ys), len(xs), fp(accum), spp0, C.byref(cnt))
    return accum, cnt


def render_tasks(scene, params, accum=None, spp0=0):
    """One pass organised like the reference's task pool (orc_render_pass_tasks): same bits as render()."""
    if accum is None:
        accum = np.zeros((params.height, params.width, 3), np.float32)
    spp = C.c_int32(spp0)
    cnt = Counters()
    if lib().orc_render_pass_tasks(C.byref(scene), C.byref(params), fp(accum), C.byref(spp), C.byref(cnt)):
        raise MemoryError("orc_render_pass_tasks")
    return accum, spp.value, cnt


def debug_pass(scene, params):
    L = lib()
    H, W = params.height, params.width
    accum = np.zeros((H, W, 3), np.float32)
    prim = np.zeros((H, W), np.int32); mat = np.zeros((H, W), np.int32); ptype = np.zeros((H, W), np.int32)
    L.orc_debug_pass(C.byref(scene), C.byref(params), fp(accum), ip(prim), ip(mat), ip(ptype))
    return accum, prim, mat, ptype


def resolve(accum, spp):
    H, W, _ = accum.shape
    out = np.zeros((H, W), np.uint32)
    lib().orc_resolve(fp(np.ascontiguousarray(accum)), W, H, spp, up(out))
    return out


def bmp_bytes(image_u32):
    H, W = image_u32.shape
    img = np.ascontiguousarray(image_u32)
    n = lib().orc_bmp_bytes(up(img), W, H, None, 0)
    buf = np.zeros(n, np.uint8)
    lib().orc_bmp_bytes(up(img), W, H, buf.ctypes.data_as(C.POINTER(C.c_uint8)), n)
    return buf.tobytes()


def closest_hits(scene, origins, dirs):
    o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(dirs, np.float32)
    n = o.shape[0]
    out = (Hit * n)()
    lib().orc_closest_hits(C.byref(scene), fp(o), fp(d), n, out)
    return out


def bvh_walk(nodes, tris, origins, dirs, tfar=None, width=8):
    nodes = np.ascontiguousarray(nodes, np.float32); tris = np.ascontiguousarray(tris, np.float32)
    o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(dirs, np.float32)
    n = o.shape[0]
    t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32); cnt = BvhCounters()
    tf = None if tfar is None else fp(np.ascontiguousarray(tfar, np.float32))
    lib().orc_bvh_walk_w(fp(nodes), nodes.size // (8 * width), fp(tris), tris.size // 12, width, fp(o), fp(d), tf, n, fp(t), ip(prim), C.byref(cnt))
    return t, prim, cnt

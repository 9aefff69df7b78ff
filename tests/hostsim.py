"""ctypes binding of tests/host_sim/libhost_sim.so (TEST-ONLY CPU build of the product's per-slot device functions)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib(art):
    global _lib
    if _lib is None:
        asan = os.environ.get("ART_ASAN", "") not in ("", "0")            # tests/run_sanitizers.sh
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host_sim")] + (["ASAN=1"] if asan else []))
        L = C.CDLL(os.path.join(HERE, "host_sim", "libhost_sim_asan.so" if asan else "libhost_sim.so"))
        L.hs_last_error.restype = C.c_char_p
        L.hs_render.argtypes = [C.POINTER(art.ArtSceneDesc), C.POINTER(art.ArtPassParams), C.c_int, C.c_int, C.c_int, art.f32p, C.POINTER(C.c_uint64)]
        L.hs_trace.argtypes = [C.POINTER(art.ArtSceneDesc), art.f32p, art.f32p, art.f32p, C.c_longlong, C.POINTER(art.ArtHit), C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def pixmap(art, w, h, rank, nranks, tile=32):
    """the pixels (y * w + x) rank owns in an nranks-way job: the product's build_pixmap"""
    L = lib(art)
    L.hs_pixmap.restype = C.c_longlong
    L.hs_pixmap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_longlong]
    out = np.zeros(w * h, np.uint32)
    n = L.hs_pixmap(w, h, rank, nranks, tile, out.ctypes.data_as(C.POINTER(C.c_uint32)), out.size)
    return out[:n].copy()


def flattened_copy(art, sd):
    """the explicit world-space flattening of an instanced art.SceneDesc (the product's flatten_instances through tests/host_sim): a new
    SceneDesc with ONE ART_MESH_CLOSEST mesh, triangles in the order (instance, triangle of the mesh), and no instances"""
    L = lib(art)
    L.hs_flatten_instances.argtypes = [C.POINTER(art.ArtSceneDesc), art.f32p, art.f32p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_longlong)]
    cnt = (C.c_longlong * 2)()
    assert L.hs_flatten_instances(C.byref(sd.desc), None, None, None, None, cnt) == 0, L.hs_last_error()
    nv, nt = int(cnt[0]), int(cnt[1])
    pos = np.zeros((nv, 3), np.float32); nrm = np.zeros((nv, 3), np.float32); idx = np.zeros((nt, 3), np.int32); matid = np.zeros(nt, np.int32)
    i32p = C.POINTER(C.c_int32)
    assert L.hs_flatten_instances(C.byref(sd.desc), pos.ctypes.data_as(art.f32p), nrm.ctypes.data_as(art.f32p), idx.ctypes.data_as(i32p), matid.ctypes.data_as(i32p), cnt) == 0
    return art.SceneDesc(meshes=[dict(mode=art.MESH_CLOSEST, pos=pos, nrm=nrm, idx=idx, matid=matid)], **sd._kw)


def set_fold_dense(art, on):
    """1: the fold over dense per-level records (the GPU's compacted schedule), 0: the slot-indexed fold stack"""
    lib(art).hs_set_fold_dense(int(on))


def set_skip_null_shadow(art, on):
    """DevFrame::skip_null_shadow of the following renders"""
    lib(art).hs_set_skip_null_shadow(int(on))


def set_shard(art, rank, nranks, tile=32):
    lib(art).hs_set_shard(rank, nranks, tile)


def render(art, sd, params, w, h, spp0=0, accum=None):
    if accum is None:
        accum = np.zeros((h, w, 3), np.float32)
    rays = C.c_uint64(0)
    rc = lib(art).hs_render(C.byref(sd.desc), C.byref(params), w, h, spp0, accum.ctypes.data_as(art.f32p), C.byref(rays))
    if rc:
        raise RuntimeError(lib(art).hs_last_error().decode())
    return accum, rays.value


def trace(art, sd, origins, dirs, tfar=None):
    o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(dirs, np.float32)
    n = o.shape[0]
    out = (art.ArtHit * n)()
    st = (C.c_uint64 * 4)()
    tf = None if tfar is None else np.ascontiguousarray(tfar, np.float32).ctypes.data_as(art.f32p)
    rc = lib(art).hs_trace(C.byref(sd.desc), o.ctypes.data_as(art.f32p), d.ctypes.data_as(art.f32p), tf, n, out, st)
    if rc:
        raise RuntimeError(lib(art).hs_last_error().decode())
    return out, [int(v) for v in st]


def bvh(art, sd):
    L = lib(art)
    info = (C.c_int * 4)()
    L.hs_bvh.argtypes = [C.POINTER(art.ArtSceneDesc), art.f32p, C.c_longlong, art.f32p, C.c_longlong, C.POINTER(C.c_int)]
    if L.hs_bvh(C.byref(sd.desc), None, 0, None, 0, info):
        raise RuntimeError(L.hs_last_error().decode())
    nodes = np.zeros(info[0] * 8 * info[3], np.float32); tris = np.zeros(info[1] * 12, np.float32)
    if L.hs_bvh(C.byref(sd.desc), nodes.ctypes.data_as(art.f32p), nodes.size, tris.ctypes.data_as(art.f32p), tris.size, info):
        raise RuntimeError("hs_bvh failed")
    return nodes, tris, dict(n_nodes=info[0], n_tris=info[1], max_stack=info[2], width=info[3])


def set_bvh_param(art, name, value):
    """Builder parameter (art_bvh.h BvhBuildParams) for the following host-simulation builds."""
    L = lib(art)
    L.hs_set_bvh_param.argtypes = [C.c_char_p, C.c_double]
    L.hs_set_bvh_param.restype = None
    L.hs_set_bvh_param(name.encode(), float(value))


def awkward_instances():
    """Instance transforms the seeded scenes do not contain: the identity, a mirror image (negative determinant: the triangles' winding as
    seen from outside flips, so does the one-sided test of geometry.adb:231-263 -- in the flattened scene exactly the same way), the SAME
    transform twice (coincident triangles: equal t, the lower hit index must win in both renderings), a shear, a tiny and a large instance
    (scale 1e-3 / 2.2, the large one cutting through the walls of the box), and two instances that interpenetrate."""
    def M(a, t):
        m = np.zeros((3, 4)); m[:, :3] = np.asarray(a, np.float64); m[:, 3] = t
        return m
    I = np.eye(3)
    rot = np.array([[0.8, 0.0, 0.6], [0.0, 1.0, 0.0], [-0.6, 0.0, 0.8]])
    return [(0, M(I, (0.0, 1.2, 2.0))),
            (0, M(np.diag([-1.0, 1.0, 1.0]) * 0.9, (1.3, 2.6, 2.4))),
            (1, M(rot * 0.8, (-1.0, 2.2, 1.5))),
            (1, M(rot * 0.8, (-1.0, 2.2, 1.5))),
            (0, M(np.array([[1.0, 0.6, 0.0], [0.0, 1.0, 0.3], [0.0, 0.0, 1.0]]) * 0.7, (-1.2, 3.6, 3.2))),
            (0, M(I * 1.0e-3, (0.2, 2.5, 4.0))),
            (1, M(rot.T * 2.2, (0.0, 0.4, 2.5))),
            (0, M(rot * 0.6, (0.9, 1.0, 3.3))),
            (1, M(np.diag([0.5, -0.7, 0.5]), (1.0, 1.1, 3.2)))]

"""ada-ray-tracer_amd -- host-side binding of libart_hip.so (MI355X / gfx950 render backend).

The product is the C ABI in include/art_hip.h; this module is the thin ctypes layer the tests, bench.py
and the multi-GPU harness use, plus a mirror of the reference's host interface
(Ray_Tracer.Init_Render / Resize_Viewport / Render_Pass / GetSPP / Finished, ray_tracer.ads:40-48)
so that callers read like test.adb:32-75.

The directory name contains a hyphen, so import it through ``__graft_entry__.load_package()`` (which
registers it as ``ada_ray_tracer_amd``).  There is no CPU fallback: every call needs the HIP library
and a GPU and raises ``ArtError`` otherwise.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("ART_LIB", os.path.join(PKG_DIR, "libart_hip.so"))   # ART_LIB: A/B builds of the same source
# the C++ mirror of the Ada host layer (host/); ART_ASAN=1: its AddressSanitizer + UBSan build (tests/run_sanitizers.sh, CPU only)
HOST_LIB_PATH = os.path.join(PKG_DIR, "libart_host_asan.so" if os.environ.get("ART_ASAN", "") not in ("", "0") else "libart_host.so")

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)

MAT_NULL, MAT_LIGHT, MAT_LAMBERT, MAT_MIRROR, MAT_GLASS, MAT_PHONG = range(6)
LIGHT_RECT, LIGHT_SPHERE = 0, 1
MESH_REFERENCE_BF, MESH_CLOSEST = 0, 1
RT_DEBUG, RT_WHITTED, PT_STUPID, PT_SHADOW, PT_MIS = range(5)
LAYOUT_ADA_XY, LAYOUT_ROW_MAJOR = 0, 1
TRACE_COOP, TRACE_SIMPLE = 0, 1
BVH_HOST_SAH, BVH_GPU_LBVH, BVH_GPU_PLOC, BVH_GPU_SAH = 0, 1, 2, 3     # option "bvh_builder"
DEFAULT_BVH_BUILDER = BVH_GPU_SAH


class ArtError(RuntimeError):
    pass


class ArtMaterial(C.Structure):
    _fields_ = [("type", C.c_int32), ("light", C.c_int32), ("p", C.c_float * 8)]


class ArtLight(C.Structure):
    _fields_ = [("shape", C.c_int32), ("mat", C.c_int32),
                ("boxMin", C.c_float * 3), ("boxMax", C.c_float * 3), ("normal", C.c_float * 3),
                ("center", C.c_float * 3), ("radius", C.c_float),
                ("intensity", C.c_float * 3), ("surfaceArea", C.c_float)]


class ArtSphere(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("r", C.c_float), ("mat", C.c_int32)]


class ArtMesh(C.Structure):
    _fields_ = [("mode", C.c_int32), ("nverts", C.c_int32), ("ntris", C.c_int32),
                ("pos", f32p), ("nrm", f32p), ("uv", f32p), ("idx", i32p), ("matid", i32p),
                ("bbmin", C.c_float * 3), ("bbmax", C.c_float * 3)]


class ArtInstance(C.Structure):
    _fields_ = [("mesh", C.c_int32), ("m", C.c_float * 12)]


class ArtSceneDesc(C.Structure):
    _fields_ = [("n_spheres", C.c_int32), ("spheres", C.POINTER(ArtSphere)),
                ("has_cornell", C.c_int32),
                ("cb_min", C.c_float * 3), ("cb_max", C.c_float * 3),
                ("cb_mat", C.c_int32 * 6), ("cb_nrm", (C.c_float * 3) * 6),
                ("n_lights", C.c_int32), ("lights", C.POINTER(ArtLight)),
                ("n_materials", C.c_int32), ("materials", C.POINTER(ArtMaterial)),
                ("n_meshes", C.c_int32), ("meshes", C.POINTER(ArtMesh)),
                ("cam_pos", C.c_float * 3), ("cam_matrix", C.c_float * 16),
                ("n_instances", C.c_int32), ("instances", C.POINTER(ArtInstance))]


class ArtPassParams(C.Structure):
    _fields_ = [("render_type", C.c_int32), ("aa_on", C.c_int32), ("max_depth", C.c_int32), ("vthreads", C.c_int32),
                ("background", C.c_float * 3), ("seed", C.c_uint64), ("layout", C.c_int32)]


class ArtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("trace_ms", C.c_double), ("pass_ms", C.c_double),
                ("trace_launches", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("node_visits", C.c_uint64), ("leaf_visits", C.c_uint64), ("traced_rays", C.c_uint64),
                ("node_phase_iters", C.c_uint64), ("leaf_phase_iters", C.c_uint64), ("wave_iters", C.c_uint64), ("lost_paths", C.c_uint64)]


class ArtStageStats(C.Structure):
    _fields_ = [("shade_ms", C.c_double), ("raygen_ms", C.c_double), ("fold_ms", C.c_double), ("shade_launches", C.c_uint64), ("batches", C.c_uint64),
                ("items_in", C.c_uint64 * 16), ("items_out", C.c_uint64 * 16)]


class ArtReduceInfo(C.Structure):
    _fields_ = [("devices", C.c_int32), ("rccl_ranks", C.c_int32), ("path", C.c_int32), ("reduces", C.c_int32),
                ("reduce_ms", C.c_double), ("device_pass_ms", C.c_double * 8),
                ("device_busy_ms", C.c_double * 8), ("device_idle_ms", C.c_double * 8), ("device_start_skew_ms", C.c_double * 8),
                ("passes", C.c_int32), ("passes_overlapped", C.c_int32)]


class ArtHit(C.Structure):
    _fields_ = [("t", C.c_float), ("is_hit", C.c_int32), ("prim_type", C.c_int32), ("prim_index", C.c_int32),
                ("mat_id", C.c_int32), ("mat", C.c_int32), ("normal", C.c_float * 3), ("u", C.c_float), ("v", C.c_float)]


class ArtBvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_tris", C.c_int32), ("max_stack", C.c_int32), ("node_width", C.c_int32),
                ("build_ms", C.c_double)]


class HitCpp(C.Structure):
    _fields_ = [("primIndex", C.c_int32), ("geomIndex", C.c_int32), ("instIndex", C.c_int32), ("t", C.c_float),
                ("normal", C.c_float * 3), ("texCoord", C.c_float * 2)]


EXPORTED_SYMBOLS = [
    "art_init", "art_init_devices", "art_device_count", "art_reduce", "art_get_reduce_info", "art_set_stream", "art_upload_scene", "art_resize", "art_set_shard", "art_render_pass",
    "art_debug_hit_pass", "art_bind_accum", "art_accum_device", "art_download", "art_synchronize", "art_trace_rays",
    "art_export_bvh", "art_get_stats", "art_get_stage_stats", "art_set_option", "art_last_error", "art_shutdown",
    "gcore_init_and_clear", "gcore_destroy", "gcore_add_mesh_3f", "gcore_instance_meshes", "gcore_commit_scene",
    "gcore_closest_hit", "gcore_closest_hit_n", "gcore_set_two_level", "gcore_set_single_ray_on_gpu",
]


def build_library(force=False):
    """hipcc cross-compiles libart_hip.so for gfx950 (works without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", PKG_DIR, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", PKG_DIR])
    return LIB_PATH


_lib = None


def load_library():
    """dlopen the in-tree libart_hip.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ArtError("libart_hip.so is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no fallback path")
    L = C.CDLL(LIB_PATH)
    L.art_last_error.restype = C.c_char_p
    L.art_accum_device.restype = C.c_void_p
    L.art_set_stream.argtypes = [C.c_void_p]
    L.art_init_devices.argtypes = [C.c_int32, i32p]
    L.art_upload_scene.argtypes = [C.POINTER(ArtSceneDesc)]
    L.art_resize.argtypes = [C.c_int32, C.c_int32]
    L.art_set_shard.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    L.art_render_pass.argtypes = [C.POINTER(ArtPassParams), f32p, u32p, i32p]
    L.art_debug_hit_pass.argtypes = [C.POINTER(ArtPassParams), f32p, u32p, i32p, i32p, i32p]
    L.art_bind_accum.argtypes = [C.c_void_p]
    L.art_download.argtypes = [f32p, u32p, C.c_int32, C.c_int32]
    L.art_trace_rays.argtypes = [f32p, f32p, f32p, C.c_int64, C.POINTER(ArtHit), C.c_int32, C.POINTER(ArtStats)]
    L.art_export_bvh.argtypes = [f32p, C.c_int64, f32p, C.c_int64, C.POINTER(ArtBvhInfo)]
    L.art_get_stats.argtypes = [C.POINTER(ArtStats)]
    L.art_get_reduce_info.argtypes = [C.POINTER(ArtReduceInfo)]
    L.art_get_stage_stats.argtypes = [C.POINTER(ArtStageStats)]
    L.art_set_option.argtypes = [C.c_char_p, C.c_int64]
    L.gcore_add_mesh_3f.argtypes = [f32p, C.c_int, i32p, C.c_int]
    L.gcore_add_mesh_3f.restype = C.c_int
    L.gcore_instance_meshes.argtypes = [C.c_int, f32p, C.c_int]
    L.gcore_closest_hit.argtypes = [f32p, f32p, C.c_float, C.c_float, C.POINTER(HitCpp)]
    L.gcore_closest_hit.restype = C.c_bool
    L.gcore_closest_hit_n.argtypes = [C.c_int, f32p, f32p, f32p, f32p, C.POINTER(HitCpp), C.POINTER(C.c_ubyte)]
    L.gcore_closest_hit_n.restype = C.c_int
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise ArtError(load_library().art_last_error().decode())


def _fp(a):
    return None if a is None else a.ctypes.data_as(f32p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(i32p)


def _up(a):
    return None if a is None else a.ctypes.data_as(u32p)


class SceneDesc:
    """Flattened scene held in numpy arrays (keeps them alive) + the ArtSceneDesc view of them."""

    def __init__(self, spheres=(), lights=(), materials=(), meshes=(), cornell=None,
                 cam_pos=(0.0, 2.55, 12.5), cam_matrix=None, instances=()):
        """instances: [(mesh index, 12 floats: object -> world 3x4 row-major)] -- then `meshes` are object-space prototypes (ArtSceneDesc::n_instances)"""
        self._kw = dict(spheres=spheres, lights=lights, materials=materials, cornell=cornell, cam_pos=cam_pos, cam_matrix=cam_matrix)      # (for flattened_copy)
        self.spheres = (ArtSphere * max(1, len(spheres)))()
        for i, (pos, r, mat) in enumerate(spheres):
            self.spheres[i].pos = (C.c_float * 3)(*pos); self.spheres[i].r = r; self.spheres[i].mat = mat
        self.lights = (ArtLight * max(1, len(lights)))()
        for i, l in enumerate(lights):
            L = self.lights[i]
            L.shape = l["shape"]; L.mat = l["mat"]
            for k in ("boxMin", "boxMax", "normal", "center", "intensity"):
                setattr(L, k, (C.c_float * 3)(*l.get(k, (0.0, 0.0, 0.0))))
            L.radius = l.get("radius", 0.0); L.surfaceArea = l["surfaceArea"]
        self.materials = (ArtMaterial * max(1, len(materials)))()
        for i, m in enumerate(materials):
            M = self.materials[i]
            M.type = m["type"]; M.light = m.get("light", 0)
            p = list(m.get("p", ())) + [0.0] * 8
            M.p = (C.c_float * 8)(*p[:8])
        self._mesh_arrays = []
        self.meshes = (ArtMesh * max(1, len(meshes)))()
        for i, m in enumerate(meshes):
            pos = np.ascontiguousarray(m["pos"], np.float32).reshape(-1, 3)
            nrm = np.ascontiguousarray(m["nrm"], np.float32).reshape(-1, 3)
            idx = np.ascontiguousarray(m["idx"], np.int32).reshape(-1, 3)
            uv = np.ascontiguousarray(m.get("uv", np.zeros((pos.shape[0], 2))), np.float32).reshape(-1, 2)
            matid = np.ascontiguousarray(m.get("matid", np.zeros(idx.shape[0])), np.int32)
            self._mesh_arrays.append((pos, nrm, idx, uv, matid))
            M = self.meshes[i]
            M.mode = m["mode"]; M.nverts = pos.shape[0]; M.ntris = idx.shape[0]
            M.pos = _fp(pos); M.nrm = _fp(nrm); M.uv = _fp(uv); M.idx = _ip(idx); M.matid = _ip(matid)
            bbmin = m.get("bbmin", pos.min(0)); bbmax = m.get("bbmax", pos.max(0))
            M.bbmin = (C.c_float * 3)(*[float(v) for v in bbmin]); M.bbmax = (C.c_float * 3)(*[float(v) for v in bbmax])
        d = ArtSceneDesc()
        d.n_spheres = len(spheres); d.spheres = self.spheres
        d.n_lights = len(lights); d.lights = self.lights
        d.n_materials = len(materials); d.materials = self.materials
        d.n_meshes = len(meshes); d.meshes = self.meshes
        if cornell is not None:
            d.has_cornell = 1
            d.cb_min = (C.c_float * 3)(*cornell["min"]); d.cb_max = (C.c_float * 3)(*cornell["max"])
            d.cb_mat = (C.c_int32 * 6)(*cornell["mat"])
            for k in range(6):
                d.cb_nrm[k] = (C.c_float * 3)(*cornell["nrm"][k])
        d.cam_pos = (C.c_float * 3)(*cam_pos)
        cm = np.eye(4, dtype=np.float32).ravel() if cam_matrix is None else np.asarray(cam_matrix, np.float32).ravel()
        d.cam_matrix = (C.c_float * 16)(*[float(v) for v in cm])
        self.instances = (ArtInstance * max(1, len(instances)))()
        for i, (mesh, m) in enumerate(instances):
            self.instances[i].mesh = int(mesh); self.instances[i].m = (C.c_float * 12)(*[float(v) for v in np.asarray(m, np.float32).ravel()[:12]])
        d.n_instances = len(instances); d.instances = self.instances
        self.desc = d


class Backend:
    """One process-wide backend instance (the C library is a singleton, like g_data / ray_tracer.ads globals)."""

    def __init__(self, device=-1, devices=None):
        """device: one GPU (art_init).  devices: a list of ordinals, or a count n for 0..n-1 -> one process drives them all
        (art_init_devices): pixel tiles sharded inside the library, RCCL reduce to the first device."""
        self.lib = load_library()
        if devices is None:
            _check(self.lib.art_init(device))
        elif isinstance(devices, int):
            _check(self.lib.art_init_devices(devices, None))
        else:
            arr = (C.c_int32 * len(devices))(*devices)
            _check(self.lib.art_init_devices(len(devices), arr))

    def reduce(self):
        _check(self.lib.art_reduce())

    def set_stream(self, hip_stream):
        _check(self.lib.art_set_stream(hip_stream))

    def set_option(self, name, value):
        _check(self.lib.art_set_option(name.encode(), int(value)))

    def upload_scene(self, scene):
        desc = getattr(scene, "desc", scene)          # SceneDesc / scenes.HostSceneDesc, or a raw ArtSceneDesc
        _check(self.lib.art_upload_scene(C.byref(desc)))

    def resize(self, width, height):
        self.width, self.height = width, height
        _check(self.lib.art_resize(width, height))

    def set_shard(self, rank, nranks, tile=32):
        _check(self.lib.art_set_shard(rank, nranks, tile))

    @staticmethod
    def pass_params(render_type=PT_MIS, aa_on=True, max_depth=8, vthreads=1, seed=1, background=(0.0, 0.0, 0.0),
                    layout=LAYOUT_ROW_MAJOR):
        p = ArtPassParams()
        p.render_type, p.aa_on, p.max_depth, p.vthreads = render_type, int(aa_on), max_depth, vthreads
        p.background = (C.c_float * 3)(*background); p.seed = seed; p.layout = layout
        return p

    def render_pass(self, params, spp, want_accum=True, want_screen=False):
        """Returns (accum or None, screen or None, new spp).  Arrays are [H,W,..] for ROW_MAJOR, [W,H,..] for ADA_XY."""
        shape = (self.height, self.width) if params.layout == LAYOUT_ROW_MAJOR else (self.width, self.height)
        accum = np.zeros(shape + (3,), np.float32) if want_accum else None
        screen = np.zeros(shape, np.uint32) if want_screen else None
        s = C.c_int32(spp)
        _check(self.lib.art_render_pass(C.byref(params), _fp(accum), _up(screen), C.byref(s)))
        return accum, screen, s.value

    def render_pass_device(self, params, spp):
        s = C.c_int32(spp)
        _check(self.lib.art_render_pass(C.byref(params), None, None, C.byref(s)))
        return s.value

    def debug_hit_pass(self, params):
        shape = (self.height, self.width) if params.layout == LAYOUT_ROW_MAJOR else (self.width, self.height)
        accum = np.zeros(shape + (3,), np.float32); screen = np.zeros(shape, np.uint32)
        prim = np.zeros(shape, np.int32); mat = np.zeros(shape, np.int32); ptype = np.zeros(shape, np.int32)
        _check(self.lib.art_debug_hit_pass(C.byref(params), _fp(accum), _up(screen), _ip(prim), _ip(mat), _ip(ptype)))
        return accum, screen, prim, mat, ptype

    def download(self, spp, layout=LAYOUT_ROW_MAJOR, want_screen=True):
        shape = (self.height, self.width) if layout == LAYOUT_ROW_MAJOR else (self.width, self.height)
        accum = np.zeros(shape + (3,), np.float32)
        screen = np.zeros(shape, np.uint32) if want_screen else None
        _check(self.lib.art_download(_fp(accum), _up(screen), layout, spp))
        return accum, screen

    def bind_accum(self, device_ptr):
        _check(self.lib.art_bind_accum(device_ptr))

    def synchronize(self):
        _check(self.lib.art_synchronize())

    def stage_stats(self):
        """the wavefront stages around the trace kernel: GPU ms per kind of kernel, items read / kept per bounce (device 0, cumulative)"""
        st = ArtStageStats()
        _check(self.lib.art_get_stage_stats(C.byref(st)))
        return st

    def reduce_info(self):
        """what the multi-device path did: ranks of the RCCL communicator, GPU time of the reduces, GPU time of every device's passes"""
        ri = ArtReduceInfo()
        _check(self.lib.art_get_reduce_info(C.byref(ri)))
        return ri

    def trace_rays(self, origins, dirs, tfar=None, kernel=TRACE_COOP, want_stats=False):
        o = np.ascontiguousarray(origins, np.float32); d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        out = (ArtHit * n)()
        st = ArtStats()
        tf = None if tfar is None else np.ascontiguousarray(tfar, np.float32)
        _check(self.lib.art_trace_rays(_fp(o), _fp(d), _fp(tf), n, out, kernel, C.byref(st) if want_stats else None))
        return (out, st) if want_stats else out

    def bvh_info(self):
        info = ArtBvhInfo()
        _check(self.lib.art_export_bvh(None, 0, None, 0, C.byref(info)))
        return info

    def export_bvh(self):
        info = ArtBvhInfo()
        _check(self.lib.art_export_bvh(None, 0, None, 0, C.byref(info)))
        nodes = np.zeros(info.n_nodes * 8 * info.node_width, np.float32); tris = np.zeros(info.n_tris * 12, np.float32)
        _check(self.lib.art_export_bvh(_fp(nodes), nodes.size, _fp(tris), tris.size, C.byref(info)))
        return nodes, tris, info

    def stats(self):
        st = ArtStats()
        _check(self.lib.art_get_stats(C.byref(st)))
        return st

    def shutdown(self):
        self.lib.art_shutdown()


class RayTracer:
    """Mirror of package Ray_Tracer (ray_tracer.ads:18-48) on top of the backend.

    width/height/Threads_Num/Anti_Aliasing_On/Max_Trace_Depth/Background_Color are the package variables of
    ray_tracer.ads:20-27; Render_Pass forwards to art_render_pass instead of waking Path_Trace_Thread tasks."""

    def __init__(self, backend, scene):
        self.backend = backend
        backend.upload_scene(scene)
        self.width, self.height = 1024, 768          # ray_tracer.ads:20-21
        self.Threads_Num = 14 * 2                     # ray_tracer.ads:23
        self.Anti_Aliasing_On = True                  # ray_tracer.ads:24
        self.Max_Trace_Depth = 8                      # ray_tracer.ads:25
        self.Background_Color = (0.0, 0.0, 0.0)       # ray_tracer.ads:27
        self.seed = 1
        self.g_rend_type = PT_MIS
        self.g_finish = False
        self.g_spp = 0
        self.screen_buffer = None                     # ScreenBufferData(x, y): [width, height] u32
        self.g_accBuff = None                         # AccumBuff(x, y): [width, height, 3] f32

    def Init_Render(self, a_rendType):                # ray_tracer.adb:197-200
        self.g_rend_type = a_rendType

    def Resize_Viewport(self, size_x, size_y):        # ray_tracer.adb:297-320
        self.width, self.height = size_x, size_y
        self.backend.resize(size_x, size_y)
        self.g_spp = 0

    def GetSPP(self):                                 # ray_tracer.adb:322-325
        return self.g_spp

    def Finished(self):                               # ray_tracer.adb:202-205
        return self.g_finish

    def Render_Pass(self):                            # ray_tracer.adb:240-293
        p = Backend.pass_params(self.g_rend_type, self.Anti_Aliasing_On, self.Max_Trace_Depth, self.Threads_Num, self.seed,
                                self.Background_Color, LAYOUT_ADA_XY)
        if self.g_rend_type in (RT_DEBUG, RT_WHITTED):
            self.g_accBuff, self.screen_buffer, _, _, _ = self.backend.debug_hit_pass(p)
            self.g_finish = True
            return
        self.g_accBuff, self.screen_buffer, self.g_spp = self.backend.render_pass(p, self.g_spp, True, True)


def save_bmp(path, image_u32_rowmajor):
    """Bitmap.SaveBMP (bitmap.adb:31-85): 14+40 byte headers, then per pixel the bytes (bits 16-23, 8-15, 0-7), no row padding."""
    img = np.ascontiguousarray(image_u32_rowmajor, np.uint32)
    h, w = img.shape
    hdr = np.zeros(54, np.uint8)
    hdr[0:2] = (0x42, 0x4D)
    hdr[2:6] = np.frombuffer(np.uint32(54 + w * h * 3).tobytes(), np.uint8)
    hdr[10:14] = np.frombuffer(np.uint32(54).tobytes(), np.uint8)
    hdr[14:18] = np.frombuffer(np.uint32(40).tobytes(), np.uint8)
    hdr[18:22] = np.frombuffer(np.uint32(w).tobytes(), np.uint8)
    hdr[22:26] = np.frombuffer(np.uint32(h).tobytes(), np.uint8)
    hdr[26:28] = (1, 0); hdr[28:30] = (24, 0)
    px = np.stack([(img >> 16) & 255, (img >> 8) & 255, img & 255], -1).astype(np.uint8)
    data = hdr.tobytes() + px.tobytes()
    if path is not None:
        with open(path, "wb") as f:
            f.write(data)
    return data

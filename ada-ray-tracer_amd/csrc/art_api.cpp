// art_api.cpp -- C-ABI driver of the render backend (include/art_hip.h).  Owns the HBM-resident scene,
// the wavefront path buffers and the per-pass kernel schedule that replaces Ray_Tracer.Render_Pass'
// task pool (ray_tracer.adb:240-293):
//
//   per batch of P = pixels x samples path slots:
//     raygen -> [ trace(ext + shadow rays) -> shade(bounce) ] x max_depth -> trace(last shadow rays)
//            -> finish (inside-out radiance fold) -> accumulate (reference summation order)
//   then resolve (gamma, clamp, pack) when the caller asks for the LDR image.
//
// There is no CPU fallback: every entry point that needs the GPU fails with art_last_error() set when
// HIP reports no usable device.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "art_api_internal.h"
#include "art_lbvh.h"

namespace art {

Ctx g_devs[kMaxDevices];
int g_ndev = 1;
Ctx* g_cur = &g_devs[0];
std::mutex g_mu;
static ncclComm_t g_comms[kMaxDevices];
static bool g_comms_ready = false;     // distinct GPUs: the framebuffer reduce goes through RCCL
// events around every reduce on device 0's stream (art_get_reduce_info: the reduce's GPU time is part of the evidence of a multi-GPU run)
static std::vector<hipEvent_t> g_reduce_events;
static double g_reduce_ms = 0.0; static int g_reduces = 0, g_reduce_path = 0;
static bool g_multi_api = false;       // the process came up through art_init_devices: passes carry host-clock marks (art_get_reduce_info), also with n = 1
static std::vector<hipEvent_t> g_reduce_free;     // event pairs folded into g_reduce_ms, ready for the next reduce (a host that downloads every frame re-uses two events)
static int g_passes = 0, g_passes_overlapped = 0;  // multi-device passes folded so far / those in which every device had started before any had finished
static void reset_reduce_info() {
  for (hipEvent_t e : g_reduce_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : g_reduce_free) (void)hipEventDestroy(e);
  g_reduce_events.clear(); g_reduce_free.clear(); g_reduce_ms = 0.0; g_reduces = 0; g_reduce_path = 0;
  g_passes = g_passes_overlapped = 0;
}
// completed pairs -> g_reduce_ms (device 0 current, its stream idle)
static void fold_reduce_events() {
  for (size_t i = 0; i + 1 < g_reduce_events.size(); i += 2) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, g_reduce_events[i], g_reduce_events[i + 1]) == hipSuccess) g_reduce_ms += ms;
    else (void)hipGetLastError();
    g_reduce_free.push_back(g_reduce_events[i]); g_reduce_free.push_back(g_reduce_events[i + 1]);
  }
  g_reduce_events.clear();
}
// multi-device mode: the host clock at which a device's stream reaches a point (hipLaunchHostFunc), so that the devices' pass times share
// ONE clock -- HIP events of different devices cannot be compared
static double host_ms() {
  static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
static void clock_cb(void* p) { *(double*)p = host_ms(); }
// every device's passes (all idle) -> busy / idle / start skew per device.  Pass i of every device is one Render_Pass: idle = the slowest
// device's end - this device's end (what an uneven tile deal costs), start skew = this device's start - the first device's start (what a
// host that enqueues device after device costs), overlapped = every device had started before any had finished.
static void fold_pass_clocks() {
  size_t n = SIZE_MAX;
  for (int k = 0; k < g_ndev; ++k) n = std::min(n, g_devs[k].pass_clock.size());
  if (!g_multi_api || n == SIZE_MAX) n = 0;
  for (size_t i = 0; i < n; ++i) {
    double first = 1e300, last_start = -1e300, first_end = 1e300, last = -1e300;
    for (int k = 0; k < g_ndev; ++k) { const Ctx::PassClock& pc = *g_devs[k].pass_clock[i]; first = std::min(first, pc.t0); last_start = std::max(last_start, pc.t0); first_end = std::min(first_end, pc.t1); last = std::max(last, pc.t1); }
    for (int k = 0; k < g_ndev; ++k) { Ctx& c = g_devs[k]; const Ctx::PassClock& pc = *c.pass_clock[i]; c.busy_ms += pc.t1 - pc.t0; c.idle_ms += last - pc.t1; c.start_skew_ms += pc.t0 - first; }
    g_passes += 1; if (last_start < first_end) g_passes_overlapped += 1;
  }
  for (int k = 0; k < g_ndev; ++k) { for (Ctx::PassClock* pc : g_devs[k].pass_clock) delete pc; g_devs[k].pass_clock.clear(); }
}
static bool g_same_gpu = false;        // rehearsal: several contexts on ONE physical GPU, the reduce is a local sum (no collective possible)
static const bool g_debug_live = getenv("ART_DEBUG_LIVE") != nullptr;   // development aid: work-set sizes per stage on stderr (syncs the stream); read once
static const bool g_debug_addr = getenv("ART_DEBUG_ADDR") != nullptr;   // development aid: device addresses of the path state per batch layout on stderr (profiles/r6_bimodal)
static thread_local std::string t_err;
static std::string g_err;

int fail(const std::string& msg) { g_err = msg; return 1; }

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) return fail(std::string(#expr) + ": " + hipGetErrorString(_e));   \
  } while (0)

// make device k's context the current one (g_ctx) and its GPU the current HIP device
static int use_dev(int k) {
  g_cur = &g_devs[k];
  if (g_cur->device >= 0) HIP_TRY(hipSetDevice(g_cur->device));
  return 0;
}
// multi-device loops leave device 0 current on every exit path
struct Dev0Guard { ~Dev0Guard() { g_cur = &g_devs[0]; if (g_devs[0].device >= 0) (void)hipSetDevice(g_devs[0].device); } };

template <typename T>
static int upload(DevBuf& b, const std::vector<T>& v) {
  b.release();
  if (v.empty()) return 0;
  HIP_TRY(hipMalloc(&b.p, v.size() * sizeof(T)));
  b.bytes = v.size() * sizeof(T);
  HIP_TRY(hipMemcpy(b.p, v.data(), b.bytes, hipMemcpyHostToDevice));
  return 0;
}

static int ensure(DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return 0;
  b.release();
  HIP_TRY(hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return 0;
}

void DevBuf::release() {
  if (reserved) {                                        // a mapped range (alloc_spread)
    for (size_t i = 0; i < chunks.size(); ++i) { (void)hipMemUnmap((char*)p + i * chunk_bytes, chunk_bytes); (void)hipMemRelease(chunks[i]); }
    (void)hipMemAddressFree(p, reserved);
    chunks.clear(); reserved = 0; chunk_bytes = 0;
  } else if (p) (void)hipFree(p);
  p = nullptr; bytes = 0;
}

// `bytes` of device memory as ONE address range over separately created physical chunks of `chunk` bytes (Ctx::paths_spread_mb: the driver
// then maps the range in pieces no larger than a chunk, which is what the shade stage's forty streams want).  holes: a spacer chunk is
// created behind every chunk and released at the end (the first form of the experiment; not what helps).  Every failure undoes what was
// done and reports it; the caller falls back to hipMalloc.
static hipError_t alloc_spread(DevBuf& b, size_t bytes, size_t chunk, int device, bool holes, int fail_at = -1) {      // fail_at: test option spread_fail_at -- chunk number whose creation is made to fail
  hipMemAllocationProp prop; std::memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
  size_t gran = 0;
  hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
  if (e != hipSuccess || gran == 0) return e != hipSuccess ? e : hipErrorNotSupported;
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t n = (bytes + chunk - 1) / chunk;
  void* va = nullptr;
  e = hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0);
  if (e != hipSuccess) return e;
  std::vector<hipMemGenericAllocationHandle_t> got, spacers;
  size_t mapped = 0;
  auto undo = [&]() {
    for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((char*)va + i * chunk, chunk);
    for (auto h : got) (void)hipMemRelease(h);
    for (auto h : spacers) (void)hipMemRelease(h);
    (void)hipMemAddressFree(va, n * chunk);
    (void)hipGetLastError();
  };
  for (size_t i = 0; i < n; ++i) {
    hipMemGenericAllocationHandle_t h;
    e = ((int64_t)i == (int64_t)fail_at) ? hipErrorOutOfMemory : hipMemCreate(&h, chunk, &prop, 0);
    if (e != hipSuccess) { undo(); return e; }
    got.push_back(h);
    hipMemGenericAllocationHandle_t sp;                  // the hole behind it (none if the device is too full: the layout degrades, nothing fails)
    if (holes && i + 1 < n) { if (hipMemCreate(&sp, chunk, &prop, 0) == hipSuccess) spacers.push_back(sp); else (void)hipGetLastError(); }
  }
  for (size_t i = 0; i < n; ++i) {
    e = hipMemMap((char*)va + i * chunk, chunk, 0, got[i], 0);
    if (e != hipSuccess) { undo(); return e; }
    mapped = i + 1;
  }
  hipMemAccessDesc acc; std::memset(&acc, 0, sizeof acc);
  acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(va, n * chunk, &acc, 1);
  if (e != hipSuccess) { undo(); return e; }
  for (auto h : spacers) (void)hipMemRelease(h);
  b.p = va; b.bytes = bytes; b.chunks = std::move(got); b.chunk_bytes = chunk; b.reserved = n * chunk;
  return hipSuccess;
}

int ensure_device() {
  Ctx& c = g_ctx;
  if (c.device_ready) return 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail("no HIP device available (this library has no CPU path): " + std::string(hipGetErrorString(e)));
  if (c.device >= 0) HIP_TRY(hipSetDevice(c.device));
  HIP_TRY(hipGetDevice(&c.device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c.device));
  c.num_cus = prop.multiProcessorCount;
  c.arch = prop.gcnArchName;
  // LDS per CU: 160 KB on gfx950 (MI355X_MICROARCH.md); the runtime's own figure where it is larger than the per-workgroup 64 KB
  c.lds_per_cu = (c.arch.rfind("gfx950", 0) == 0) ? 160 * 1024 : std::max<int>(64 * 1024, (int)prop.maxSharedMemoryPerMultiProcessor);
  c.arch = prop.gcnArchName;
  HIP_TRY(hipMalloc(&c.d_cursor, kCursorInts * sizeof(int)));   // [0] work cursor, [1] live-ray queue length, [2] overflow queue length, [32*k] cursor of queue segment k
  HIP_TRY(hipMalloc(&c.d_counters, 16 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(c.d_counters, 0, 16 * sizeof(unsigned long long)));
  c.device_ready = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------
static int upload_scene_arrays(HostScene& hs) {
  Ctx& c = g_ctx;
  if (upload(c.b_spheres, hs.spheres) || upload(c.b_sphere_mat, hs.sphere_mat) || upload(c.b_lights, hs.lights) ||
      upload(c.b_materials, hs.materials) || upload(c.b_bf_pos, hs.bf_pos) || upload(c.b_bf_nrm, hs.bf_nrm) ||
      upload(c.b_bf_uv, hs.bf_uv) || upload(c.b_bf_idx, hs.bf_idx) || (!hs.gpu_built && upload(c.b_nodes, hs.bvh.nodes)) || (!hs.gpu_built && upload(c.b_qnodes, hs.bvh.qnodes)) ||
      (!hs.gpu_built && upload(c.b_tris, hs.bvh.tris)) || upload(c.b_m_shade, hs.m_shade))
    return 1;
  if (hs.hdr.n_inst > 0 && (upload(c.b_inst, hs.inst) || upload(c.b_qnodes, hs.two.qnodes) || upload(c.b_tlas_nodes, hs.two.tlas.nodes) || upload(c.b_tlas_tris, hs.two.tlas.tris) ||
                            upload(c.b_blas_nodes, hs.two.blas_nodes) || upload(c.b_blas_tris, hs.two.blas_tris)))
    return 1;
  DevScene& s = c.scene;
  s = hs.hdr;
  if (hs.hdr.n_inst > 0) {
    s.inst = (const DevInstance*)c.b_inst.p; s.tlas_nodes = (const float*)c.b_tlas_nodes.p; s.tlas_tris = (const float*)c.b_tlas_tris.p;
    s.blas_nodes = (const float*)c.b_blas_nodes.p; s.blas_tris = (const float*)c.b_blas_tris.p;
  }
  s.spheres = (const DevSphere*)c.b_spheres.p; s.sphere_mat = (const int32_t*)c.b_sphere_mat.p;
  s.lights = (const DevLight*)c.b_lights.p; s.materials = (const DevMaterial*)c.b_materials.p;
  s.bf_pos = (const float*)c.b_bf_pos.p; s.bf_nrm = (const float*)c.b_bf_nrm.p; s.bf_uv = (const float*)c.b_bf_uv.p; s.bf_idx = (const int32_t*)c.b_bf_idx.p;
  s.nodes = (const float*)c.b_nodes.p; s.tris = (const float*)c.b_tris.p;
  s.m_shade = (const float*)c.b_m_shade.p;
  if (!c.d_scene) HIP_TRY(hipMalloc(&c.d_scene, sizeof(DevScene)));
  HIP_TRY(hipMemcpy(c.d_scene, &s, sizeof(DevScene), hipMemcpyHostToDevice));
  return 0;
}

int upload_scene(const ArtSceneDesc* d) {
  if (!d) return fail("art_upload_scene: null scene");
  std::string err;
  HostScene hs;
  if (!flatten_scene(*d, g_devs[0].bvh_params, hs, err)) return fail(err);      // validation + host BVH build: once, no GPU needed
  const std::vector<float> tri9 = std::move(hs.deferred_tri9);                    // option bvh_builder >= 1: every device builds its own tree
  hs.deferred_tri9.clear();
  Dev0Guard guard;
  for (int k = 0; k < g_ndev; ++k) {                                              // scene + BVH replicated on every GPU
    if (use_dev(k)) return 1;
    Ctx& c = g_ctx;
    c.bvh_params = g_devs[0].bvh_params;
    if (ensure_device()) return 1;
    int dev_stack = hs.bvh.max_stack;
    if (!tri9.empty()) {
      DevBuf t9;
      if (upload(t9, tri9)) return 1;
      GpuBvh g;
      const bool ok = build_bvh8_gpu((const float*)t9.p, (int)(tri9.size() / 9), c.bvh_params, c.stream, g, err);
      t9.release();
      if (!ok) { if (g.nodes) (void)hipFree(g.nodes); if (g.tris) (void)hipFree(g.tris); if (g.qnodes) (void)hipFree(g.qnodes); return fail("GPU BVH build: " + err); }
      c.b_nodes.release(); c.b_tris.release(); c.b_qnodes.release();
      c.b_qnodes.p = g.qnodes; c.b_qnodes.bytes = g.qnodes ? (size_t)g.n_nodes * kQNodeBytes : 0;
      c.b_nodes.p = g.nodes; c.b_nodes.bytes = (size_t)g.n_nodes * node_floats(c.bvh_params.width) * 4;
      c.b_tris.p = g.tris; c.b_tris.bytes = (size_t)g.n_tris * kTriFloats * 4;
      dev_stack = g.max_stack;
      if (k == 0) {
        hs.bvh.width = c.bvh_params.width;
        hs.bvh.n_nodes = g.n_nodes; hs.bvh.n_tris = g.n_tris; hs.bvh.max_stack = g.max_stack;
        hs.bvh_build_ms = g.build_ms; hs.gpu_built = true;
      }
      hs.hdr.n_nodes = g.n_nodes; hs.hdr.n_tris = g.n_tris;                       // node numbering may differ between devices (atomics), sizes do not
    }
    // the trace kernel addresses nodes and triangles with 32-bit byte offsets; the 4-wide entry word keeps bit 31 for the leaf flag
    const uint64_t off_limit = (hs.hdr.node_width == 4) ? (1ull << 31) : (1ull << 32);
    if (hs.hdr.n_inst == 0 && ((uint64_t)hs.hdr.n_nodes * node_floats(hs.hdr.node_width) * 4 >= (1ull << 32) || (uint64_t)hs.hdr.n_tris * (hs.hdr.node_width == 4 ? kQTriBytes : kTriBytes) >= off_limit)
        )  return fail("mesh too large: the trace kernel addresses nodes and triangles with 32-bit byte offsets (max ~33M triangles at width 4, ~89M at width 8)");
    if (hs.hdr.node_width == 4 && hs.hdr.n_nodes > 0 && hs.gpu_built && !c.b_qnodes.p) return fail("internal: GPU build returned no quantised nodes");
    if (dev_stack > kStackEntries) return fail("BVH traversal stack bound " + std::to_string(dev_stack) + " exceeds " + std::to_string(kStackEntries));
    if (upload_scene_arrays(hs)) return 1;
    c.b_qtris.release();
    if (hs.hdr.node_width == 4 && hs.hdr.n_tris > 0 && hs.hdr.n_inst == 0) {       // 64-byte padded copy of the triangle records for the 4-wide kernel
      if (ensure(c.b_qtris, (size_t)hs.hdr.n_tris * kQTriBytes)) return 1;
      launch_pad_tris(c.stream, (const float*)c.b_tris.p, (float*)c.b_qtris.p, hs.hdr.n_tris);
      HIP_TRY(hipStreamSynchronize(c.stream));
    } else if (hs.hdr.n_inst > 0) {                      // instanced scene: the meshes' object-space records, padded the same way
      const int nrec = (int)(hs.two.blas_tris.size() / kTriFloats);
      if (ensure(c.b_qtris, (size_t)nrec * kQTriBytes)) return 1;
      launch_pad_tris(c.stream, (const float*)c.b_blas_tris.p, (float*)c.b_qtris.p, nrec);
      HIP_TRY(hipStreamSynchronize(c.stream));
    }
    c.bvh_stack_bound = std::max(8, dev_stack);
    c.blocks_per_cu = 0;   // re-query occupancy
    c.scene_ready = true;
    c.auto_phase = 0; c.auto_redo = 0; c.auto_gen += 1;  // a new scene: the shade stage measures its items-per-thread choice again
  }
  g_devs[0].host_scene = std::move(hs);
  return 0;
}

// ------------------------------------------------------------------------------------------------
static int build_shard() {
  Ctx& c = g_ctx;
  const std::vector<uint32_t> pm = build_pixmap(c.width, c.height, c.rank, c.nranks, c.tile);
  c.npix_local = (int)pm.size();
  return upload(c.b_pixmap, pm);
}

static int resize_one(int w, int h);
int resize(int w, int h) {
  if (w <= 0 || h <= 0 || (int64_t)w * h > (1ll << 30)) return fail("art_resize: bad size");
  Dev0Guard guard;
  for (int k = 0; k < g_ndev; ++k) { if (use_dev(k) || resize_one(w, h)) return 1; }
  for (int k = 0; k < g_ndev; ++k) { Ctx& c = g_devs[k]; c.busy_ms = c.idle_ms = c.start_skew_ms = 0.0; for (Ctx::PassClock* pc : c.pass_clock) delete pc; c.pass_clock.clear(); }
  if (g_devs[0].device_ready && !use_dev(0)) reset_reduce_info();
  return 0;
}
static int resize_one(int w, int h) {
  Ctx& c = g_ctx;
  if (ensure_device()) return 1;
  if (!c.pass_clock.empty()) HIP_TRY(hipStreamSynchronize(c.stream));      // (host callbacks of earlier passes still write into their cells)
  c.width = w; c.height = h;
  const size_t n = (size_t)w * h;
  if (ensure(c.b_accum, n * 12) || ensure(c.b_screen, n * 4)) return 1;
  float* acc = c.ext_accum ? c.ext_accum : (float*)c.b_accum.p;
  HIP_TRY(hipMemsetAsync(acc, 0, n * 12, c.stream));
  c.spp = 0;
  c.stats = ArtStats();
  c.stage = ArtStageStats();
  c.auto_phase = 0; c.auto_redo = 0; c.auto_gen += 1;    // (the batch size follows the frame: measure again)
  c.lost_reported = 0;                                   // (the device counters are zeroed below)
  if (c.d_items) HIP_TRY(hipMemsetAsync(c.d_items, 0, 32 * sizeof(unsigned long long), c.stream));
  c.camera_rays = 0;
  HIP_TRY(hipMemsetAsync(c.d_counters, 0, 16 * sizeof(unsigned long long), c.stream));
  return build_shard();
}

static float* accum_ptr() { return g_ctx.ext_accum ? g_ctx.ext_accum : (float*)g_ctx.b_accum.p; }
static int download_from(const float* acc_dev, float* accum_host, uint32_t* screen_host, int layout, int spp);

// Path arrays for P slots and `depth` fold levels, carved out of one allocation.  Hot state exists in TWO banks: every bounce reads one
// and writes the survivors densely into the other (k_shade_compact); cold state (fold stack, terminal value, per-sample radiance, final
// flags) is indexed by slot.  Hot state per item, record layout (round 3, the cooperative schedule): the trace records of its two rays
// (2 x 64 B; the next stage reads the extension ray from there), their hit records (2 x 16 B), prev pdf, flags, shadow epsilon, pending
// explicit colour, item -> slot map, and the extension ray once more as six SoA words for the next stage: 21 words per bank + 32 words of
// records in one array shared by the banks.  Plain layout (one-ray-per-lane schedule, debug pass): rays as SoA arrays, 29 words.
constexpr size_t kRecSlack = 4096;       // records past the last one the trace kernel's chunk prefetch may touch
// (record schedule: 6 ray words, ONE 16-byte hit record + the shadow ray's result word, 7 per-path words)
static size_t hot_stride(size_t P) { return ((P + 63) & ~(size_t)63) + (size_t)g_ctx.hot_pad; }      // (+ option hot_pad: items between the fields of a bank's block, Ctx::hot_pad)
static size_t hot_floats(size_t P, bool rec) { return rec ? (size_t)kHotFields * hot_stride(P) + 64 : (14 + 8 + 7) * P; }
// (the trace records are not double-banked: a bank's records are dead once its rays are traced, and the next stage reads none of them)
// cold state: e (depth + 1 levels: dense fold records keep e_k at level k + 1) and w (depth levels) x 3, child (depth levels), term, rad, final flags
// (+ 4 P words: the queue of the items k_shade_compact defers to its heavy-material instantiation, 16 B each, worst case every item)
static size_t path_floats(size_t P, int depth, bool rec) { return 2 * hot_floats(P, rec) + (rec ? 32 * P + kRecSlack * 16 + 16 + 4 * P + 4 : 0) + (7 * (size_t)depth + 3 + 3 + 3 + 1) * P + 64; }

static int ensure_paths(size_t P, int depth, bool rec) { return ensure(g_ctx.b_paths, path_floats(P, depth, rec) * 4 + 256); }

// q[0], q[1]: the two banks (slot_id = their own map); both share the cold arrays.  An identity-layout user takes q[0] with slot_id = nullptr
// and final_flags = flags.
static void carve(DevPaths q[2], int P, int depth, bool rec, uint4** heavy = nullptr) {
  float* const f0 = (float*)g_ctx.b_paths.p;
  float* f = f0; const size_t p = (size_t)P;
  auto take = [&](size_t n) { float* r = f; f += n; return r; };
  auto align = [&](size_t floats) { f += (floats - ((size_t)(f - f0) & (floats - 1))) & (floats - 1); };
  Rec4* records = nullptr;
  if (rec) { align(16); records = (Rec4*)take(32 * p + kRecSlack * 16); }       // 64-byte records, ONE array for both banks
  if (rec) { uint4* const h = (uint4*)take(4 * p); if (heavy) *heavy = h; }     // (still 16-byte aligned)
  for (int k = 0; k < 2; ++k) {
    DevPaths& b = q[k];
    if (rec) {
      // the bank as ONE block (art_scene.h HotField); the pointer fields name its pieces for the kernels that take them one by one
      b.rec = records;
      align(64);
      const size_t st = hot_stride(p);
      float* const h = take((size_t)kHotFields * st);
      b.hot = h; b.stride = (int32_t)st;
      b.ray_ox = h + HF_OX * st; b.ray_oy = h + HF_OY * st; b.ray_oz = h + HF_OZ * st; b.ray_dx = h + HF_DX * st; b.ray_dy = h + HF_DY * st; b.ray_dz = h + HF_DZ * st;
      b.ray_tfar = nullptr;
      b.hit = (DevHit*)(h + HF_HIT * st); b.sh_t = h + HF_SHT * st;
      b.prev_pdf = h + HF_PDF * st; b.flags = (uint32_t*)(h + HF_FLAGS * st); b.sh_min_t = h + HF_SHMIN * st; b.slot_id = (const uint32_t*)(h + HF_SLOT * st);
      b.cand_r = b.cand_g = b.cand_b = nullptr;                                 // (dense fold records: the pending colour has no hot words)
      b.shadow_rule = g_ctx.shadow_anyhit ? 1 : 0; b.has_bvh = g_ctx.scene.n_tris > 0 ? 1 : 0;
      continue;
    } else {
      b.hot = nullptr; b.stride = 0;
      b.rec = nullptr; b.rec_mode = REC_NONE;
      b.ray_ox = take(2 * p); b.ray_oy = take(2 * p); b.ray_oz = take(2 * p);
      b.ray_dx = take(2 * p); b.ray_dy = take(2 * p); b.ray_dz = take(2 * p); b.ray_tfar = take(2 * p);
    }
    align(4);                                                                   // 16-byte hit records
    b.hit = (DevHit*)take(8 * p); b.sh_t = nullptr;
    b.prev_pdf = take(p); b.flags = (uint32_t*)take(p); b.sh_min_t = take(p);
    b.cand_r = take(p); b.cand_g = take(p); b.cand_b = take(p);
    b.slot_id = (const uint32_t*)take(p);
    b.shadow_rule = g_ctx.shadow_anyhit ? 1 : 0; b.has_bvh = g_ctx.scene.n_tris > 0 ? 1 : 0;
  }
  DevPaths& a = q[0];
  a.e_r = take((depth + 1) * p); a.e_g = take((depth + 1) * p); a.e_b = take((depth + 1) * p);
  a.w_r = take(depth * p); a.w_g = take(depth * p); a.w_b = take(depth * p);
  a.child = (int32_t*)take(depth * p);
  a.cold = rec ? a.e_r : nullptr; a.depth = depth;       // (the seven takes above are consecutive: ONE block, art_scene.h DevPaths::cold)
  a.fold_dense = rec ? 1 : 0;                   // the compacted (record) schedule keeps dense fold records; the plain one folds by slot
  a.synth0 = rec ? 1 : 0;                       // ... and lets bounce 0 recompute the camera ray instead of reading it back (nothing else reads raygen's bank)
  a.term_r = take(p); a.term_g = take(p); a.term_b = take(p);
  a.rad_r = take(p); a.rad_g = take(p); a.rad_b = take(p);
  a.final_flags = (uint32_t*)take(p);
  DevPaths& c = q[1];
  c.e_r = a.e_r; c.e_g = a.e_g; c.e_b = a.e_b; c.w_r = a.w_r; c.w_g = a.w_g; c.w_b = a.w_b;
  c.term_r = a.term_r; c.term_g = a.term_g; c.term_b = a.term_b; c.rad_r = a.rad_r; c.rad_g = a.rad_g; c.rad_b = a.rad_b;
  c.final_flags = a.final_flags; c.child = a.child; c.fold_dense = a.fold_dense; c.synth0 = a.synth0; c.cold = a.cold; c.depth = a.depth;
}

// LDS stack per ray: the tree's worst-case bound if 8 workgroups per CU (8 waves per SIMD) still fit in the CU's 160 KB, else the
// largest size that does; then pushes are checked and the few rays that go deeper are finished by k_trace_overflow.
static void stack_plan(int kernel, int& entries, bool& overflow) {
  Ctx& c = g_ctx;
  const int per_block = c.lds_per_cu / 8, groups = 64 / c.scene.node_width;        // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  (void)kernel;
  int cap = per_block / (4 * groups * 8) - 3;     // entries per ray (+ 2 guard entries + the sink of masked pushes)
  if (c.lds_stack_cap > 0) cap = c.lds_stack_cap;
  cap = std::min(cap, 64 * 1024 / (4 * groups * 8) - 3);      // one workgroup's dynamic LDS stays within the 64 KB a launch may ask for by default
  entries = std::min(c.bvh_stack_bound, cap);
  overflow = c.bvh_stack_bound > entries;
}

static int coop_grid() {
  Ctx& c = g_ctx;
  int entries; bool ovf; stack_plan(c.trace_kernel, entries, ovf);
  if (c.opt_blocks_per_cu > 0) return c.num_cus * c.opt_blocks_per_cu;
  return c.num_cus * trace_coop_blocks_per_cu(entries, c.scene.node_width);
}

static void fill_trace_args(TraceArgs& a, const DevPaths& q, int n_rays) {
  Ctx& c = g_ctx;
  a.n_rays = n_rays; a.width = c.scene.node_width; a.instanced = c.scene.n_inst > 0 ? (c.inst_coop ? 1 : 2) : 0;
  a.inst = c.scene.inst; a.inst_shift = c.scene.inst_shift;
  { int e; bool o; stack_plan(c.trace_kernel, e, o); a.stack_entries = e; a.stack_overflow = o ? 1 : 0; }
  a.node_min = c.node_min ? c.node_min : (c.scene.n_inst > 0 ? 2 : 4); a.refill_min = c.refill_min; a.segments = c.queue_segments; a.chunk = c.ray_chunk;
  a.ray_ox = q.ray_ox; a.ray_oy = q.ray_oy; a.ray_oz = q.ray_oz; a.ray_dx = q.ray_dx; a.ray_dy = q.ray_dy; a.ray_dz = q.ray_dz; a.ray_tfar = q.ray_tfar;
  a.hit = q.hit; a.sh_t = q.sh_t;
  a.nodes = c.scene.nodes; a.qnodes = (const uint32_t*)c.b_qnodes.p; a.tris = c.scene.tris; a.qtris = (const float*)c.b_qtris.p; a.n_tris = c.scene.n_tris;
  a.sh_min = (c.shadow_anyhit && q.sh_min_t && n_rays > q.P) ? q.sh_min_t : nullptr; a.shadow_begin = q.P;
  a.cursor = c.d_cursor; a.stats = c.d_counters + 3; a.live_rays = c.d_counters;
  a.queue = (int*)c.b_queue.p; a.queue_count = c.d_cursor + 1;
  a.rec = (float4*)c.b_queue.p;
  a.ovf_queue = (int*)c.b_ovf.p; a.ovf_count = c.d_cursor + 2;
  a.item_count = nullptr;
  a.queue_fixed = -1; a.queue_items = nullptr; a.queue_mul = 1;
}

// one trace launch, bracketed by HIP events on the launch stream.  records != nullptr: the bank's rays are already trace records in item
// order (DevPaths::rec, written by the stage that emitted them): no k_analytic pass, the queue is the record array itself.
struct RecordQueue { int fixed; const int* items; int mul; };
// event pairs around groups of launches on the launch stream: ev_begin(kind) ... ev_end()
static int ev_begin(int kind, int trial = 0) {      // trial: 1 / 2 = a shade launch of the items-per-thread trial A / B (Ctx::opt_shade_per)
  Ctx& c = g_ctx;
  if (c.ev_pool.size() < c.ev_used + 2) {
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    c.ev_pool.push_back(e0); c.ev_pool.push_back(e1);
  }
  if (c.ev_kind.size() < c.ev_pool.size() / 2) c.ev_kind.resize(c.ev_pool.size() / 2);
  c.ev_kind[c.ev_used / 2] = (uint8_t)(kind | (trial << 4) | (trial ? (c.auto_gen & 3) << 6 : 0));
  HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used], c.stream));
  return 0;
}
static int ev_end() { Ctx& c = g_ctx; HIP_TRY(hipEventRecord(c.ev_pool[c.ev_used + 1], c.stream)); c.ev_used += 2; return 0; }

static int trace(const DevPaths& q, int n_rays, bool timed = true, const int* item_count = nullptr, const RecordQueue* records = nullptr) {
  Ctx& c = g_ctx;
  const bool coop = (c.trace_kernel == TRACE_COOP);
  if ((int64_t)n_rays > (1ll << 28)) return fail("internal: more than 2^28 rays in one trace launch (32-bit byte offsets of the 16-byte hit records)");
  if (records && !coop) return fail("internal: trace records without the cooperative kernel");
  if (coop && !records && ensure(c.b_queue, ((size_t)n_rays + kRecSlack) * kTraceRecBytes)) return 1;      // live-ray queue: one 64-byte trace record per queued ray (+ one chunk of slack for the chunk prefetch)
  TraceArgs a; fill_trace_args(a, q, n_rays);
  a.item_count = item_count;
  if (records) { a.rec = (float4*)q.rec; a.queue_fixed = records->fixed; a.queue_items = records->items; a.queue_mul = records->mul; }
  if (coop && a.stack_overflow && ensure(c.b_ovf, (size_t)n_rays * sizeof(int))) return 1;
  a.ovf_queue = (int*)c.b_ovf.p;
  if (coop) {
    HIP_TRY(hipMemsetAsync(c.d_cursor, 0, kCursorInts * sizeof(int), c.stream));
  }
  if (coop && !records) launch_analytic(c.stream, c.scene, a, c.count_tests);   // outside the trace-kernel event pair
  if (timed && ev_begin(0)) return 1;
  // a workgroup keeps 4 waves x (64 / width) rays in flight: a handful of rays (the legacy per-ray seam) gets a handful of workgroups
  const int rays_per_block = 4 * (64 / std::max(1, c.scene.node_width));
  const int grid = (int)std::min<int64_t>(coop_grid(), ((int64_t)n_rays + rays_per_block - 1) / rays_per_block);
  launch_trace(c.stream, c.d_scene, a, c.trace_kernel, c.count_tests, std::max(1, grid));
  if (timed && ev_end()) return 1;
  HIP_TRY(hipGetLastError());
  return 0;
}

// drain event pairs into stats (requires the stream to be idle)
static int collect_timing() {
  Ctx& c = g_ctx;
  for (size_t i = 0; i + 1 < c.ev_used; i += 2) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, c.ev_pool[i], c.ev_pool[i + 1]));
    const int tagged = (i / 2 < c.ev_kind.size()) ? c.ev_kind[i / 2] : 0;
    const int kind = tagged & 15, trial = (tagged >> 4) & 3, gen = tagged >> 6;
    if (kind == 0) { c.stats.trace_ms += ms; c.stats.trace_launches += 1; }
    else if (kind == 1) { c.stage.shade_ms += ms; c.stage.shade_launches += 1; if ((trial == 1 || trial == 2) && gen == (c.auto_gen & 3)) c.auto_ms[trial - 1] += ms; }
    else if (kind == 2) c.stage.raygen_ms += ms;
    else c.stage.fold_ms += ms;
  }
  c.ev_used = 0;
  if (c.auto_phase == 3) {         // both trial batches are done (the stream is idle): keep the faster setting from here on
    c.auto_per = (c.auto_ms[1] < c.auto_ms[0]) ? 2 : 4;
    c.auto_phase = 4;
    if (g_debug_addr) std::fprintf(stderr, "ART_DEBUG_ADDR shade_per trial: 4 items per thread %.3f ms, 2 items per thread %.3f ms per batch of %lld paths -> %d\n", c.auto_ms[0], c.auto_ms[1], (long long)c.auto_P[0], c.auto_per);
  }
  if (c.d_items) {
    unsigned long long it[32];
    HIP_TRY(hipMemcpy(it, c.d_items, sizeof it, hipMemcpyDeviceToHost));
    for (int k = 0; k < 16; ++k) { c.stage.items_in[k] = it[k]; c.stage.items_out[k] = it[16 + k]; }
  }
  unsigned long long cnt[16];
  HIP_TRY(hipMemcpy(cnt, c.d_counters, sizeof cnt, hipMemcpyDeviceToHost));
  c.stats.node_phase_iters = cnt[8]; c.stats.leaf_phase_iters = cnt[9]; c.stats.wave_iters = cnt[10];
  c.stats.rays = cnt[0];
  c.stats.lost_paths = cnt[15];
  c.stats.box_tests = cnt[3]; c.stats.tri_tests = cnt[4]; c.stats.node_visits = cnt[5]; c.stats.leaf_visits = cnt[6]; c.stats.traced_rays = cnt[7];
  return 0;
}

static int check_pass(const ArtPassParams* p) {
  Ctx& c = g_ctx;
  if (!p) return fail("null ArtPassParams");
  if (!c.scene_ready) return fail("no scene uploaded (art_upload_scene)");
  if (c.width <= 0) return fail("no viewport (art_resize)");
  if (p->max_depth < 1 || p->max_depth > 16) return fail("max_depth must be 1..16");
  if (p->vthreads < 1) return fail("vthreads must be >= 1");
  if (p->layout != ART_LAYOUT_ADA_XY && p->layout != ART_LAYOUT_ROW_MAJOR) return fail("unknown layout");
  return 0;
}

static void make_frame(const ArtPassParams* p, DevFrame& f) {
  Ctx& c = g_ctx;
  f.width = c.width; f.height = c.height;
  f.render_type = p->render_type; f.aa_on = p->aa_on ? 1 : 0; f.max_depth = p->max_depth;
  f.seed_lo = (uint32_t)p->seed; f.seed_hi = (uint32_t)(p->seed >> 32);
  std::memcpy(f.background, p->background, 12);
  const float fov = kHalfPi;                                   // ray_tracer.adb:63  Pi/2.0
  f.cam_z = -(float)c.width / safe_tan(fov / 2.0f);            // ray_tracer.adb:67
  f.skip_null_shadow = c.skip_null_shadow ? 1 : 0;
}

static int render_pass_one(const ArtPassParams* p, int32_t* spp_inout);
// One Render_Pass on every device of the process: each GPU renders the pixel tiles it owns into its own accum buffer, asynchronously on
// its own stream (the host only enqueues: ~100 calls per device and pass).  The buffers meet in reduce_accum() when somebody asks
// for the image.
int render_pass_device(const ArtPassParams* p, int32_t* spp_inout) {
  Dev0Guard guard;
  int32_t spp_out = spp_inout ? *spp_inout : 0;
  for (int k = 0; k < g_ndev; ++k) {
    int32_t spp_k = spp_inout ? *spp_inout : 0;
    if (use_dev(k) || render_pass_one(p, spp_inout ? &spp_k : nullptr)) return 1;
    if (k == 0) spp_out = spp_k;
  }
  if (spp_inout) *spp_inout = spp_out;
  return 0;
}
static int render_pass_one(const ArtPassParams* p, int32_t* spp_inout) {
  Ctx& c = g_ctx;
  if (check_pass(p)) return 1;
  if (p->render_type == ART_RT_DEBUG || p->render_type == ART_RT_WHITTED) return fail("debug render types go through art_debug_hit_pass");
  if (p->render_type < ART_PT_STUPID || p->render_type > ART_PT_MIS) return fail("unknown render_type");
  if (c.scene.n_inst > 0 && c.trace_kernel != TRACE_COOP) return fail("an instanced scene renders through the record schedule only (option trace_kernel = 0)");
  const int per = p->aa_on ? 4 : 1;
  if (spp_inout) c.spp = *spp_inout;
  if (p->aa_on && (c.spp % 4) != 0) return fail("with anti-aliasing on, spp must be a multiple of 4 (Generate4RayDirections order)");
  const int S = p->vthreads * per;           // samples this pass
  const int npix = c.npix_local;
  DevFrame F; make_frame(p, F);
  // batch = pixel chunk x sample chunk with pc * sc <= batch_paths.  The result does not depend on the batching (the RNG is keyed by
  // pixel, sample and bounce), so when HBM is short (a shared GPU, a caller holding memory) the batch is halved until it fits.
  int pc = 0, sc = 0;
  if (npix > 0) {
    int64_t cap = std::max<int64_t>(c.batch_paths, per);
    // Two buffers belong to a batch: the path state and the live-ray queue (one 64-byte trace record for each of the up to 2 rays of a path).
    auto try_alloc = [&](DevBuf& b, size_t bytes, hipError_t& e) {          // true: b holds at least `bytes`
      e = hipSuccess;
      if (b.p && b.bytes >= bytes) return true;
      b.release();
      // physically contiguous if the driver can (option paths_contiguous; art_api_internal.h Ctx::paths_contiguous says why), else as it comes
      e = hipErrorOutOfMemory; c.paths_are_contiguous = false; c.paths_are_spread = false;
      // the default (-1): 64 MB chunks for a path state of a gigabyte or more -- where the mapping granularity decides the stage's rate
      const int chunk_mb = c.paths_spread_mb > 0 ? c.paths_spread_mb : (c.paths_spread_mb < 0 && !c.paths_contiguous && bytes >= ((size_t)1 << 30)) ? 64 : 0;
      if (chunk_mb > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        e = alloc_spread(b, bytes, (size_t)chunk_mb << 20, c.device, c.paths_spread_holes, c.spread_fail_at);
        if (g_debug_addr) std::fprintf(stderr, "ART_DEBUG_ADDR alloc_spread %.2f GB in chunks of %d MB: %s, %.1f ms\n", (double)bytes / 1e9, chunk_mb, hipGetErrorString(e),
                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (e == hipSuccess) { c.paths_are_spread = true; return true; }
        b.p = nullptr; (void)hipGetLastError();
      }
      if (c.paths_contiguous) {
        e = hipExtMallocWithFlags(&b.p, bytes, hipDeviceMallocContiguous);
        if (e == hipSuccess) c.paths_are_contiguous = true; else { b.p = nullptr; (void)hipGetLastError(); }
      }
      if (e != hipSuccess) e = hipMalloc(&b.p, bytes);
      if (e == hipSuccess) { b.bytes = bytes; return true; }
      b.p = nullptr; (void)hipGetLastError();
      return false;
    };
    const bool rec_layout = (c.trace_kernel == TRACE_COOP);              // the cooperative schedule keeps its rays as trace records inside the path state
    for (;;) {
      pc = (int)std::min<int64_t>(npix, std::max<int64_t>(1, cap / per));
      sc = (int)std::min<int64_t>(S, std::max<int64_t>(per, (cap / pc) / per * per));
      hipError_t e;
      if (try_alloc(g_ctx.b_paths, path_floats((size_t)pc * sc, p->max_depth, rec_layout) * 4 + 256, e)) {
        if (g_debug_live) std::fprintf(stderr, "path state: %d pixels x %d samples per batch, %.2f GB\n", pc, sc, (double)g_ctx.b_paths.bytes / 1e9);
        if (g_debug_addr) {
          DevPaths bk[2]; std::memset(bk, 0, sizeof bk);
          carve(bk, pc * sc, p->max_depth, rec_layout);
          std::fprintf(stderr, "ART_DEBUG_ADDR spread %d contiguous %d paths %p bytes %zu P %d rec %p hot0 %p hot1 %p stride %d cold %p live %p counters %p cursor %p\n", c.paths_are_spread ? 1 : 0, c.paths_are_contiguous ? 1 : 0, g_ctx.b_paths.p, g_ctx.b_paths.bytes, pc * sc,
                       (void*)bk[0].rec, (void*)bk[0].hot, (void*)bk[1].hot, bk[0].stride, (void*)bk[0].cold, (void*)c.d_live, (void*)c.d_counters, (void*)c.d_cursor);
        }
        break;
      }
      if (g_debug_live) std::fprintf(stderr, "path state: %.2f GB refused (%s)\n", (double)(path_floats((size_t)pc * sc, p->max_depth, rec_layout) * 4 + 256) / 1e9, hipGetErrorString(e));
      if (e != hipErrorOutOfMemory || cap <= 65536) return fail(std::string("path buffers: ") + hipGetErrorString(e));
      cap /= 2;
    }
  }
  struct PassEvents {             // destroyed on every error return; handed to the context on success
    hipEvent_t a = nullptr, b = nullptr;
    ~PassEvents() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
  } ev;
  HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b));
  HIP_TRY(hipEventRecord(ev.a, c.stream));
  Ctx::PassClock* clock = nullptr;
  if (g_multi_api) { clock = new Ctx::PassClock; c.pass_clock.push_back(clock); HIP_TRY(hipLaunchHostFunc(c.stream, clock_cb, &clock->t0)); }
  if (npix > 0) {
    for (int px0 = 0; px0 < npix; px0 += pc) {
      const int pn = std::min(pc, npix - px0);
      for (int s0 = 0; s0 < S; s0 += sc) {
        const int sn = std::min(sc, S - s0);
        DevPaths bank[2]; std::memset(bank, 0, sizeof bank);
        for (DevPaths& b : bank) { b.P = pn * sn; b.npix = pn; b.pixmap = (const uint32_t*)c.b_pixmap.p + px0; b.sample_base = (uint32_t)(c.spp + s0); }
        uint4* heavy = nullptr;
        carve(bank, bank[0].P, p->max_depth, c.trace_kernel == TRACE_COOP, &heavy);
        // items per thread of the shade stage for this batch: the option, the measured choice, or a trial (art_api_internal.h Ctx::opt_shade_per).
        // Nothing here waits for the GPU: a trial batch only tags its shade launches' event pairs, and the next call that synchronises
        // anyway reads them (collect_timing) -- Render_Pass releases all its workers before it waits for any (ray_tracer.adb:271-277).
        int trial = 0, shade_per = c.opt_shade_per ? c.opt_shade_per : (c.auto_phase >= 4 ? c.auto_per : 4);
        if (c.opt_shade_per == 0 && c.auto_phase < 3 && c.trace_kernel == TRACE_COOP && !c.shade_split) {
          const int64_t Pb = (int64_t)pn * sn;
          if (c.auto_phase == 0) c.auto_phase = 1;                                      // the warm batch: 4 items per thread, not measured
          else if (c.auto_phase == 1) { trial = 1; c.auto_ms[0] = 0.0; c.auto_P[0] = Pb; c.auto_phase = 2; }
          else if (Pb == c.auto_P[0]) { trial = 2; shade_per = 2; c.auto_ms[1] = 0.0; c.auto_P[1] = Pb; c.auto_phase = 3; }
          else if (++c.auto_redo > 3) { c.auto_per = 4; c.auto_phase = 4; }             // batch sizes keep changing: no trial, 4 items per thread
          else { trial = 1; c.auto_gen += 1; c.auto_ms[0] = 0.0; c.auto_P[0] = Pb; }    // a batch of another size: trial A again, on this size (new generation: the old trial's events no longer count)
        }
        c.camera_rays += (uint64_t)bank[0].P;
        if (c.trace_kernel == TRACE_COOP) {
          // Compacted work sets: raygen fills bank 0 (one item per slot); stage b shades the items of bank b & 1 and writes the survivors
          // densely into the other bank.  d_live[0] / d_live[32]: the banks' item counts.  Every stage leaves its rays as trace records in
          // its output bank (round 3), at positions given by the item index; the trace kernel reads them from there in item order.
          if (!c.d_live) HIP_TRY(hipMalloc(&c.d_live, 32 * 18 * sizeof(int)));          // d_live[32 k]: items of level k (the input set of bounce k), k = 1 .. max_depth <= 16
          unsigned long long* const rays_b = c.count_tests ? c.d_counters + 7 : nullptr;          // the counting pass's ray count (stats[4])
          bank[0].rec_mode = REC_EXT;
          DevPaths q = bank[0];                            // identity layout for raygen
          q.slot_id = nullptr;
          if (!c.d_items) { HIP_TRY(hipMalloc(&c.d_items, 32 * sizeof(unsigned long long))); HIP_TRY(hipMemsetAsync(c.d_items, 0, 32 * sizeof(unsigned long long), c.stream)); }
          if (ev_begin(2)) return 1;
          launch_raygen(c.stream, F, c.scene, q);
          if (ev_end()) return 1;
          launch_bump(c.stream, c.d_counters, rays_b, (unsigned long long)q.P);
          { const RecordQueue rq = {q.P, nullptr, 1}; if (trace(q, q.P, true, nullptr, &rq)) return 1; }
          for (int b = 0; b < p->max_depth; ++b) {
            const int in = b & 1, out = in ^ 1;
            const DevPaths& qi = (b == 0) ? q : bank[in];
            const bool last = b + 1 >= p->max_depth;
            bank[out].rec_mode = (p->render_type == ART_PT_STUPID) ? REC_EXT : (last ? REC_SHADOW : REC_BOTH);
            int* const n_in = c.d_live + 32 * b; int* const n_out = c.d_live + 32 * (b + 1);      // per level: the fold walks them again
            HIP_TRY(hipMemsetAsync(n_out, 0, 2 * sizeof(int), c.stream));         // n_out[1]: the items this stage defers to its heavy-material kernel
            if (ev_begin(1, trial)) return 1;
            launch_shade_compact(c.stream, F, c.scene, qi, bank[out], b, b == 0 ? nullptr : n_in, n_out,
                                 const_cast<uint32_t*>(bank[out].slot_id), c.d_counters + 15, c.d_counters, rays_b, c.shade_split ? heavy : nullptr, n_out + 1, shade_per);
            if (ev_end()) return 1;
            if (b == 0 && c.inject_lost) { launch_bump(c.stream, c.d_counters + 15, nullptr, 1ull); c.inject_lost = 0; }      // test option: what a stage does when it loses a path
            if (g_debug_live) {
              int n = -1; unsigned long long r0 = 0;
              (void)hipStreamSynchronize(c.stream); (void)hipMemcpy(&n, n_out, 4, hipMemcpyDeviceToHost); (void)hipMemcpy(&r0, c.d_counters, 8, hipMemcpyDeviceToHost);
              std::fprintf(stderr, "stage %d: items out %d of %d, rays so far %llu\n", b, n, q.P, r0);
            }
            if (!last || p->render_type != ART_PT_STUPID) {
              const RecordQueue rq = {-1, n_out, bank[out].rec_mode == REC_BOTH ? 2 : 1};
              if (trace(bank[out], 2 * q.P, true, n_out, &rq)) return 1;
            }
          }
          const int last = p->max_depth & 1;              // the bank the last stage wrote
          if (ev_begin(3)) return 1;
          launch_resolve_last(c.stream, bank[last], c.d_live + 32 * p->max_depth, p->max_depth - 1);
          if (bank[last].fold_dense) launch_fold_levels(c.stream, F, bank[last], p->max_depth, c.d_live);
          else launch_fold(c.stream, F, bank[last]);
          launch_accumulate(c.stream, F, q, sn, accum_ptr());
          if (ev_end()) return 1;
          launch_acc_items(c.stream, c.d_live, p->max_depth, q.P, c.d_items);
          c.stage.batches += 1;
        } else {                                          // one-ray-per-lane cross-check kernel: the plain schedule over all slots, in place
          DevPaths q = bank[0];
          q.slot_id = nullptr;
          q.final_flags = q.flags;
          launch_raygen(c.stream, F, c.scene, q);
          for (int b = 0; b < p->max_depth; ++b) {
            if (trace(q, b == 0 ? q.P : 2 * q.P)) return 1;
            launch_shade(c.stream, F, c.scene, q, b);
          }
          if (p->render_type != ART_PT_STUPID) { if (trace(q, 2 * q.P)) return 1; }
          launch_finish(c.stream, F, q, p->max_depth - 1);
          launch_accumulate(c.stream, F, q, sn, accum_ptr());
        }
        HIP_TRY(hipGetLastError());
      }
    }
  }
  HIP_TRY(hipEventRecord(ev.b, c.stream));
  if (clock) HIP_TRY(hipLaunchHostFunc(c.stream, clock_cb, &clock->t1));
  const hipEvent_t p0 = ev.a, p1 = ev.b; ev.a = ev.b = nullptr;
  c.pass_events.push_back(p0); c.pass_events.push_back(p1);
  c.spp += S;
  c.stats.samples += (uint64_t)npix * S;
  if (spp_inout) *spp_inout = c.spp;
  return 0;
}

static int synchronize_one();
int synchronize() {
  Dev0Guard guard;
  int rc = 0;
  for (int k = 0; k < g_ndev; ++k) { if (use_dev(k) || synchronize_one()) rc = 1; }      // (every device is waited for and checked, also after a failure on one)
  if (g_devs[0].device_ready && !use_dev(0)) fold_reduce_events();
  return rc;
}
static int synchronize_one() {
  Ctx& c = g_ctx;
  if (!c.device_ready) return 0;
  HIP_TRY(hipStreamSynchronize(c.stream));
  for (size_t i = 0; i + 1 < c.pass_events.size(); i += 2) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, c.pass_events[i], c.pass_events[i + 1]));
    c.stats.pass_ms += ms;
    (void)hipEventDestroy(c.pass_events[i]); (void)hipEventDestroy(c.pass_events[i + 1]);
  }
  c.pass_events.clear();
  if (collect_timing()) return 1;
  // a lost path is a wrong picture, not a statistic -- a stage that found a path without an output item, a ray its bank has no
  // record for, a material its instantiation was not compiled for, or a staged trace record that did not name the hit slot its position
  // implies (art_shade.h emit_ray) fails the call that waits for the render
  // (the baseline is lost_reported, which only this function advances -- collect_timing may run any number of times between
  // two synchronises without absorbing a loss)
  if (c.stats.lost_paths != c.lost_reported) {
    const uint64_t n = c.stats.lost_paths - c.lost_reported;
    c.lost_reported = c.stats.lost_paths;
    return fail("render self-check: " + std::to_string(n) + " path(s) lost by the wavefront stages (ArtStats::lost_paths); the image is not valid");
  }
  return 0;
}

// every pixel has one owner, so the other devices hold exact zeros there and the sum over devices is exact: ONE
// ncclReduce(sum, float32, W*H*3, root = device 0) over xGMI, grouped over the process' communicators, each on its device's stream
// (ordered after that device's render kernels).  The per-device buffers stay as they are (they keep accumulating over passes); the
// sum lands in device 0's b_reduced.  Several contexts on ONE physical GPU (rehearsal on a 1-GPU box) cannot form a communicator:
// there the sum is a chain of local adds in device order, which is the same exact sum.
static int reduce_accum(const float** out) {
  Ctx& c0 = g_devs[0];
  if (g_ndev == 1 && !g_comms_ready) { *out = c0.ext_accum ? c0.ext_accum : (const float*)c0.b_accum.p; return 0; }
  const size_t count = (size_t)c0.width * c0.height * 3;
  Dev0Guard guard;
  if (use_dev(0) || ensure(c0.b_reduced, count * 4)) return 1;
  // a pair of events around the reduce: from the pool of folded pairs (synchronize() folds them) when there is one
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (g_reduce_free.size() >= 2) { ev1 = g_reduce_free.back(); g_reduce_free.pop_back(); ev0 = g_reduce_free.back(); g_reduce_free.pop_back(); }
  else {
    HIP_TRY(hipEventCreate(&ev0));
    if (hipEventCreate(&ev1) != hipSuccess) { (void)hipEventDestroy(ev0); return fail("hipEventCreate failed"); }
  }
  if (hipEventRecord(ev0, c0.stream) != hipSuccess) { g_reduce_free.push_back(ev0); g_reduce_free.push_back(ev1); return fail("hipEventRecord failed"); }      // (a pair without a record must not reach the list)
  g_reduce_events.push_back(ev0); g_reduce_events.push_back(ev1);
  g_reduces += 1; g_reduce_path = g_comms_ready ? 1 : 2;
  struct Stop { hipEvent_t e; hipStream_t s; ~Stop() { (void)hipEventRecord(e, s); } } stop{ev1, c0.stream};       // recorded on every way out
  if (g_comms_ready) {
    if (ncclGroupStart() != ncclSuccess) return fail("ncclGroupStart failed");
    for (int k = 0; k < g_ndev; ++k) {
      const ncclResult_t r = ncclReduce(g_devs[k].b_accum.p, k == 0 ? c0.b_reduced.p : nullptr, count, ncclFloat32, ncclSum, 0, g_comms[k], g_devs[k].stream);
      if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail(std::string("ncclReduce: ") + ncclGetErrorString(r)); }
    }
    const ncclResult_t r = ncclGroupEnd();
    if (r != ncclSuccess) return fail(std::string("ncclGroupEnd: ") + ncclGetErrorString(r));
  } else if (g_same_gpu) {
    HIP_TRY(hipMemcpyAsync(c0.b_reduced.p, c0.b_accum.p, count * 4, hipMemcpyDeviceToDevice, c0.stream));
    for (int k = 1; k < g_ndev; ++k) {
      hipEvent_t e;
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      HIP_TRY(hipEventRecord(e, g_devs[k].stream));
      HIP_TRY(hipStreamWaitEvent(c0.stream, e, 0));
      (void)hipEventDestroy(e);
      launch_add_f32(c0.stream, (const float*)g_devs[k].b_accum.p, (float*)c0.b_reduced.p, count);
    }
    // the adds read every context's accum on c0.stream: a later render pass on stream k must not overwrite it before they are done
    hipEvent_t done;
    HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(done, c0.stream));
    for (int k = 1; k < g_ndev; ++k) HIP_TRY(hipStreamWaitEvent(g_devs[k].stream, done, 0));
    (void)hipEventDestroy(done);
  } else return fail("internal: multi-device mode without a reduce path");
  *out = (const float*)c0.b_reduced.p;
  return 0;
}

int download(float* accum_host, uint32_t* screen_host, int layout, int spp) {
  Ctx& c = g_ctx;
  if (c.width <= 0) return fail("no viewport");
  const float* acc = nullptr;
  if (reduce_accum(&acc)) return 1;
  return download_from(acc, accum_host, screen_host, layout, spp);
}

static int download_from(const float* acc_dev, float* accum_host, uint32_t* screen_host, int layout, int spp) {
  Ctx& c = g_ctx;
  const int W = c.width, H = c.height; const size_t n = (size_t)W * H;
  if (screen_host) launch_resolve(c.stream, acc_dev, (int)n, 1.0f / (float)spp, (uint32_t*)c.b_screen.p);
  if (layout == ART_LAYOUT_ADA_XY) {
    if (ensure(c.b_stage, n * 12)) return 1;
    if (accum_host) {
      launch_to_xmajor_f3(c.stream, acc_dev, (float*)c.b_stage.p, W, H);
      HIP_TRY(hipMemcpyAsync(accum_host, c.b_stage.p, n * 12, hipMemcpyDeviceToHost, c.stream));
      HIP_TRY(hipStreamSynchronize(c.stream));
    }
    if (screen_host) {
      launch_to_xmajor_u32(c.stream, (const uint32_t*)c.b_screen.p, (uint32_t*)c.b_stage.p, W, H);
      HIP_TRY(hipMemcpyAsync(screen_host, c.b_stage.p, n * 4, hipMemcpyDeviceToHost, c.stream));
    }
  } else {
    if (accum_host) HIP_TRY(hipMemcpyAsync(accum_host, acc_dev, n * 12, hipMemcpyDeviceToHost, c.stream));
    if (screen_host) HIP_TRY(hipMemcpyAsync(screen_host, c.b_screen.p, n * 4, hipMemcpyDeviceToHost, c.stream));
  }
  HIP_TRY(hipStreamSynchronize(c.stream));
  HIP_TRY(hipGetLastError());
  return 0;
}

static int debug_pass_one(const ArtPassParams* p, float* accum_host, uint32_t* screen_host, int32_t* prim_index, int32_t* mat_id, int32_t* prim_type);
// Debug_Ray_Tracing is one primary ray per pixel: in multi-device mode device 0 takes the whole frame for it (its tile ownership is
// restored afterwards) -- there is nothing to shard.
int debug_pass(const ArtPassParams* p, float* accum_host, uint32_t* screen_host, int32_t* prim_index, int32_t* mat_id, int32_t* prim_type) {
  if (g_ndev == 1) return debug_pass_one(p, accum_host, screen_host, prim_index, mat_id, prim_type);
  Dev0Guard guard;
  if (use_dev(0)) return 1;
  Ctx& c = g_ctx;
  const int rank = c.rank, nranks = c.nranks;
  c.rank = 0; c.nranks = 1;
  int rc = (c.width > 0) ? build_shard() : 0;
  if (!rc) rc = debug_pass_one(p, accum_host, screen_host, prim_index, mat_id, prim_type);
  c.rank = rank; c.nranks = nranks;
  if (c.width > 0 && build_shard()) return 1;
  return rc;
}
static int debug_pass_one(const ArtPassParams* p, float* accum_host, uint32_t* screen_host, int32_t* prim_index, int32_t* mat_id, int32_t* prim_type) {
  Ctx& c = g_ctx;
  if (check_pass(p)) return 1;
  ArtPassParams pp = *p; pp.aa_on = 0;
  DevFrame F; make_frame(&pp, F);
  const int npix = c.npix_local; const size_t n = (size_t)c.width * c.height;
  if (ensure(c.b_ids, n * 12)) return 1;
  HIP_TRY(hipMemsetAsync(c.b_ids.p, 0xff, n * 12, c.stream));
  int32_t* d_pi = (int32_t*)c.b_ids.p; int32_t* d_mi = d_pi + n; int32_t* d_pt = d_mi + n;
  const int pc = (int)std::min<int64_t>(npix, c.batch_paths);
  if (npix > 0 && ensure_paths((size_t)pc, 1, false)) return 1;
  for (int px0 = 0; px0 < npix; px0 += pc) {
    const int pn = std::min(pc, npix - px0);
    DevPaths bank[2]; std::memset(bank, 0, sizeof bank);
    carve(bank, pn, 1, false);
    DevPaths q = bank[0];
    q.P = pn; q.npix = pn; q.pixmap = (const uint32_t*)c.b_pixmap.p + px0; q.sample_base = 0; q.slot_id = nullptr; q.final_flags = q.flags;
    launch_raygen(c.stream, F, c.scene, q);
    c.camera_rays += (uint64_t)pn;
    if (trace(q, pn)) return 1;
    launch_debug(c.stream, F, c.scene, q, accum_ptr(), d_pi, d_mi, d_pt);
  }
  HIP_TRY(hipGetLastError());
  // ray_tracer.adb:249-257: the debug image is resolved without dividing by spp
  if (download_from(accum_ptr(), accum_host, screen_host, p->layout, 1)) return 1;
  auto copy_ids = [&](int32_t* host, const int32_t* dev) -> int {
    if (!host) return 0;
    if (p->layout == ART_LAYOUT_ADA_XY) {
      if (ensure(c.b_stage, n * 12)) return 1;
      launch_to_xmajor_u32(c.stream, (const uint32_t*)dev, (uint32_t*)c.b_stage.p, c.width, c.height);
      HIP_TRY(hipMemcpyAsync(host, c.b_stage.p, n * 4, hipMemcpyDeviceToHost, c.stream));
    } else HIP_TRY(hipMemcpyAsync(host, dev, n * 4, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    return 0;
  };
  if (copy_ids(prim_index, d_pi) || copy_ids(mat_id, d_mi) || copy_ids(prim_type, d_pt)) return 1;
  return synchronize_one();
}

int trace_rays(const float* origins, const float* dirs, const float* tfar, int64_t n, ArtHit* out, int kernel, ArtStats* st) {
  Ctx& c = g_ctx;
  if (!c.scene_ready) return fail("no scene uploaded");
  if (n <= 0 || n > (1ll << 28) || !origins || !dirs || !out) return fail("art_trace_rays: bad arguments");
  if (kernel != TRACE_COOP && kernel != TRACE_SIMPLE) return fail("art_trace_rays: unknown kernel");
  const size_t N = (size_t)n;
  if (ensure(c.b_rays, N * 12 * 4)) return 1;      // 7 N ray floats, N of padding, N 16-byte hit records
  std::vector<float> soa(7 * N);
  for (size_t i = 0; i < N; ++i) {
    soa[i] = origins[3 * i]; soa[N + i] = origins[3 * i + 1]; soa[2 * N + i] = origins[3 * i + 2];
    soa[3 * N + i] = dirs[3 * i]; soa[4 * N + i] = dirs[3 * i + 1]; soa[5 * N + i] = dirs[3 * i + 2];
    soa[6 * N + i] = tfar ? tfar[i] : kInfinity;
  }
  float* d = (float*)c.b_rays.p;
  HIP_TRY(hipMemcpyAsync(d, soa.data(), 7 * N * 4, hipMemcpyHostToDevice, c.stream));
  HIP_TRY(hipMemsetAsync(d + 8 * N, 0xff, 4 * N * 4, c.stream));
  if (st) HIP_TRY(hipMemsetAsync(c.d_counters + 3, 0, 8 * sizeof(unsigned long long), c.stream));
  DevPaths q; std::memset(&q, 0, sizeof q);
  q.ray_ox = d; q.ray_oy = d + N; q.ray_oz = d + 2 * N; q.ray_dx = d + 3 * N; q.ray_dy = d + 4 * N; q.ray_dz = d + 5 * N; q.ray_tfar = d + 6 * N;
  q.hit = (DevHit*)(d + 8 * N);                                      // 16-byte aligned: the buffer is, and 8 N floats precede it
  const int saved_kernel = c.trace_kernel; const bool saved_count = c.count_tests;
  c.trace_kernel = kernel; c.count_tests = (st != nullptr);
  const int rc = trace(q, (int)n, /*timed=*/st != nullptr);       // without stats: no events, no counter read-back (the per-ray seam)
  c.trace_kernel = saved_kernel; c.count_tests = saved_count;
  if (rc) return 1;
  std::vector<DevHit> hits(N);
  HIP_TRY(hipMemcpyAsync(hits.data(), d + 8 * N, N * sizeof(DevHit), hipMemcpyDeviceToHost, c.stream));
  HIP_TRY(hipStreamSynchronize(c.stream));
  HIP_TRY(hipGetLastError());
  HostScene& hs = c.host_scene;
  bind_host_pointers(hs);
  for (size_t i = 0; i < N; ++i) {
    ArtHit& h = out[i];
    const uint32_t key = hits[i].key;
    h.t = hits[i].t; h.u = hits[i].u; h.v = hits[i].v;
    if (key == KEY_MISS) { h.is_hit = 0; h.prim_type = -1; h.prim_index = -1; h.mat_id = -1; h.mat = -1; h.normal[0] = h.normal[1] = h.normal[2] = 0.0f; continue; }
    const f3 o = mk3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), dd = mk3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    const Surface sf = surface_at(hs.hdr, o, dd, h.t, key, h.u, h.v);
    const uint32_t cls = key & ~KEY_INDEX_MASK;
    h.is_hit = 1; h.prim_index = (int32_t)(key & KEY_INDEX_MASK); h.mat_id = sf.mat_id; h.mat = sf.mat;
    h.prim_type = (cls == KEY_CORNELL) ? 0 : (cls == KEY_SPHERE) ? 1 : (cls == KEY_QUAD) ? 3 : 2;
    h.normal[0] = sf.normal.x; h.normal[1] = sf.normal.y; h.normal[2] = sf.normal.z;
  }
  if (st) {
    if (synchronize()) return 1;
    *st = c.stats;
  }
  return 0;
}

void shutdown() {
  if (g_devs[0].device_ready && !use_dev(0)) reset_reduce_info();
  if (g_comms_ready) { for (int k = 0; k < g_ndev; ++k) (void)ncclCommDestroy(g_comms[k]); g_comms_ready = false; }
  g_same_gpu = false; g_multi_api = false;
  for (int k = g_ndev - 1; k >= 0; --k) {
  g_cur = &g_devs[k];
  Ctx& c = g_ctx;
  if (c.device_ready) {
    if (c.device >= 0) (void)hipSetDevice(c.device);
    (void)hipDeviceSynchronize();
    DevBuf* bufs[] = {&c.b_spheres, &c.b_sphere_mat, &c.b_lights, &c.b_materials, &c.b_bf_pos, &c.b_bf_nrm, &c.b_bf_uv, &c.b_bf_idx,
                      &c.b_nodes, &c.b_qnodes, &c.b_tris, &c.b_qtris, &c.b_m_shade, &c.b_accum, &c.b_screen, &c.b_stage,
                      &c.b_pixmap, &c.b_paths, &c.b_rays, &c.b_ids, &c.b_queue, &c.b_ovf,
                      &c.b_inst, &c.b_tlas_nodes, &c.b_tlas_tris, &c.b_blas_nodes, &c.b_blas_tris};
    for (DevBuf* b : bufs) b->release();
    if (c.d_cursor) (void)hipFree(c.d_cursor);
    if (c.d_scene) (void)hipFree(c.d_scene);
    if (c.d_counters) (void)hipFree(c.d_counters);
    if (c.d_live) (void)hipFree(c.d_live);
    if (c.d_items) (void)hipFree(c.d_items);
    for (hipEvent_t e : c.ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : c.pass_events) (void)hipEventDestroy(e);
    for (Ctx::PassClock* pc : c.pass_clock) delete pc;      // (after hipDeviceSynchronize: no callback is pending)
    c.b_reduced.release();
    if (c.own_stream) (void)hipStreamDestroy(c.own_stream);
  }
  c = Ctx();
  }
  g_ndev = 1; g_cur = &g_devs[0];
}

}  // namespace art

// ================================================================================================
using namespace art;

extern "C" {

int art_init(int device_ordinal) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ctx.device_ready && device_ordinal >= 0 && device_ordinal != g_ctx.device) return fail("art_init: already bound to another device; call art_shutdown first");
  g_ctx.device = device_ordinal;
  if (const char* e = getenv("ART_TRACE_KERNEL")) g_ctx.trace_kernel = (std::strcmp(e, "simple") == 0) ? TRACE_SIMPLE : TRACE_COOP;
  if (const char* e = getenv("ART_BATCH_PATHS")) g_ctx.batch_paths = std::min<int64_t>(1ll << 27, std::max<int64_t>(1024, atoll(e)));
  return ensure_device();
}

// One process, n GPUs (SURVEY 8e / 8b "Threading": the Ada host calls Render_Pass from its environment task, test.adb:50, so the
// fan-out over the node's GPUs has to happen below the C ABI).  ordinals == NULL: devices 0..n-1.  Every device gets its own
// context, stream, replicated scene and path buffers; device k owns the 32x32 pixel tiles (bx, by) with (bx + 3 by) mod n == k (diagonals: build_pixmap); art_render_pass /
// art_download add the float3 framebuffers into device 0 with ONE RCCL reduce over xGMI.  Listing the same ordinal several times puts
// several contexts on one GPU (a rehearsal of the whole path on a 1-GPU box; the reduce is then a local sum).
int art_init_devices(int32_t n, const int32_t* ordinals) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (n < 1 || n > kMaxDevices) return fail("art_init_devices: 1.." + std::to_string(kMaxDevices) + " devices");
  for (int k = 0; k < kMaxDevices; ++k) if (g_devs[k].device_ready) return fail("art_init_devices: already initialised; call art_shutdown first");
  int avail = 0;
  hipError_t e = hipGetDeviceCount(&avail);
  if (e != hipSuccess || avail <= 0) return fail("no HIP device available (this library has no CPU path): " + std::string(hipGetErrorString(e)));
  int ord[kMaxDevices]; bool distinct = true, same = true;
  for (int k = 0; k < n; ++k) {
    ord[k] = ordinals ? ordinals[k] : k;
    if (ord[k] < 0 || ord[k] >= avail) return fail("art_init_devices: device " + std::to_string(ord[k]) + " does not exist (" + std::to_string(avail) + " visible)");
    for (int i = 0; i < k; ++i) { if (ord[i] == ord[k]) distinct = false; else same = false; }
  }
  if (n > 1 && !distinct && !same) return fail("art_init_devices: the ordinals must be all different (one context per GPU) or all equal (rehearsal on one GPU)");
  const Ctx opts = g_devs[0];                       // options set before initialisation apply to every device
  Dev0Guard guard;
  g_ndev = n;
  for (int k = 0; k < n; ++k) {
    Ctx& c = g_devs[k];
    c = Ctx();
    c.trace_kernel = opts.trace_kernel; c.batch_paths = opts.batch_paths; c.bvh_params = opts.bvh_params; c.node_min = opts.node_min; c.refill_min = opts.refill_min;
    c.queue_segments = opts.queue_segments; c.ray_chunk = opts.ray_chunk; c.shadow_anyhit = opts.shadow_anyhit; c.shade_split = opts.shade_split; c.skip_null_shadow = opts.skip_null_shadow; c.inst_coop = opts.inst_coop; c.opt_shade_per = opts.opt_shade_per; c.lds_stack_cap = opts.lds_stack_cap; c.paths_contiguous = opts.paths_contiguous; c.hot_pad = opts.hot_pad; c.paths_spread_mb = opts.paths_spread_mb; c.paths_spread_holes = opts.paths_spread_holes;
    c.opt_blocks_per_cu = opts.opt_blocks_per_cu; c.count_tests = opts.count_tests;
    c.device = ord[k]; c.rank = k; c.nranks = n; c.tile = 32;
    if (use_dev(k) || ensure_device()) { shutdown(); return 1; }
    if (n > 1) {
      if (hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking) != hipSuccess) { shutdown(); return fail("hipStreamCreate failed"); }
      c.stream = c.own_stream;
    }
  }
  // ART_FORCE_RCCL=1: build the communicator for a single device too, so that a 1-GPU box exercises the very RCCL calls of the n-GPU path
  const bool force_rccl = (n == 1) && getenv("ART_FORCE_RCCL") && std::atoi(getenv("ART_FORCE_RCCL")) != 0;
  if ((n > 1 && distinct) || force_rccl) {
    const ncclResult_t r = ncclCommInitAll(g_comms, n, ord);
    if (r != ncclSuccess) { shutdown(); return fail(std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); }
    g_comms_ready = true;
  }
  g_same_gpu = (n > 1 && !distinct);
  g_multi_api = true;
  return 0;
}

int32_t art_device_count(void) { std::lock_guard<std::mutex> lk(g_mu); return g_ndev; }

#define SINGLE_DEVICE_ONLY(name) if (g_ndev > 1) return fail(name ": not available after art_init_devices(n > 1); the library shards and reduces by itself")

int art_set_stream(void* hip_stream) { std::lock_guard<std::mutex> lk(g_mu); SINGLE_DEVICE_ONLY("art_set_stream"); g_ctx.stream = (hipStream_t)hip_stream; return 0; }

int art_upload_scene(const ArtSceneDesc* scene) { std::lock_guard<std::mutex> lk(g_mu); return upload_scene(scene); }

int art_resize(int32_t w, int32_t h) { std::lock_guard<std::mutex> lk(g_mu); return resize(w, h); }

int art_set_shard(int32_t rank, int32_t nranks, int32_t tile) {
  std::lock_guard<std::mutex> lk(g_mu);
  SINGLE_DEVICE_ONLY("art_set_shard");
  if (nranks < 1 || rank < 0 || rank >= nranks || tile < 1) return fail("art_set_shard: bad arguments");
  g_ctx.rank = rank; g_ctx.nranks = nranks; g_ctx.tile = tile;
  if (g_ctx.width > 0) return build_shard();
  return 0;
}

int art_render_pass(const ArtPassParams* p, float* accum_host, uint32_t* screen_host, int32_t* spp_inout) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (render_pass_device(p, spp_inout)) return 1;
  if (accum_host || screen_host) { if (download(accum_host, screen_host, p->layout, g_ctx.spp)) return 1; }
  return synchronize();
}

int art_debug_hit_pass(const ArtPassParams* p, float* accum_host, uint32_t* screen_host, int32_t* prim_index, int32_t* mat_id, int32_t* prim_type) {
  std::lock_guard<std::mutex> lk(g_mu);
  return debug_pass(p, accum_host, screen_host, prim_index, mat_id, prim_type);
}

int art_bind_accum(void* device_accum_rowmajor) { std::lock_guard<std::mutex> lk(g_mu); SINGLE_DEVICE_ONLY("art_bind_accum"); g_ctx.ext_accum = (float*)device_accum_rowmajor; return 0; }
void* art_accum_device(void) {          // device 0; in multi-device mode the reduced framebuffer (valid after a reduce: art_reduce / art_download)
  std::lock_guard<std::mutex> lk(g_mu);
  return (g_ndev > 1) ? g_devs[0].b_reduced.p : (void*)accum_ptr();
}

// multi-device: enqueue the framebuffer reduce to device 0 (bench.py times it inside its step loop); no-op with one device
int art_reduce(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  const float* acc = nullptr;
  return reduce_accum(&acc);
}

int art_download(float* accum_host, uint32_t* screen_host, int32_t layout, int32_t spp) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (spp <= 0) return fail("art_download: spp must be positive");
  return download(accum_host, screen_host, layout, spp);
}

int art_synchronize(void) { std::lock_guard<std::mutex> lk(g_mu); return synchronize(); }

int art_trace_rays(const float* origins, const float* dirs, const float* tfar, int64_t n, ArtHit* out, int32_t kernel, ArtStats* stats) {
  std::lock_guard<std::mutex> lk(g_mu);          // ray queries run on device 0
  return trace_rays(origins, dirs, tfar, n, out, kernel, stats);
}

int art_export_bvh(float* nodes, int64_t node_cap, float* tris, int64_t tri_cap, ArtBvhInfo* info) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.scene_ready) return fail("no scene uploaded");
  Bvh8& b = g_ctx.host_scene.bvh;
  if (g_ctx.scene.n_inst > 0 && (nodes || tris)) return fail("art_export_bvh: an instanced scene has a two-level tree; only its sizes are reported (ArtBvhInfo)");
  if (g_ctx.host_scene.gpu_built && b.nodes.empty() && (nodes || tris)) {     // the GPU-built tree is fetched on first request
    b.nodes.resize((size_t)b.n_nodes * node_floats(b.width)); b.tris.resize((size_t)b.n_tris * kTriFloats);
    HIP_TRY(hipMemcpy(b.nodes.data(), g_ctx.b_nodes.p, b.nodes.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b.tris.data(), g_ctx.b_tris.p, b.tris.size() * 4, hipMemcpyDeviceToHost));
  }
  if (info) { info->n_nodes = b.n_nodes; info->n_tris = b.n_tris; info->max_stack = b.max_stack; info->node_width = b.width; info->build_ms = g_ctx.host_scene.bvh_build_ms; }
  if (nodes) { if (node_cap < (int64_t)b.nodes.size()) return fail("art_export_bvh: node buffer too small"); std::memcpy(nodes, b.nodes.data(), b.nodes.size() * 4); }
  if (tris) { if (tri_cap < (int64_t)b.tris.size()) return fail("art_export_bvh: triangle buffer too small"); std::memcpy(tris, b.tris.data(), b.tris.size() * 4); }
  return 0;
}

}  // extern "C"
namespace art {
// the uploaded tree of device 0 as host arrays (the packets art_export_bvh returns); the caller holds g_mu
int fetch_host_bvh(std::vector<float>& nodes, std::vector<float>& tris, int& width, int& n_tris) {
  Ctx& c = g_devs[0];
  if (!c.scene_ready) return fail("no scene uploaded");
  Bvh8& b = c.host_scene.bvh;
  width = b.width; n_tris = b.n_tris;
  if (c.host_scene.gpu_built && b.nodes.empty()) {
    nodes.resize((size_t)b.n_nodes * node_floats(b.width)); tris.resize((size_t)b.n_tris * kTriFloats);
    if (use_dev(0)) return 1;
    HIP_TRY(hipMemcpy(nodes.data(), c.b_nodes.p, nodes.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(tris.data(), c.b_tris.p, tris.size() * 4, hipMemcpyDeviceToHost));
  } else { nodes = b.nodes; tris = b.tris; }
  return 0;
}
}  // namespace art
extern "C" {

int art_get_reduce_info(ArtReduceInfo* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out) return fail("null ArtReduceInfo");
  if (synchronize()) return 1;
  std::memset(out, 0, sizeof *out);
  out->devices = g_ndev;
  if (g_comms_ready) { int n = 0; if (ncclCommCount(g_comms[0], &n) != ncclSuccess) return fail("ncclCommCount failed"); out->rccl_ranks = n; }
  fold_pass_clocks();                      // (synchronize() above folded the reduces' event pairs and left every stream idle)
  out->path = g_reduce_path; out->reduces = g_reduces; out->reduce_ms = g_reduce_ms;
  for (int k = 0; k < g_ndev && k < 8; ++k) {
    const Ctx& c = g_devs[k];
    out->device_pass_ms[k] = c.stats.pass_ms; out->device_busy_ms[k] = c.busy_ms; out->device_idle_ms[k] = c.idle_ms; out->device_start_skew_ms[k] = c.start_skew_ms;
  }
  out->passes = g_passes; out->passes_overlapped = g_passes_overlapped;
  return 0;
}

int art_get_stage_stats(ArtStageStats* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out) return fail("null ArtStageStats");
  if (synchronize()) return 1;
  *out = g_devs[0].stage;
  return 0;
}

int art_get_stats(ArtStats* out) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out) return fail("null stats");
  if (synchronize()) return 1;
  *out = g_devs[0].stats;                 // trace_ms / trace_launches / wave counters: device 0's (per-device quantities)
  for (int k = 1; k < g_ndev; ++k) {      // work counters: the whole job
    const ArtStats& t = g_devs[k].stats;
    out->rays += t.rays; out->samples += t.samples; out->box_tests += t.box_tests; out->tri_tests += t.tri_tests;
    out->node_visits += t.node_visits; out->leaf_visits += t.leaf_visits; out->traced_rays += t.traced_rays;
    out->lost_paths += t.lost_paths;      // the compacted work set's self-check ("must stay 0") covers every device
    out->pass_ms = std::max(out->pass_ms, t.pass_ms);
  }
  return 0;
}

static int set_option_one(const std::string& n, int64_t value);
int art_set_option(const char* name, int64_t value) {      // applies to every device of the process
  std::lock_guard<std::mutex> lk(g_mu);
  if (!name) return fail("null option");
  const std::string n(name);
  Ctx* const saved = g_cur;
  int rc = 0;
  for (int k = 0; k < g_ndev && !rc; ++k) { g_cur = &g_devs[k]; rc = set_option_one(n, value); }
  g_cur = saved;
  return rc;
}
static int set_option_one(const std::string& n, int64_t value) {
  if (n == "trace_kernel") { if (value != TRACE_COOP && value != TRACE_SIMPLE) return fail("trace_kernel: 0 (cooperative) or 1 (simple)"); g_ctx.trace_kernel = (int)value; }
  else if (n == "queue_segments") { if (value != 1 && value != 2 && value != 4 && value != 8) return fail("queue_segments: 1, 2, 4 or 8"); g_ctx.queue_segments = (int)value; }
  else if (n == "batch_paths") { if (value < 1024 || value > (1ll << 27)) return fail("batch_paths: 1024..2^27 (2 rays per path slot; the trace kernel addresses a ray's 16-byte hit record by a 32-bit byte offset)"); g_ctx.batch_paths = value; g_ctx.auto_phase = 0; g_ctx.auto_redo = 0; g_ctx.auto_gen += 1; }
  else if (n == "hot_pad") { if (value < 0 || value > (1 << 24) || (value & 63)) return fail("hot_pad: a multiple of 64 items, 0 .. 2^24"); g_ctx.hot_pad = (int)value; g_ctx.b_paths.release(); }
  else if (n == "paths_spread") { if (value < -1 || value > 65536) return fail("paths_spread: chunk size in MB, 0 = off (plain hipMalloc), -1 = automatic"); g_ctx.paths_spread_mb = (int)value; g_ctx.b_paths.release(); }
  else if (n == "spread_fail_at") { g_ctx.spread_fail_at = (int)value; g_ctx.b_paths.release(); }      // test option: creating chunk number `value` of the path state fails (-1: never)
  else if (n == "paths_spread_holes") { g_ctx.paths_spread_holes = value != 0; g_ctx.b_paths.release(); }
  else if (n == "paths_contiguous") { g_ctx.paths_contiguous = value != 0; g_ctx.b_paths.release(); }
  else if (n == "inject_lost") { g_ctx.inject_lost = value != 0; }      // test option: the next pass counts one lost path in its first batch
  else if (n == "blocks_per_cu") { g_ctx.opt_blocks_per_cu = (int)value; g_ctx.blocks_per_cu = 0; }
  else if (n == "count_tests") { g_ctx.count_tests = value != 0; }
  else if (n == "shadow_anyhit") { g_ctx.shadow_anyhit = value != 0; }
  else if (n == "shade_split") { g_ctx.shade_split = value != 0; }
  else if (n == "skip_null_shadow") { g_ctx.skip_null_shadow = value != 0; }
  else if (n == "inst_coop") { g_ctx.inst_coop = value != 0; }
  else if (n == "shade_per") { if (value != 0 && value != 2 && value != 4) return fail("shade_per: 0 (measured), 2 or 4"); g_ctx.opt_shade_per = (int)value; }
  else if (n == "ray_chunk") { if (value < 16 || value > 4096 || (value & 15)) return fail("ray_chunk: a multiple of 16, 16..4096"); g_ctx.ray_chunk = (int)value; }
  else if (n == "refill_min") { if (value < 1 || value > 8) return fail("refill_min: 1..8"); g_ctx.refill_min = (int)value; }
  else if (n == "node_min") { if (value < 0 || value > 8) return fail("node_min: 1..8, 0 = automatic"); g_ctx.node_min = (int)value; }
  else if (n == "bvh_width") { if (value != 4 && value != 8) return fail("bvh_width: 4 or 8"); g_ctx.bvh_params.width = (int)value; }
  else if (n == "lds_stack_cap") { if (value < 0 || value > kStackEntries) return fail("lds_stack_cap: 0 (automatic) .. 160"); g_ctx.lds_stack_cap = (int)value; }
  else if (n == "bvh_max_leaf") { if (value < 0 || value > kMaxLeafTris) return fail("bvh_max_leaf: 1..8, 0 = defaults"); g_ctx.bvh_params.max_leaf = value ? (int)value : BvhBuildParams().max_leaf; g_ctx.bvh_params.gpu_max_leaf = (int)value; }
  else if (n == "inst_open") { if (value < 0 || value > 4096) return fail("inst_open: 1 .. 4096 entry points per instance, 0 = chosen from the instances' overlap"); g_ctx.bvh_params.inst_open = (int)value; }
  else if (n == "bvh_spatial_splits") { g_ctx.bvh_params.spatial_alpha = value ? 1.0e-5f : -1.0f; }   // host builder: SBVH reference splitting
  else if (n == "bvh_builder") { if (value < 0 || value > 3) return fail("bvh_builder: 0 host SAH, 1 GPU LBVH, 2 GPU PLOC, 3 GPU SAH"); g_ctx.bvh_params.builder = (int)value; }
  else if (n == "bvh_ploc_radius") { if (value < 1 || value > 64) return fail("bvh_ploc_radius: 1..64"); g_ctx.bvh_params.ploc_radius = (int)value; }
  else if (n == "bvh_leaf_base_milli") { g_ctx.bvh_params.leaf_base = (float)value / 1000.0f; }
  else if (n == "bvh_tri_cost_milli") { g_ctx.bvh_params.tri_cost = (float)value / 1000.0f; }
  else if (n == "bvh_node_cost_milli") { g_ctx.bvh_params.node_cost = (float)value / 1000.0f; }
  else return fail("unknown option " + n);
  return 0;
}

const char* art_last_error(void) { t_err = g_err; return t_err.c_str(); }

void art_shutdown(void) { std::lock_guard<std::mutex> lk(g_mu); shutdown(); }

}  // extern "C"

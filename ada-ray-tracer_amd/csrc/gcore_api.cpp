// gcore_api.cpp -- the legacy geometry-core seam (cpp/embree_connect.cpp:51-244) on top of the HIP BVH.
// Same six symbols that scene_hydra_embree.adb:37-66 imports, with the defects listed in SURVEY 3.4
// corrected: counts are element counts, vertexNum vertices are copied (the reference copies
// maxVertexId, :122,129), the global scene exists, and hitFound means t_near < t < t_far (the reference
// compares ray.tnear with itself, :220, so it never reports a hit).
//
// Semantics kept from Embree: every mesh is its own geometry (geomIndex 0), instances carry a 3x4
// row-major transform read from a 16-float block (:169), hits are two-sided true closest hits, the
// normal is the unnormalised geometric normal Ng = cross(v1-v0, v2-v0) and texCoord = barycentrics (u,v)
// with hit = (1-u-v) v0 + u v1 + v v2.  Two-sidedness is obtained by uploading each triangle in both
// windings (the kernel's Moeller-Trumbore test is the reference's one-sided one, geometry.adb:243).
//
// gcore_closest_hit is the per-ray compatibility path (SURVEY 8b "per-ray fallback for debugging": "host BVH walk or batched queue");
// the frame-level art_render_pass is the fast path.  Round 4: a single-ray call walks the committed tree ON THE CALLING THREAD (the
// reference calls it from up to 28 Ada tasks, one ray each, and Embree answers on the caller's core: embree_connect.cpp:196-238) --
// the same tree, the same published walk and the same triangle arithmetic as the GPU kernels, so the same hits, bit for bit; no launch,
// no lock beyond a shared one against a concurrent commit.  gcore_closest_hit_n (the batch form) stays on the GPU.
// The scene committed here REPLACES the art_* scene of the process (one backend singleton, like g_data in embree_connect.cpp:12-22):
// a process uses either seam at a time.
#include <algorithm>
#include <array>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <atomic>
#include <shared_mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include <cmath>

#include "art_api_internal.h"
#include "art_instanced_build.h"

namespace {

// ---- two-level scene (art_instanced.h): used when instancing is real (>= kTwoLevelMinInstances instances, or forced)
constexpr int kTwoLevelMinInstances = 16;
struct TwoLevel {
  bool on = false;
  art::TwoLevelHost host;
  // device copies
  void *d_tlas_nodes = nullptr, *d_tlas_tris = nullptr, *d_blas_nodes = nullptr, *d_blas_tris = nullptr, *d_inst = nullptr;
  void *d_rays = nullptr, *d_hits = nullptr; size_t ray_cap = 0;
  void release() {
    void** ps[] = {&d_tlas_nodes, &d_tlas_tris, &d_blas_nodes, &d_blas_tris, &d_inst, &d_rays, &d_hits};
    for (void** p : ps) { if (*p) (void)hipFree(*p); *p = nullptr; }
    ray_cap = 0;
  }
};
int g_force_two_level = -1;          // -1 automatic, 0 never, 1 always (gcore_set_two_level, tests)


struct GMesh { std::vector<float> verts; std::vector<int32_t> idx; };
struct GInst { int mesh; float m[12]; };

struct GState {
  bool inited = false, committed = false;
  std::vector<GMesh> meshes;
  std::vector<GInst> insts;
  // per uploaded triangle (both windings share the entry): instance, primitive within the mesh
  std::vector<int32_t> tri_inst, tri_prim;
  std::vector<float> wverts;   // world-space vertices of every instance, 3 per vertex
  std::vector<int32_t> widx;
  TwoLevel two;
  // the committed flattened tree on the host (binary32 node packets + triangle records, as art_export_bvh returns them): what a
  // single-ray gcore_closest_hit walks.  (Two-level scenes walk two.host, which the build left on the host anyway.)
  // the flattened tree as host arrays, for the single-ray host walk.  Fetched on the FIRST single-ray call after a commit (callers
  // that only use gcore_closest_hit_n or the art_* seam never pay the device -> host copy of the tree, nor its host memory)
  std::vector<float> h_nodes, h_tris; int h_width = 4, h_ntris = 0;
} g;
// single-ray queries hold it shared, commit / destroy exclusively (the reference has no such guard: it commits before it renders)
std::shared_mutex g_scene_rw;
std::atomic<bool> g_h_ready{false};      // GState::h_nodes / h_tris hold the committed tree

template <typename T> bool to_device(void** p, const std::vector<T>& v) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (v.empty()) return true;
  if (hipMalloc(p, v.size() * sizeof(T)) != hipSuccess) return false;
  return hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

// one tree per mesh (object space, both windings), one tree over the instances' world boxes (art_instanced_build.cpp), then to HBM
bool build_two_level(std::string& err) {
  TwoLevel& T = g.two;
  T.release(); T = TwoLevel();
  std::vector<art::InstMeshIn> meshes; std::vector<art::InstIn> insts;
  for (const GMesh& m : g.meshes) meshes.push_back({m.verts.data(), m.verts.size() / 3, m.idx.data(), m.idx.size() / 3});
  for (const GInst& in : g.insts) { art::InstIn x; x.mesh = in.mesh; std::memcpy(x.m, in.m, sizeof x.m); insts.push_back(x); }
  if (!art::build_two_level_host(meshes, insts, T.host, err)) return false;
  if (T.host.inst.size() != insts.size()) std::printf("[c_gcore]: %d instances with a singular matrix skipped\n", (int)(insts.size() - T.host.inst.size()));
  g.tri_inst = T.host.inst_src;          // instance record -> caller's instance index
  if (!to_device(&T.d_tlas_nodes, T.host.tlas.nodes) || !to_device(&T.d_tlas_tris, T.host.tlas.tris) || !to_device(&T.d_blas_nodes, T.host.blas_nodes) ||
      !to_device(&T.d_blas_tris, T.host.blas_tris) || !to_device(&T.d_inst, T.host.inst)) { err = "device upload of the two-level scene failed"; return false; }
  T.on = true;
  return true;
}

}  // namespace

extern "C" {

void gcore_destroy(void) {
  std::unique_lock<std::shared_mutex> wr(g_scene_rw);
  std::lock_guard<std::mutex> lk(art::g_mu);
  g.two.release();
  g = GState(); g_h_ready = false;
}

// -1 automatic (two-level from 16 instances on), 0 always flatten, 1 always two-level; takes effect at the next gcore_commit_scene
void gcore_set_two_level(int mode) { std::lock_guard<std::mutex> lk(art::g_mu); g_force_two_level = mode; }

void gcore_init_and_clear(void) {
  gcore_destroy();
  std::lock_guard<std::mutex> lk(art::g_mu);
  g.inited = true;
  if (art::ensure_device()) std::printf("[c_gcore]: %s\n", art_last_error());
}

int gcore_add_mesh_3f(const float* a_vertices3f, int a_vertexNum, const int* a_indices, int a_indicesNum) {
  std::unique_lock<std::shared_mutex> wr(g_scene_rw);
  std::lock_guard<std::mutex> lk(art::g_mu);
  if (!a_vertices3f || !a_indices || a_vertexNum <= 0 || a_indicesNum < 3) {
    std::printf("[c_gcore]: gcore_add_mesh_3f, bad arguments\n");
    return 0;
  }
  GMesh m;
  m.verts.assign(a_vertices3f, a_vertices3f + 3 * (size_t)a_vertexNum);
  const int triNum = a_indicesNum / 3;
  m.idx.resize(3 * (size_t)triNum);
  for (int i = 0; i < triNum; ++i) {
    int A = a_indices[3 * i], B = a_indices[3 * i + 1], C = a_indices[3 * i + 2];
    if (A >= a_vertexNum || B >= a_vertexNum || C >= a_vertexNum || A < 0 || B < 0 || C < 0) A = B = C = a_vertexNum - 1;   // embree_connect.cpp:98-109
    m.idx[3 * i] = A; m.idx[3 * i + 1] = B; m.idx[3 * i + 2] = C;
  }
  g.meshes.push_back(std::move(m));
  g.committed = false;
  return (int)g.meshes.size() - 1;   // like the reference: the first mesh has id 0
}

void gcore_instance_meshes(int a_geomId, const float* a_matrices16f, int a_matrixNum) {
  std::unique_lock<std::shared_mutex> wr(g_scene_rw);
  std::lock_guard<std::mutex> lk(art::g_mu);
  if (a_geomId < 0 || a_geomId >= (int)g.meshes.size() || !a_matrices16f) {
    std::printf("gcore_instance_meshes, bad meshId = %d\n", a_geomId);
    return;
  }
  for (int k = 0; k < a_matrixNum; ++k) {
    GInst in; in.mesh = a_geomId;
    std::memcpy(in.m, a_matrices16f + 16 * (size_t)k, 12 * sizeof(float));   // RTC_FORMAT_FLOAT3X4_ROW_MAJOR
    g.insts.push_back(in);
  }
  g.committed = false;
}

void gcore_commit_scene(void) {
  std::unique_lock<std::shared_mutex> wr(g_scene_rw);
  std::lock_guard<std::mutex> lk(art::g_mu);
  g.committed = false;
  g.tri_inst.clear(); g.tri_prim.clear(); g.wverts.clear(); g.widx.clear(); g.h_nodes.clear(); g.h_tris.clear(); g.h_ntris = 0; g_h_ready = false;
  g.two.release(); g.two.on = false;
  const bool two_level = (g_force_two_level == 1) || (g_force_two_level < 0 && (int)g.insts.size() >= kTwoLevelMinInstances);
  if (two_level) {                       // embree_connect.cpp:147-184: one tree per mesh, instances on top
    if (g.insts.empty()) { std::printf("[c_gcore]: gcore_commit_scene, no instances\n"); return; }
    std::string err;
    if (art::ensure_device()) { std::printf("[c_gcore]: %s\n", art_last_error()); return; }
    if (build_two_level(err)) { g.committed = true; return; }
    // e.g. a tree deeper than the two-level search's stacks: the flattened upload below has its own (checked) limits
    std::printf("[c_gcore]: two-level build: %s; flattening the instances instead\n", err.c_str());
    g.two.release(); g.two.on = false; g.tri_inst.clear();
  }
  for (size_t ii = 0; ii < g.insts.size(); ++ii) {
    const GInst& in = g.insts[ii];
    const GMesh& m = g.meshes[in.mesh];
    const int32_t base = (int32_t)(g.wverts.size() / 3);
    for (size_t v = 0; v < m.verts.size() / 3; ++v) {
      const float x = m.verts[3 * v], y = m.verts[3 * v + 1], z = m.verts[3 * v + 2];
      for (int r = 0; r < 3; ++r) g.wverts.push_back(in.m[4 * r] * x + in.m[4 * r + 1] * y + in.m[4 * r + 2] * z + in.m[4 * r + 3]);
    }
    for (size_t t = 0; t < m.idx.size() / 3; ++t) {
      const int32_t a = base + m.idx[3 * t], b = base + m.idx[3 * t + 1], c = base + m.idx[3 * t + 2];
      g.widx.push_back(a); g.widx.push_back(b); g.widx.push_back(c);   // front winding: prim 2k
      g.widx.push_back(a); g.widx.push_back(c); g.widx.push_back(b);   // back winding:  prim 2k+1
      g.tri_inst.push_back((int32_t)ii); g.tri_prim.push_back((int32_t)t);
    }
  }
  if (g.widx.empty()) { std::printf("[c_gcore]: gcore_commit_scene, no instances\n"); return; }
  const int32_t nv = (int32_t)(g.wverts.size() / 3), nt = (int32_t)(g.widx.size() / 3);
  std::vector<float> nrm(g.wverts.size(), 0.0f);
  std::vector<int32_t> matid((size_t)nt, 0);
  ArtMaterial mat; std::memset(&mat, 0, sizeof mat); mat.type = ART_MAT_LAMBERT;
  ArtLight light; std::memset(&light, 0, sizeof light); light.shape = ART_LIGHT_SPHERE; light.radius = 1.0f; light.surfaceArea = 1.0f; light.mat = 0;
  ArtMesh mesh; std::memset(&mesh, 0, sizeof mesh);
  mesh.mode = ART_MESH_CLOSEST; mesh.nverts = nv; mesh.ntris = nt; mesh.pos = g.wverts.data(); mesh.nrm = nrm.data(); mesh.idx = g.widx.data(); mesh.matid = matid.data();
  ArtSceneDesc sd; std::memset(&sd, 0, sizeof sd);
  sd.n_lights = 1; sd.lights = &light; sd.n_materials = 1; sd.materials = &mat; sd.n_meshes = 1; sd.meshes = &mesh;
  sd.cam_matrix[0] = sd.cam_matrix[5] = sd.cam_matrix[10] = sd.cam_matrix[15] = 1.0f;
  if (art::upload_scene(&sd)) { std::printf("[c_gcore]: %s\n", art_last_error()); return; }
  g.h_nodes.clear(); g.h_tris.clear(); g_h_ready = false;
  g.committed = true;
}

// ---- ray queries.  One GPU launch answers any number of rays, so concurrent callers are combined: the reference calls
// gcore_closest_hit from up to 28 Ada tasks at once (scene_hydra_embree.adb:426-446, Threads_Num ray_tracer.ads:23).  The first caller
// to arrive becomes the leader and launches what is pending; callers arriving meanwhile queue up and ride the NEXT launch together
// (flat combining: no thread is created, nobody spins).  gcore_closest_hit_n is the explicit batch form of the same query.
namespace {

struct Req { const float* pos; const float* dir; float t_near, t_far; HitCpp* out; bool found; bool done; };
std::mutex q_mu;
std::condition_variable q_cv;
std::vector<Req*> q_pending;
bool q_leader = false;

// a hit of the flattened upload -> HitCpp.  prim_index: the uploaded triangle (2k front winding, 2k + 1 back winding of input triangle k)
void fill_flat(int32_t prim_index, float t, float hu, float hv, float tn, HitCpp* pHit) {
  const int32_t k = prim_index >> 1; const bool flipped = (prim_index & 1) != 0;
  pHit->primIndex = g.tri_prim[k];
  pHit->geomIndex = 0;                 // each mesh scene holds a single geometry (embree_connect.cpp:139)
  pHit->instIndex = g.tri_inst[k];
  pHit->t = t + tn;
  const int32_t* ix = &g.widx[6 * (size_t)k];
  const float* A = &g.wverts[3 * (size_t)ix[0]]; const float* B = &g.wverts[3 * (size_t)ix[1]]; const float* C = &g.wverts[3 * (size_t)ix[2]];
  const float e1[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] }, e2[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
  pHit->normal[0] = e1[1] * e2[2] - e1[2] * e2[1];
  pHit->normal[1] = e1[2] * e2[0] - e1[0] * e2[2];
  pHit->normal[2] = e1[0] * e2[1] - e1[1] * e2[0];
  // kernel barycentrics: v = weight of its 2nd vertex, u = weight of its 3rd (geometry.adb:245-246)
  pHit->texCoord[0] = flipped ? hu : hv;   // weight of v1
  pHit->texCoord[1] = flipped ? hv : hu;   // weight of v2
}
void fill_hit(const ArtHit& h, float tn, Req& r) {
  r.found = false;
  if (!h.is_hit || h.prim_type != 2) return;
  fill_flat(h.prim_index, h.t, h.u, h.v, tn, r.out);
  r.found = true;
}
// a hit of the two-level search -> HitCpp
void fill_two_level(const art::InstHit& h, float tn, HitCpp* pHit) {
  const TwoLevel& T = g.two;
  const art::InstRec& R = T.host.inst[(size_t)h.inst];
  const GMesh& gm = g.meshes[(size_t)R.mesh];
  const int32_t tri = h.prim >> 1; const bool flipped = (h.prim & 1) != 0;
  const GInst& gi = g.insts[(size_t)g.tri_inst[(size_t)h.inst]];
  pHit->primIndex = tri; pHit->geomIndex = 0; pHit->instIndex = g.tri_inst[(size_t)h.inst];
  pHit->t = h.t + tn;
  // Ng of the instanced triangle in WORLD space: cross of the transformed edges (what the flattened upload reports)
  float w[3][3];
  for (int c = 0; c < 3; ++c) {
    const float* v = &gm.verts[3 * (size_t)gm.idx[3 * (size_t)tri + c]];
    for (int rr = 0; rr < 3; ++rr) w[c][rr] = gi.m[4 * rr] * v[0] + gi.m[4 * rr + 1] * v[1] + gi.m[4 * rr + 2] * v[2] + gi.m[4 * rr + 3];
  }
  const float e1[3] = {w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2]}, e2[3] = {w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2]};
  pHit->normal[0] = e1[1] * e2[2] - e1[2] * e2[1]; pHit->normal[1] = e1[2] * e2[0] - e1[0] * e2[2]; pHit->normal[2] = e1[0] * e2[1] - e1[1] * e2[0];
  pHit->texCoord[0] = flipped ? h.u : h.v; pHit->texCoord[1] = flipped ? h.v : h.u;
}

// ---- one ray on the calling thread: the published walk of art_isect.h (bvh_closest / instanced_closest compile for the host) over the
// committed tree.  Same boxes (the binary32 packets ARE the dequantised nodes the GPU kernel tests), same Moeller-Trumbore arithmetic
// (-ffp-contract=off, the slab's fma is a real fma on both sides), the hit is a lexicographic minimum: the GPU's answer, bit for bit.
// Two bodies of the same source: one built for CPUs with FMA3 (every x86-64 host of an MI355X has it), one generic (fmaf through libm).
static inline __attribute__((always_inline)) bool host_walk_body(const float* pos, const float* dir, float t_near, float t_far, HitCpp* pHit) {
  if (!g.committed || !pos || !dir || !pHit) return false;
  const float t0 = (t_near > 0.0f) ? t_near : 0.0f;
  const float f = t_far - t0;
  if (!(f > 0.0f)) return false;
  const art::f3 d = art::mk3(dir[0], dir[1], dir[2]);
  const art::f3 o = art::mk3(pos[0] + t0 * dir[0], pos[1] + t0 * dir[1], pos[2] + t0 * dir[2]);
  if (g.two.on) {
    const art::InstHit h = art::instanced_closest(g.two.host.view(), o, d, f);
    if (h.inst < 0) return false;
    fill_two_level(h, t0, pHit);
    return true;
  }
  art::DevScene view;                                            // only these four members are read by bvh_closest
  view.node_width = g.h_width; view.n_tris = g.h_ntris; view.nodes = g.h_nodes.data(); view.tris = g.h_tris.data();
  art::Cand c = art::cand_init(f);
  art::ShadowState sh; sh.shm = -1.0f; sh.far = false; sh.rep = c;
  art::bvh_closest<false>(view, o, d, c, nullptr, sh);
  if (c.key == art::KEY_MISS || (c.key & ~art::KEY_INDEX_MASK) != art::KEY_TRI) return false;
  fill_flat((int32_t)(c.key & art::KEY_INDEX_MASK), c.t, c.u, c.v, t0, pHit);
  return true;
}
__attribute__((target("fma"))) bool host_walk_fma(const float* pos, const float* dir, float t_near, float t_far, HitCpp* pHit) { return host_walk_body(pos, dir, t_near, t_far, pHit); }
bool host_walk_generic(const float* pos, const float* dir, float t_near, float t_far, HitCpp* pHit) { return host_walk_body(pos, dir, t_near, t_far, pHit); }
#if defined(__HIP_DEVICE_COMPILE__)
const bool g_cpu_has_fma = false;                                                             // (the file's device pass: host-only code, never run there)
#else
const bool g_cpu_has_fma = (__builtin_cpu_init(), __builtin_cpu_supports("fma") != 0);      // (a static initialiser may run before libgcc's own constructor)
#endif
int g_force_gpu_single = 0;          // gcore_set_single_ray_on_gpu (tests): 1 = the flat-combined GPU launches of rounds 2-3

// answers reqs[0..n) with ONE launch (art_trace_rays without statistics: no events, no counter read-back)
void run_batch(Req* const* reqs, size_t n) {
  std::lock_guard<std::mutex> lk(art::g_mu);
  std::vector<float> o(3 * n), d(3 * n), far(n), tn(n);
  std::vector<ArtHit> hits(n);
  std::vector<char> live(n, 0);
  size_t m = 0;                                               // rays with a non-empty interval, packed to the front
  std::vector<size_t> who(n);
  for (size_t i = 0; i < n; ++i) {
    Req& r = *reqs[i];
    r.found = false;
    if (!g.committed || !r.pos || !r.dir || !r.out) continue;
    const float t0 = (r.t_near > 0.0f) ? r.t_near : 0.0f;
    const float f = r.t_far - t0;
    if (!(f > 0.0f)) continue;
    for (int a = 0; a < 3; ++a) { o[3 * m + a] = r.pos[a] + t0 * r.dir[a]; d[3 * m + a] = r.dir[a]; }
    far[m] = f; tn[m] = t0; who[m] = i; ++m;
  }
  if (m == 0) return;
  if (g.two.on) {
    TwoLevel& T = g.two;
    if (T.ray_cap < m) {
      if (T.d_rays) (void)hipFree(T.d_rays);
      if (T.d_hits) (void)hipFree(T.d_hits);
      T.d_rays = T.d_hits = nullptr; T.ray_cap = 0;
      const size_t cap = std::max<size_t>(m, 1024);
      if (hipMalloc(&T.d_rays, cap * 7 * sizeof(float)) != hipSuccess || hipMalloc(&T.d_hits, cap * sizeof(art::InstHit)) != hipSuccess) return;
      T.ray_cap = cap;
    }
    float* dr = (float*)T.d_rays;
    hipStream_t st = art::g_devs[0].stream;
    if (hipMemcpyAsync(dr, o.data(), 3 * m * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dr + 3 * T.ray_cap, d.data(), 3 * m * 4, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipMemcpyAsync(dr + 6 * T.ray_cap, far.data(), m * 4, hipMemcpyHostToDevice, st) != hipSuccess) return;
    art::InstScene S;
    S.tlas_nodes = (const float*)T.d_tlas_nodes; S.tlas_tris = (const float*)T.d_tlas_tris; S.blas_nodes = (const float*)T.d_blas_nodes;
    S.blas_tris = (const float*)T.d_blas_tris; S.inst = (const art::InstRec*)T.d_inst; S.n_inst = (int32_t)T.host.inst.size(); S.width = 4;
    art::launch_trace_instanced(st, S, dr, dr + 3 * T.ray_cap, dr + 6 * T.ray_cap, (int)m, (art::InstHit*)T.d_hits);
    std::vector<art::InstHit> ih(m);
    if (hipMemcpyAsync(ih.data(), T.d_hits, m * sizeof(art::InstHit), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return;
    for (size_t k = 0; k < m; ++k) {
      Req& r = *reqs[who[k]];
      const art::InstHit& h = ih[k];
      if (h.inst < 0) continue;
      fill_two_level(h, tn[k], r.out);
      r.found = true;
    }
    return;
  }
  if (art::trace_rays(o.data(), d.data(), far.data(), (int64_t)m, hits.data(), art::TRACE_COOP, nullptr)) return;
  for (size_t k = 0; k < m; ++k) fill_hit(hits[k], tn[k], *reqs[who[k]]);
}

}  // namespace

// 1: single-ray calls ride flat-combined GPU launches as in rounds 2-3 (kept for A/B and for the parity test host == GPU); 0: host walk
void gcore_set_single_ray_on_gpu(int on) { std::unique_lock<std::shared_mutex> wr(g_scene_rw); g_force_gpu_single = on; }

bool gcore_closest_hit(const float a_rayPos[3], const float a_rayDir[3], float t_near, float t_far, HitCpp* pHit) {
  if (!g_force_gpu_single && !g_h_ready.load(std::memory_order_acquire)) {      // first single-ray call since the commit: fetch the tree, once, under the writers' lock
    std::unique_lock<std::shared_mutex> wr(g_scene_rw);
    if (g.committed && !g.two.on && !g_h_ready.load(std::memory_order_relaxed)) {
      std::lock_guard<std::mutex> lk(art::g_mu);
      if (art::fetch_host_bvh(g.h_nodes, g.h_tris, g.h_width, g.h_ntris)) { std::printf("[c_gcore]: %s\n", art_last_error()); return false; }
    }
    g_h_ready.store(true, std::memory_order_release);
  }
  {
    std::shared_lock<std::shared_mutex> rd(g_scene_rw);
    if (!g_force_gpu_single) return g_cpu_has_fma ? host_walk_fma(a_rayPos, a_rayDir, t_near, t_far, pHit) : host_walk_generic(a_rayPos, a_rayDir, t_near, t_far, pHit);
  }
  Req me; me.pos = a_rayPos; me.dir = a_rayDir; me.t_near = t_near; me.t_far = t_far; me.out = pHit; me.found = false; me.done = false;
  std::unique_lock<std::mutex> lk(q_mu);
  q_pending.push_back(&me);
  if (q_leader) {                                             // somebody is launching: wait for the launch that carries my ray
    q_cv.wait(lk, [&] { return me.done; });
    return me.found;
  }
  q_leader = true;
  while (!q_pending.empty()) {
    std::vector<Req*> batch; batch.swap(q_pending);
    lk.unlock();
    run_batch(batch.data(), batch.size());
    lk.lock();
    for (Req* r : batch) r->done = true;
    q_cv.notify_all();
  }
  q_leader = false;
  return me.found;
}

// Batch form of gcore_closest_hit (an extension of the seam: embree_connect.cpp has no counterpart; Embree's own batch entry point
// would be rtcIntersect1M).  positions / directions: 3 floats per ray; t_near / t_far: one value per ray, or NULL for 0 / 1e5 (the
// values scene_hydra_embree.adb:436 passes); hits[i] is written and found[i] set to 1 where ray i has a hit in (t_near, t_far).
// Returns the number of hits.
int gcore_closest_hit_n(int a_rayNum, const float* a_rayPos3f, const float* a_rayDir3f, const float* t_near, const float* t_far,
                        HitCpp* pHits, unsigned char* pFound) {
  if (a_rayNum <= 0 || !a_rayPos3f || !a_rayDir3f || !pHits || !pFound) return 0;
  std::vector<Req> reqs((size_t)a_rayNum);
  std::vector<Req*> ptr((size_t)a_rayNum);
  for (int i = 0; i < a_rayNum; ++i) {
    Req& r = reqs[(size_t)i];
    r.pos = a_rayPos3f + 3 * (size_t)i; r.dir = a_rayDir3f + 3 * (size_t)i;
    r.t_near = t_near ? t_near[i] : 0.0f; r.t_far = t_far ? t_far[i] : 100000.0f;
    r.out = pHits + i; r.found = false; r.done = false;
    ptr[(size_t)i] = &r;
  }
  run_batch(ptr.data(), ptr.size());
  int n = 0;
  for (int i = 0; i < a_rayNum; ++i) { pFound[i] = reqs[(size_t)i].found ? 1 : 0; n += pFound[i]; }
  return n;
}

}  // extern "C"

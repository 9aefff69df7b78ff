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
// gcore_closest_hit is the per-ray compatibility path (SURVEY 8b "per-ray fallback for debugging"); the frame-level
// art_render_pass is the fast path.  Concurrent callers are combined into one launch; gcore_closest_hit_n takes a batch.
// The scene committed here REPLACES the art_* scene of the process (one backend singleton, like g_data in embree_connect.cpp:12-22):
// a process uses either seam at a time.
#include <algorithm>
#include <array>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include <cmath>

#include "art_api_internal.h"

namespace {

// ---- two-level scene (art_instanced.h): used when instancing is real (>= kTwoLevelMinInstances instances, or forced)
constexpr int kTwoLevelMinInstances = 16;
struct TwoLevel {
  bool on = false;
  std::vector<art::InstRec> inst;
  art::Bvh8 tlas;
  std::vector<float> blas_nodes, blas_tris;
  std::vector<int32_t> mesh_node_base, mesh_tri_base, mesh_ntris;
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
} g;

bool invert_3x4(const float m[12], float out[12]) {            // world -> object in binary64, rounded once
  const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g0 = m[8], h = m[9], i = m[10];
  const double det = a * (e * i - f * h) - b * (d * i - f * g0) + c * (d * h - e * g0);
  if (!(std::fabs(det) > 1.0e-300) || !std::isfinite(det)) return false;
  const double r[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det,
                       (f * g0 - d * i) / det, (a * i - c * g0) / det, (c * d - a * f) / det,
                       (d * h - e * g0) / det, (b * g0 - a * h) / det, (a * e - b * d) / det};
  for (int row = 0; row < 3; ++row) {
    for (int k = 0; k < 3; ++k) out[4 * row + k] = (float)r[3 * row + k];
    out[4 * row + 3] = (float)-(r[3 * row] * (double)m[3] + r[3 * row + 1] * (double)m[7] + r[3 * row + 2] * (double)m[11]);
  }
  return true;
}

template <typename T> bool to_device(void** p, const std::vector<T>& v) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (v.empty()) return true;
  if (hipMalloc(p, v.size() * sizeof(T)) != hipSuccess) return false;
  return hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) == hipSuccess;
}

// one tree per mesh (object space, both windings), one tree over the instances' world boxes
bool build_two_level(std::string& err) {
  TwoLevel& T = g.two;
  T.release(); T = TwoLevel();
  art::BvhBuildParams bp; bp.width = 4;
  const size_t nm = g.meshes.size();
  T.mesh_node_base.assign(nm, 0); T.mesh_tri_base.assign(nm, 0); T.mesh_ntris.assign(nm, 0);
  std::vector<std::array<float, 6>> mesh_box(nm);
  for (size_t mi = 0; mi < nm; ++mi) {
    const GMesh& m = g.meshes[mi];
    const size_t nt = m.idx.size() / 3;
    std::vector<float> tri9(18 * nt);
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (size_t t = 0; t < nt; ++t) {
      const float* A = &m.verts[3 * (size_t)m.idx[3 * t]]; const float* B = &m.verts[3 * (size_t)m.idx[3 * t + 1]]; const float* C = &m.verts[3 * (size_t)m.idx[3 * t + 2]];
      float* f = &tri9[18 * t];
      std::memcpy(f, A, 12); std::memcpy(f + 3, B, 12); std::memcpy(f + 6, C, 12);          // record 2t:   front winding
      std::memcpy(f + 9, A, 12); std::memcpy(f + 12, C, 12); std::memcpy(f + 15, B, 12);     // record 2t+1: back winding
      for (const float* P : {A, B, C}) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], P[a]); hi[a] = std::max(hi[a], P[a]); }
    }
    art::Bvh8 b;
    if (!art::build_bvh8(tri9.data(), nullptr, (int32_t)(2 * nt), bp, b, err)) return false;
    T.mesh_node_base[mi] = (int32_t)(T.blas_nodes.size() / art::node_floats(4));
    T.mesh_tri_base[mi] = (int32_t)(T.blas_tris.size() / art::kTriFloats);
    T.mesh_ntris[mi] = b.n_tris;
    T.blas_nodes.insert(T.blas_nodes.end(), b.nodes.begin(), b.nodes.end());
    T.blas_tris.insert(T.blas_tris.end(), b.tris.begin(), b.tris.end());
    mesh_box[mi] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
  }
  // proxies: ONE triangle per instance whose corners span exactly the instance's (padded) world box, prim = instance index
  std::vector<float> proxy9; std::vector<int32_t> proxy_id;
  for (size_t ii = 0; ii < g.insts.size(); ++ii) {
    const GInst& in = g.insts[ii];
    art::InstRec R; std::memset(&R, 0, sizeof R);
    if (!invert_3x4(in.m, R.minv)) { std::printf("[c_gcore]: instance %d has a singular matrix, skipped\n", (int)ii); continue; }
    R.node_base = T.mesh_node_base[in.mesh]; R.tri_base = T.mesh_tri_base[in.mesh]; R.n_tris = T.mesh_ntris[in.mesh]; R.mesh = in.mesh;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    const std::array<float, 6>& mb = mesh_box[in.mesh];
    for (int corner = 0; corner < 8; ++corner) {
      const double x = mb[(corner & 1) ? 3 : 0], y = mb[(corner & 2) ? 4 : 1], z = mb[(corner & 4) ? 5 : 2];
      for (int r = 0; r < 3; ++r) {
        const double w = (double)in.m[4 * r] * x + (double)in.m[4 * r + 1] * y + (double)in.m[4 * r + 2] * z + (double)in.m[4 * r + 3];
        lo[r] = std::min(lo[r], w); hi[r] = std::max(hi[r], w);
      }
    }
    float flo[3], fhi[3];
    for (int r = 0; r < 3; ++r) {        // pad: the ray is taken to object space in binary32, so the world box must not be tight
      const double pad = 1.0e-4 * (hi[r] - lo[r]) + 1.0e-5 * std::max(std::fabs(lo[r]), std::fabs(hi[r])) + 1.0e-6;
      flo[r] = (float)(lo[r] - pad); fhi[r] = (float)(hi[r] + pad);
    }
    const float p[9] = {flo[0], flo[1], flo[2], fhi[0], fhi[1], fhi[2], flo[0], fhi[1], flo[2]};
    proxy9.insert(proxy9.end(), p, p + 9);
    proxy_id.push_back((int32_t)T.inst.size());
    T.inst.push_back(R);
    g.tri_inst.push_back((int32_t)ii);   // instance record -> caller's instance index
  }
  if (T.inst.empty()) { err = "no valid instances"; return false; }
  art::BvhBuildParams tp; tp.width = 4; tp.max_leaf = 1;
  if (!art::build_bvh8(proxy9.data(), proxy_id.data(), (int32_t)proxy_id.size(), tp, T.tlas, err)) return false;
  if (!to_device(&T.d_tlas_nodes, T.tlas.nodes) || !to_device(&T.d_tlas_tris, T.tlas.tris) || !to_device(&T.d_blas_nodes, T.blas_nodes) ||
      !to_device(&T.d_blas_tris, T.blas_tris) || !to_device(&T.d_inst, T.inst)) { err = "device upload of the two-level scene failed"; return false; }
  T.on = true;
  return true;
}

}  // namespace

extern "C" {

void gcore_destroy(void) {
  std::lock_guard<std::mutex> lk(art::g_mu);
  g.two.release();
  g = GState();
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
  std::lock_guard<std::mutex> lk(art::g_mu);
  g.tri_inst.clear(); g.tri_prim.clear(); g.wverts.clear(); g.widx.clear();
  g.two.release(); g.two.on = false;
  const bool two_level = (g_force_two_level == 1) || (g_force_two_level < 0 && (int)g.insts.size() >= kTwoLevelMinInstances);
  if (two_level) {                       // embree_connect.cpp:147-184: one tree per mesh, instances on top
    if (g.insts.empty()) { std::printf("[c_gcore]: gcore_commit_scene, no instances\n"); return; }
    std::string err;
    if (art::ensure_device()) { std::printf("[c_gcore]: %s\n", art_last_error()); return; }
    if (!build_two_level(err)) { std::printf("[c_gcore]: two-level build: %s\n", err.c_str()); return; }
    g.committed = true;
    return;
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

void fill_hit(const ArtHit& h, float tn, Req& r) {
  r.found = false;
  if (!h.is_hit || h.prim_type != 2) return;
  const int32_t k = h.prim_index >> 1; const bool flipped = (h.prim_index & 1) != 0;
  HitCpp* pHit = r.out;
  pHit->primIndex = g.tri_prim[k];
  pHit->geomIndex = 0;                 // each mesh scene holds a single geometry (embree_connect.cpp:139)
  pHit->instIndex = g.tri_inst[k];
  pHit->t = h.t + tn;
  const int32_t* ix = &g.widx[6 * (size_t)k];
  const float* A = &g.wverts[3 * (size_t)ix[0]]; const float* B = &g.wverts[3 * (size_t)ix[1]]; const float* C = &g.wverts[3 * (size_t)ix[2]];
  const float e1[3] = { B[0] - A[0], B[1] - A[1], B[2] - A[2] }, e2[3] = { C[0] - A[0], C[1] - A[1], C[2] - A[2] };
  pHit->normal[0] = e1[1] * e2[2] - e1[2] * e2[1];
  pHit->normal[1] = e1[2] * e2[0] - e1[0] * e2[2];
  pHit->normal[2] = e1[0] * e2[1] - e1[1] * e2[0];
  // kernel barycentrics: v = weight of its 2nd vertex, u = weight of its 3rd (geometry.adb:245-246)
  pHit->texCoord[0] = flipped ? h.u : h.v;   // weight of v1
  pHit->texCoord[1] = flipped ? h.v : h.u;   // weight of v2
  r.found = true;
}

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
    S.blas_tris = (const float*)T.d_blas_tris; S.inst = (const art::InstRec*)T.d_inst; S.n_inst = (int32_t)T.inst.size(); S.width = 4;
    art::launch_trace_instanced(st, S, dr, dr + 3 * T.ray_cap, dr + 6 * T.ray_cap, (int)m, (art::InstHit*)T.d_hits);
    std::vector<art::InstHit> ih(m);
    if (hipMemcpyAsync(ih.data(), T.d_hits, m * sizeof(art::InstHit), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return;
    for (size_t k = 0; k < m; ++k) {
      Req& r = *reqs[who[k]];
      const art::InstHit& h = ih[k];
      if (h.inst < 0) continue;
      const art::InstRec& R = T.inst[(size_t)h.inst];
      const GMesh& gm = g.meshes[(size_t)R.mesh];
      const int32_t tri = h.prim >> 1; const bool flipped = (h.prim & 1) != 0;
      const GInst& gi = g.insts[(size_t)g.tri_inst[(size_t)h.inst]];
      HitCpp* pHit = r.out;
      pHit->primIndex = tri; pHit->geomIndex = 0; pHit->instIndex = g.tri_inst[(size_t)h.inst];
      pHit->t = h.t + tn[k];
      // Ng of the instanced triangle in WORLD space: cross of the transformed edges (what the flattened upload reports)
      float w[3][3];
      for (int c = 0; c < 3; ++c) {
        const float* v = &gm.verts[3 * (size_t)gm.idx[3 * (size_t)tri + c]];
        for (int rr = 0; rr < 3; ++rr) w[c][rr] = gi.m[4 * rr] * v[0] + gi.m[4 * rr + 1] * v[1] + gi.m[4 * rr + 2] * v[2] + gi.m[4 * rr + 3];
      }
      const float e1[3] = {w[1][0] - w[0][0], w[1][1] - w[0][1], w[1][2] - w[0][2]}, e2[3] = {w[2][0] - w[0][0], w[2][1] - w[0][1], w[2][2] - w[0][2]};
      pHit->normal[0] = e1[1] * e2[2] - e1[2] * e2[1]; pHit->normal[1] = e1[2] * e2[0] - e1[0] * e2[2]; pHit->normal[2] = e1[0] * e2[1] - e1[1] * e2[0];
      pHit->texCoord[0] = flipped ? h.u : h.v; pHit->texCoord[1] = flipped ? h.v : h.u;
      r.found = true;
    }
    return;
  }
  if (art::trace_rays(o.data(), d.data(), far.data(), (int64_t)m, hits.data(), art::TRACE_COOP, nullptr)) return;
  for (size_t k = 0; k < m; ++k) fill_hit(hits[k], tn[k], *reqs[who[k]]);
}

}  // namespace

bool gcore_closest_hit(const float a_rayPos[3], const float a_rayDir[3], float t_near, float t_far, HitCpp* pHit) {
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

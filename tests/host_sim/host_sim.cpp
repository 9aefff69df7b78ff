// host_sim.cpp -- TEST-ONLY harness (never linked into libart_hip.so, never shipped): compiles the
// product's host+device per-slot functions (csrc/art_shade.h, art_isect.h, art_math.h) with g++ and
// drives the same wavefront schedule on CPU arrays.  The build container has no GPU, so this is how
// the device logic is checked against the oracle before a GPU run; the GPU parity tests (-m gpu) go
// through the real C ABI.  It is not a fallback: the product has no CPU path.
#include <cstring>
#include <string>
#include <vector>
#include "../../ada-ray-tracer_amd/csrc/art_host_scene.h"
#include "../../ada-ray-tracer_amd/csrc/art_shade.h"
#include "../../ada-ray-tracer_amd/csrc/art_instanced_build.h"

using namespace art;

static std::string g_err;

static BvhBuildParams host_params() { BvhBuildParams p; p.builder = 0; return p; }   // no GPU here: the host SAH builder (the GPU SAH builder builds the same tree)
static BvhBuildParams g_bp = host_params();   // builder parameters for the following hs_* calls (tests of the spatial-split builder)
extern "C" void hs_set_bvh_param(const char* name, double v) {
  const std::string n(name);
  if (n == "spatial_alpha") g_bp.spatial_alpha = (float)v;
  else if (n == "spatial_budget") g_bp.spatial_budget = (float)v;
  else if (n == "spatial_bins") g_bp.spatial_bins = (int)v;
  else if (n == "max_leaf") g_bp.max_leaf = (int)v;
  else if (n == "width") g_bp.width = (int)v;
  else if (n == "quantise") g_bp.quantise = (int)v;
  else if (n == "node_cost") g_bp.node_cost = (float)v;
  else if (n == "leaf_base") g_bp.leaf_base = (float)v;
  else if (n == "tri_cost") g_bp.tri_cost = (float)v;
  else if (n == "sah_bins") g_bp.sah_bins = (int)v;
  else if (n == "collapse") g_bp.collapse = (int)v;
  else if (n == "inst_open") g_bp.inst_open = (int)v;

}

extern "C" const char* hs_last_error() { return g_err.c_str(); }

extern "C" int hs_trace(const ArtSceneDesc* sd, const float* o, const float* d, const float* tfar, long long n,
                        ArtHit* out, unsigned long long* stats4) {
  HostScene hs; BvhBuildParams bp = g_bp;
  if (!flatten_scene(*sd, bp, hs, g_err)) return 1;
  bind_host_pointers(hs);
  BvhStats st = {0, 0, 0, 0};
  for (long long i = 0; i < n; ++i) {
    const f3 oo = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    const Cand c = closest_hit<true>(hs.hdr, oo, dd, tfar ? tfar[i] : kInfinity, &st);
    ArtHit& h = out[i];
    h.t = c.t; h.u = c.u; h.v = c.v;
    if (c.key == KEY_MISS) { h.is_hit = 0; h.prim_type = -1; h.prim_index = -1; h.mat_id = -1; h.mat = -1; h.normal[0] = h.normal[1] = h.normal[2] = 0; continue; }
    const Surface sf = surface_at(hs.hdr, oo, dd, c.t, c.key, c.u, c.v);
    const uint32_t cls = c.key & ~KEY_INDEX_MASK;
    h.is_hit = 1; h.prim_index = (int32_t)(c.key & KEY_INDEX_MASK); h.mat_id = sf.mat_id; h.mat = sf.mat;
    h.prim_type = (cls == KEY_CORNELL) ? 0 : (cls == KEY_SPHERE) ? 1 : (cls == KEY_QUAD) ? 3 : 2;
    h.normal[0] = sf.normal.x; h.normal[1] = sf.normal.y; h.normal[2] = sf.normal.z;
  }
  if (stats4) { stats4[0] = st.box_tests; stats4[1] = st.tri_tests; stats4[2] = st.node_visits; stats4[3] = st.leaf_visits; }
  return 0;
}

static int g_rank = 0, g_nranks = 1, g_tile = 32;
extern "C" void hs_set_shard(int rank, int nranks, int tile) { g_rank = rank; g_nranks = nranks; g_tile = tile; }
// the product's pixel ownership map (art_host_scene.cpp build_pixmap) for one rank: returns the number of pixels, fills out[0 .. cap)
extern "C" long long hs_pixmap(int w, int h, int rank, int nranks, int tile, unsigned* out, long long cap) {
  const std::vector<uint32_t> pm = build_pixmap(w, h, rank, nranks, tile);
  for (size_t i = 0; i < pm.size() && (long long)i < cap; ++i) out[i] = pm[i];
  return (long long)pm.size();
}
// 1: fold over dense per-level records (DevPaths::fold_dense, what the GPU's compacted schedule does; here with the identity layout, where an
// item's index is its slot at every level) instead of the slot-indexed fold stack
static int g_fold_dense = 0;
extern "C" void hs_set_fold_dense(int on) { g_fold_dense = on; }
static int g_skip_null_shadow = 0;
extern "C" void hs_set_skip_null_shadow(int on) { g_skip_null_shadow = on; }

extern "C" int hs_render(const ArtSceneDesc* sd, const ArtPassParams* p, int w, int h, int spp0, float* accum /*row-major*/,
                         unsigned long long* rays_out) {
  HostScene hs; BvhBuildParams bp = g_bp;
  if (!flatten_scene(*sd, bp, hs, g_err)) return 1;
  bind_host_pointers(hs);
  DevFrame F; F.width = w; F.height = h; F.render_type = p->render_type; F.aa_on = p->aa_on ? 1 : 0; F.max_depth = p->max_depth;
  F.seed_lo = (uint32_t)p->seed; F.seed_hi = (uint32_t)(p->seed >> 32); std::memcpy(F.background, p->background, 12);
  F.cam_z = -(float)w / safe_tan(kHalfPi / 2.0f); F.skip_null_shadow = g_skip_null_shadow;
  const std::vector<uint32_t> pixmap = build_pixmap(w, h, g_rank, g_nranks, g_tile);   // the product's shard map
  const int per = p->aa_on ? 4 : 1, S = p->vthreads * per, npix = (int)pixmap.size(), P = npix * S, D = p->max_depth;
  std::vector<float> buf((size_t)(14 + 8 + 3 + 3 + 6 * D + 3 + 3) * P, 0.0f);
  DevPaths q; std::memset(&q, 0, sizeof q);
  q.P = P; q.npix = npix; q.pixmap = pixmap.data(); q.sample_base = (uint32_t)spp0;
  float* f = buf.data(); const size_t pp = (size_t)P;
  auto take = [&](size_t k) { float* r = f; f += k; return r; };
  q.ray_ox = take(2 * pp); q.ray_oy = take(2 * pp); q.ray_oz = take(2 * pp); q.ray_dx = take(2 * pp); q.ray_dy = take(2 * pp); q.ray_dz = take(2 * pp); q.ray_tfar = take(2 * pp);
  std::vector<DevHit> hitbuf(2 * pp); q.hit = hitbuf.data();
  q.prev_pdf = take(pp); q.flags = (uint32_t*)take(pp); q.sh_min_t = take(pp); q.cand_r = take(pp); q.cand_g = take(pp); q.cand_b = take(pp);
  q.e_r = take(D * pp); q.e_g = take(D * pp); q.e_b = take(D * pp); q.w_r = take(D * pp); q.w_g = take(D * pp); q.w_b = take(D * pp);
  q.term_r = take(pp); q.term_g = take(pp); q.term_b = take(pp); q.rad_r = take(pp); q.rad_g = take(pp); q.rad_b = take(pp);
  q.slot_id = nullptr; q.final_flags = q.flags;            // identity layout: one item per slot, updated in place
  std::vector<float> ebuf; std::vector<int32_t> childbuf;
  if (g_fold_dense) {                                      // e needs one level more (e_k sits at level k + 1), plus the child links
    ebuf.assign((size_t)3 * (D + 1) * pp, 0.0f); childbuf.assign((size_t)D * pp, -7);
    q.e_r = ebuf.data(); q.e_g = q.e_r + (D + 1) * pp; q.e_b = q.e_g + (D + 1) * pp;
    q.child = childbuf.data(); q.fold_dense = 1;
  }
  unsigned long long rays = 0;
  auto trace = [&](int nr) {
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : rays)
    for (int i = 0; i < nr; ++i) {
      if (!(q.ray_tfar[i] >= 0.0f)) continue;
      rays++;
      const float shm = (i >= P) ? q.sh_min_t[i - P] : -1.0f;
      const Cand c = closest_hit<false>(hs.hdr, mk3(q.ray_ox[i], q.ray_oy[i], q.ray_oz[i]), mk3(q.ray_dx[i], q.ray_dy[i], q.ray_dz[i]), q.ray_tfar[i], nullptr, shm);
      q.hit[i] = DevHit{c.t, c.key, c.u, c.v};
    }
  };
#pragma omp parallel for
  for (int s = 0; s < P; ++s) raygen_slot(F, hs.hdr, q, s);
  for (int b = 0; b < D; ++b) {
    trace(b == 0 ? P : 2 * P);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int s = 0; s < P; ++s) shade_slot(F, hs.hdr, q, s, b);
  }
  if (p->render_type != ART_PT_STUPID) trace(2 * P);
  if (g_fold_dense) {
#pragma omp parallel for
    for (int s = 0; s < P; ++s) resolve_last_shadow(q, s, D - 1);
    for (int k = D - 1; k >= 0; --k) {                     // art_kernels.hip launch_fold_levels: odd levels -> term, even levels -> rad
      float* cur[3] = {(k & 1) ? q.term_r : q.rad_r, (k & 1) ? q.term_g : q.rad_g, (k & 1) ? q.term_b : q.rad_b};
      const float* nxt[3] = {(k & 1) ? q.rad_r : q.term_r, (k & 1) ? q.rad_g : q.term_g, (k & 1) ? q.rad_b : q.term_b};
#pragma omp parallel for
      for (int s = 0; s < P; ++s) fold_level_item(F, q, k, s, k == D - 1, nxt[0], nxt[1], nxt[2], cur[0], cur[1], cur[2]);
    }
  } else {
#pragma omp parallel for
    for (int s = 0; s < P; ++s) finish_slot(F, q, s, D - 1);
  }
#pragma omp parallel for
  for (int pl = 0; pl < npix; ++pl) accumulate_pixel(F, q, pl, S, accum);
  if (rays_out) *rays_out = rays;
  return 0;
}

// the explicit flattening of an instanced scene's meshes (art_host_scene.cpp flatten_instances): what an instanced render must equal.
// Call with null outputs for the sizes (nv, nt in counts2), then with buffers of 3 nv, 3 nv, 3 nt, nt elements.
extern "C" int hs_flatten_instances(const ArtSceneDesc* sd, float* pos, float* nrm, int32_t* idx, int32_t* matid, long long* counts2) {
  std::vector<float> p, n; std::vector<int32_t> ix, mi;
  if (!flatten_instances(*sd, p, n, ix, mi, g_err)) return 1;
  counts2[0] = (long long)(p.size() / 3); counts2[1] = (long long)mi.size();
  if (pos) std::memcpy(pos, p.data(), p.size() * 4);
  if (nrm) std::memcpy(nrm, n.data(), n.size() * 4);
  if (idx) std::memcpy(idx, ix.data(), ix.size() * 4);
  if (matid) std::memcpy(matid, mi.data(), mi.size() * 4);
  return 0;
}

// BVH of the scene's CLOSEST mesh in the product's packet layout (what art_export_bvh returns on the GPU box)
extern "C" int hs_bvh(const ArtSceneDesc* sd, float* nodes, long long node_cap, float* tris, long long tri_cap, int* info3 /* n_nodes, n_tris, max_stack, width */) {
  HostScene hs; BvhBuildParams bp = g_bp;
  if (!flatten_scene(*sd, bp, hs, g_err)) return 1;
  info3[0] = hs.bvh.n_nodes; info3[1] = hs.bvh.n_tris; info3[2] = hs.bvh.max_stack; info3[3] = hs.bvh.width;
  if (nodes) { if (node_cap < (long long)hs.bvh.nodes.size()) return 2; std::memcpy(nodes, hs.bvh.nodes.data(), hs.bvh.nodes.size() * 4); }
  if (tris) { if (tri_cap < (long long)hs.bvh.tris.size()) return 2; std::memcpy(tris, hs.bvh.tris.data(), hs.bvh.tris.size() * 4); }
  return 0;
}

// ---- per-function known-answer entry points of the PRODUCT's device code (csrc/art_shade.h compiled for the host): the same four
// functions the oracle exports as orc_kat_*, with the uniforms passed in.  tests/test_oracle_kat.py checks both against the
// independent numpy-float32 transcription of lights.adb / materials.adb.
static_assert(sizeof(ArtLight) == sizeof(DevLight) && sizeof(ArtMaterial) == sizeof(DevMaterial), "ABI records are the device records");
extern "C" void hs_kat_light_sample(const ArtLight* l, float u1, float u2, const float p[3], float out10[10]) {
  DevLight d; std::memcpy(&d, l, sizeof d);
  const LightSample r = light_sample(&d, u1, u2, ld3(p));
  out10[0] = r.pos.x; out10[1] = r.pos.y; out10[2] = r.pos.z; out10[3] = r.dir.x; out10[4] = r.dir.y; out10[5] = r.dir.z;
  out10[6] = r.intensity.x; out10[7] = r.intensity.y; out10[8] = r.intensity.z; out10[9] = r.pdf;
}
extern "C" float hs_kat_light_eval_pdf(const ArtLight* l, const float p[3], const float ray_dir[3], float hit_dist) {
  DevLight d; std::memcpy(&d, l, sizeof d);
  return light_eval_pdf(&d, ld3(p), ld3(ray_dir), hit_dist);
}
extern "C" void hs_kat_mat_sample(const ArtMaterial* m, float xi1, float xi2, const float ray_dir[3], const float normal[3], float out8[8]) {
  DevMaterial d; std::memcpy(&d, m, sizeof d);
  const BsdfSample r = bsdf_sample(d, xi1, xi2, ld3(ray_dir), ld3(normal));
  out8[0] = r.color.x; out8[1] = r.color.y; out8[2] = r.color.z; out8[3] = r.dir.x; out8[4] = r.dir.y; out8[5] = r.dir.z;
  out8[6] = r.pdf; out8[7] = r.specular ? 1.0f : 0.0f;
}
extern "C" void hs_kat_mat_eval(const ArtMaterial* m, const float l[3], const float v[3], const float n[3], float out4[4]) {
  DevMaterial d; std::memcpy(&d, m, sizeof d);
  f3 b; float pdf;
  bsdf_eval(d, ld3(l), ld3(v), ld3(n), b, pdf);
  out4[0] = b.x; out4[1] = b.y; out4[2] = b.z; out4[3] = pdf;
}

// ---- two-level (instanced) closest hit of the legacy seam, on the CPU: art_instanced_build.cpp + art_instanced.h, the code the GPU runs
extern "C" int hs_trace_instanced(const float* verts, int n_verts, const int* idx, int n_tris, const float* matrices16, int n_inst,
                                  const float* o, const float* d, int n_rays, float tfar, int* out_inst, int* out_prim, float* out_t, float* out_uv) {
  std::vector<InstMeshIn> meshes = {{verts, (size_t)n_verts, idx, (size_t)n_tris}};
  std::vector<InstIn> insts((size_t)n_inst);
  for (int k = 0; k < n_inst; ++k) { insts[k].mesh = 0; std::memcpy(insts[k].m, matrices16 + 16 * (size_t)k, 12 * sizeof(float)); }
  TwoLevelHost T;
  if (!build_two_level_host(meshes, insts, T, g_err)) return 1;
  const InstScene S = T.view();
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < n_rays; ++i) {
    const InstHit h = instanced_closest(S, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tfar);
    out_inst[i] = h.inst < 0 ? -1 : T.inst_src[(size_t)h.inst]; out_prim[i] = h.prim; out_t[i] = h.t; out_uv[2 * i] = h.u; out_uv[2 * i + 1] = h.v;
  }
  return 0;
}

// art_instanced_build.cpp -- see art_instanced_build.h.  What Embree does in rtcCommitScene for embree_connect.cpp:147-184.
#include "art_instanced_build.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

namespace art {

static bool invert_3x4(const float m[12], float out[12]) {     // world -> object in binary64, rounded once
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

bool build_two_level_host(const std::vector<InstMeshIn>& meshes, const std::vector<InstIn>& insts, TwoLevelHost& T, std::string& err,
                          bool two_sided, float pad_rel, float pad_abs) {
  T = TwoLevelHost();
  BvhBuildParams bp; bp.width = 4;
  if (pad_rel >= 0.0f) bp.inflate_rel = pad_rel;
  if (pad_abs >= 0.0f) bp.inflate_abs = pad_abs;
  const size_t nm = meshes.size();
  std::vector<int32_t> node_base(nm, 0), tri_base(nm, 0), ntris(nm, 0);
  std::vector<std::array<float, 6>> mesh_box(nm);
  for (size_t mi = 0; mi < nm; ++mi) {
    const InstMeshIn& m = meshes[mi];
    if (m.n_tris == 0 || m.n_verts == 0) { err = "empty mesh"; return false; }
    std::vector<float> tri9((two_sided ? 18 : 9) * m.n_tris);
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (size_t t = 0; t < m.n_tris; ++t) {
      const float* A = m.verts + 3 * (size_t)m.idx[3 * t]; const float* B = m.verts + 3 * (size_t)m.idx[3 * t + 1]; const float* C = m.verts + 3 * (size_t)m.idx[3 * t + 2];
      float* f = &tri9[(two_sided ? 18 : 9) * t];
      std::memcpy(f, A, 12); std::memcpy(f + 3, B, 12); std::memcpy(f + 6, C, 12);          // record 2t:   front winding
      if (two_sided) { std::memcpy(f + 9, A, 12); std::memcpy(f + 12, C, 12); std::memcpy(f + 15, B, 12); }     // record 2t+1: back winding
      for (const float* P : {A, B, C}) for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], P[a]); hi[a] = std::max(hi[a], P[a]); }
    }
    Bvh8 b;
    if (!build_bvh8(tri9.data(), nullptr, (int32_t)((two_sided ? 2 : 1) * m.n_tris), bp, b, err)) return false;
    // instanced_closest walks a mesh tree with bvh_closest's private stack of kStackEntries entries and pushes unchecked (like the
    // flattened upload, which art_upload_scene refuses for the same reason)
    if (b.max_stack > kStackEntries) { err = "mesh tree stack bound " + std::to_string(b.max_stack) + " exceeds " + std::to_string(kStackEntries); return false; }
    node_base[mi] = (int32_t)(T.blas_nodes.size() / node_floats(4));
    tri_base[mi] = (int32_t)(T.blas_tris.size() / kTriFloats);
    ntris[mi] = b.n_tris;
    T.blas_nodes.insert(T.blas_nodes.end(), b.nodes.begin(), b.nodes.end());
    T.blas_tris.insert(T.blas_tris.end(), b.tris.begin(), b.tris.end());
    mesh_box[mi] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
  }
  // proxies: ONE triangle per instance whose corners span exactly the instance's (padded) world box, prim = instance record index
  std::vector<float> proxy9; std::vector<int32_t> proxy_id;
  for (size_t ii = 0; ii < insts.size(); ++ii) {
    const InstIn& in = insts[ii];
    if (in.mesh < 0 || (size_t)in.mesh >= nm) { err = "instance of a missing mesh"; return false; }
    InstRec R; std::memset(&R, 0, sizeof R);
    if (!invert_3x4(in.m, R.minv)) {                            // singular matrix: the instance has no volume, nothing can hit it
      if (!two_sided) { err = "instance " + std::to_string(ii) + ": singular matrix"; return false; }
      continue;
    }
    R.node_base = node_base[in.mesh]; R.tri_base = tri_base[in.mesh]; R.n_tris = ntris[in.mesh]; R.mesh = in.mesh;
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
    T.inst_src.push_back((int32_t)ii);
  }
  if (T.inst.empty()) { err = "no valid instances"; return false; }
  BvhBuildParams tp; tp.width = 4; tp.max_leaf = 1;
  if (!build_bvh8(proxy9.data(), proxy_id.data(), (int32_t)proxy_id.size(), tp, T.tlas, err)) return false;
  if (T.tlas.max_stack > kInstTopStack) { err = "instance tree stack bound " + std::to_string(T.tlas.max_stack) + " exceeds " + std::to_string(kInstTopStack); return false; }
  return true;
}

}  // namespace art

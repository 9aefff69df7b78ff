// art_instanced_build.cpp -- see art_instanced_build.h.  What Embree does in rtcCommitScene for embree_connect.cpp:147-184.
#include "art_instanced_build.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <utility>

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
                          bool two_sided, float pad_rel, float pad_abs, int open_factor, float scene_extent) {
  T = TwoLevelHost();
  BvhBuildParams bp; bp.width = 4;
  if (pad_rel >= 0.0f) bp.inflate_rel = pad_rel;
  if (pad_abs >= 0.0f) bp.inflate_abs = pad_abs;
  const size_t nm = meshes.size();
  std::vector<int32_t> node_base(nm, 0), tri_base(nm, 0), ntris(nm, 0);
  std::vector<std::array<float, 6>> mesh_box(nm);
  std::vector<std::vector<uint32_t>> mesh_q;             // one-sided builds: every mesh's quantised nodes, relocated below
  std::vector<std::vector<float>> mesh_pts(nm);          // the vertices a mesh's triangles use (an instance's world box is the box of their images)
  std::vector<const float*> inst_m;                      // the kept instances' matrices (the caller's arrays)
  for (size_t mi = 0; mi < nm; ++mi) {                   // the meshes' object-space boxes first: the instances' pads below need them
    const InstMeshIn& m = meshes[mi];
    if (m.n_tris == 0 || m.n_verts == 0) { err = "empty mesh"; return false; }
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (size_t k = 0; k < 3 * m.n_tris; ++k) { const float* P = m.verts + 3 * (size_t)m.idx[k]; for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], P[a]); hi[a] = std::max(hi[a], P[a]); } }
    mesh_box[mi] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]};
  }
  // ---- the absolute pad of a mesh's boxes follows its instances (round 6).  The walk tests a mesh's boxes with the ray taken to
  // object space in binary32: oo = minv * o, dd = minv * d, each coordinate a sum of four / three rounded terms.  Its error grows with
  // |minv| * |o|, not with the object-space coordinates the relative pad follows: a speck -- scale 5e-4, object coordinates within +-1,
  // placed at Cornell-box distances -- sees its ray 2000 * 5 * 2^-24 = 6e-4 object units off, five times the fixed pad of 1.1e-4, and the
  // box test culled triangles the world-space triangle test accepts (holes the flattened scene does not have:
  // tests/test_instanced_host_sim.py test_a_speck_far_from_the_origin).  Bound per object-space coordinate r along a ray of the scene:
  //   | err | <= k * 2^-24 * ( sum_j |minv_rj| * (E + L) + |minv_r3| ),   E = the largest |coordinate| a ray can start at, L = the longest ray
  // (a ray's point at parameter t is oo + t * dd: the origin's error plus t times the direction's).  With k = 8 (four terms, their products
  // and the running sum each rounded, a factor two to spare), E = the scene's extent and L = 2 E.  The pad of a mesh is the maximum over
  // its instances and the three coordinates, never below the caller's pad_abs.
  std::vector<double> mesh_pad(nm, 0.0);
  {
    double E = (double)std::max(0.0f, scene_extent);
    std::vector<std::array<float, 12>> minv(insts.size());
    std::vector<uint8_t> ok(insts.size(), 0);
    for (size_t ii = 0; ii < insts.size(); ++ii) {
      const InstIn& in = insts[ii];
      if (in.mesh < 0 || (size_t)in.mesh >= nm) continue;                       // (reported below)
      ok[ii] = invert_3x4(in.m, minv[ii].data()) ? 1 : 0;
      const std::array<float, 6>& mb = mesh_box[(size_t)in.mesh];
      for (int r = 0; r < 3; ++r) {                                             // how far out the instance reaches in the world
        double reach = std::fabs((double)in.m[4 * r + 3]);
        for (int j = 0; j < 3; ++j) reach += std::fabs((double)in.m[4 * r + j]) * std::max(std::fabs((double)mb[(size_t)j]), std::fabs((double)mb[(size_t)j + 3]));
        if (std::isfinite(reach)) E = std::max(E, reach);
      }
    }
    for (size_t ii = 0; ii < insts.size(); ++ii) {
      if (!ok[ii]) continue;
      for (int r = 0; r < 3; ++r) {
        const float* q = &minv[ii][4 * (size_t)r];
        const double bound = 8.0 * 5.9604644775390625e-8 * ((std::fabs((double)q[0]) + std::fabs((double)q[1]) + std::fabs((double)q[2])) * 3.0 * E + std::fabs((double)q[3]));
        if (std::isfinite(bound)) mesh_pad[(size_t)insts[ii].mesh] = std::max(mesh_pad[(size_t)insts[ii].mesh], bound);
      }
    }
  }
  for (size_t mi = 0; mi < nm; ++mi) {
    const InstMeshIn& m = meshes[mi];
    std::vector<float> tri9((two_sided ? 18 : 9) * m.n_tris);
    for (size_t t = 0; t < m.n_tris; ++t) {
      const float* A = m.verts + 3 * (size_t)m.idx[3 * t]; const float* B = m.verts + 3 * (size_t)m.idx[3 * t + 1]; const float* C = m.verts + 3 * (size_t)m.idx[3 * t + 2];
      float* f = &tri9[(two_sided ? 18 : 9) * t];
      std::memcpy(f, A, 12); std::memcpy(f + 3, B, 12); std::memcpy(f + 6, C, 12);          // record 2t:   front winding
      if (two_sided) { std::memcpy(f + 9, A, 12); std::memcpy(f + 12, C, 12); std::memcpy(f + 15, B, 12); }     // record 2t+1: back winding
    }
    Bvh8 b;
    BvhBuildParams bm = bp;
    bm.inflate_abs = std::max(bp.inflate_abs, (float)std::min(mesh_pad[mi], 1.0e30));
    T.mesh_pad_abs.push_back(bm.inflate_abs);
    if (!build_bvh8(tri9.data(), nullptr, (int32_t)((two_sided ? 2 : 1) * m.n_tris), bm, b, err)) return false;
    // instanced_closest walks a mesh tree with bvh_closest's private stack of kStackEntries entries and pushes unchecked (like the
    // flattened upload, which art_upload_scene refuses for the same reason)
    if (b.max_stack > kStackEntries) { err = "mesh tree stack bound " + std::to_string(b.max_stack) + " exceeds " + std::to_string(kStackEntries); return false; }
    if (!two_sided) mesh_q.push_back(b.qnodes);
    T.blas_max_stack = std::max(T.blas_max_stack, b.max_stack);
    node_base[mi] = (int32_t)(T.blas_nodes.size() / node_floats(4));
    tri_base[mi] = (int32_t)(T.blas_tris.size() / kTriFloats);
    ntris[mi] = b.n_tris;
    T.blas_nodes.insert(T.blas_nodes.end(), b.nodes.begin(), b.nodes.end());
    T.blas_tris.insert(T.blas_tris.end(), b.tris.begin(), b.tris.end());
    std::vector<uint8_t> used(m.n_verts, 0);
    for (size_t k = 0; k < 3 * m.n_tris; ++k) used[(size_t)m.idx[k]] = 1;
    for (size_t v = 0; v < m.n_verts; ++v) if (used[v]) mesh_pts[mi].insert(mesh_pts[mi].end(), m.verts + 3 * v, m.verts + 3 * v + 3);
  }
  // An instance's world box: the box of its transformed VERTICES (tight: the image of the object box's 8 corners is up to twice as large
  // for a rotated instance, and every overlap of two instances' boxes is a second mesh tree entered for nothing) while that stays cheap --
  // 2e8 vertex transforms per build, about half a second; beyond that the corners' box.
  double transforms = 0.0;
  for (const InstIn& in : insts) if (in.mesh >= 0 && (size_t)in.mesh < nm) transforms += (double)mesh_pts[(size_t)in.mesh].size() / 3.0;
  const bool tight = transforms <= 2.0e8;
  // ---- instances, and the entry points the instance tree will end at (art_scene.h DevInstance).  One per instance to begin with: the
  // mesh's root under the box of the whole instance.  One-sided builds with open_factor > 1 then OPEN entry points, largest world box
  // first, into the children of their node -- each child under the tight world box of ITS triangles -- until there are open_factor x
  // instances of them (or nothing is left to open, or the transform budget is spent).
  struct Open { double area; int32_t inst, entry; float lo[3], hi[3]; };
  std::vector<Open> heap, closed;
  auto world_box = [&](const float m[12], auto&& each_point, Open& o) {     // box of the transformed points, padded (see below)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    double mag[3] = {0.0, 0.0, 0.0};                    // the largest sum of |terms| of a coordinate: what the binary32 transform of the kernels rounds against
    const double M[12] = {m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11]};
    each_point([&](double x, double y, double z) {
      for (int r = 0; r < 3; ++r) {
        const double a = M[4 * r] * x, b = M[4 * r + 1] * y, c = M[4 * r + 2] * z, w = a + b + c + M[4 * r + 3];
        lo[r] = std::min(lo[r], w); hi[r] = std::max(hi[r], w);
        mag[r] = std::max(mag[r], std::fabs(a) + std::fabs(b) + std::fabs(c) + std::fabs(M[4 * r + 3]));
      }
    });
    for (int r = 0; r < 3; ++r) {        // pad: the ray is taken to object space in binary32, so the world box must not be tight
      const double pad = 1.0e-4 * (hi[r] - lo[r]) + 1.0e-5 * std::max(std::fabs(lo[r]), std::fabs(hi[r])) + 1.0e-6 * mag[r] + 1.0e-6;
      o.lo[r] = (float)(lo[r] - pad); o.hi[r] = (float)(hi[r] + pad);
    }
    const double ex = (double)o.hi[0] - o.lo[0], ey = (double)o.hi[1] - o.lo[1], ez = (double)o.hi[2] - o.lo[2];
    o.area = ex * ey + ey * ez + ez * ex;
  };
  for (size_t ii = 0; ii < insts.size(); ++ii) {
    const InstIn& in = insts[ii];
    if (in.mesh < 0 || (size_t)in.mesh >= nm) { err = "instance of a missing mesh"; return false; }
    InstRec R; std::memset(&R, 0, sizeof R);
    if (!invert_3x4(in.m, R.minv)) {                            // singular matrix: the instance has no volume, nothing can hit it
      if (!two_sided) { err = "instance " + std::to_string(ii) + ": singular matrix"; return false; }
      continue;
    }
    R.node_base = node_base[in.mesh]; R.tri_base = tri_base[in.mesh]; R.n_tris = ntris[in.mesh]; R.mesh = in.mesh;
    const std::array<float, 6>& mb = mesh_box[in.mesh];
    const std::vector<float>& pts = mesh_pts[(size_t)in.mesh];
    Open o; o.inst = (int32_t)T.inst.size(); o.entry = 0;
    world_box(in.m, [&](auto&& take) {
      if (tight) for (size_t v = 0; v + 2 < pts.size(); v += 3) take(pts[v], pts[v + 1], pts[v + 2]);
      else for (int corner = 0; corner < 8; ++corner) take(mb[(corner & 1) ? 3 : 0], mb[(corner & 2) ? 4 : 1], mb[(corner & 4) ? 5 : 2]);
    }, o);
    heap.push_back(o);
    T.inst.push_back(R);
    T.inst_src.push_back((int32_t)ii);
    inst_m.push_back(in.m);
  }
  if (T.inst.empty()) { err = "no valid instances"; return false; }
  if (!two_sided && open_factor == 0) {
    // automatic: rho = the instances' box areas over the area of the box around all of them = how many instance boxes a random line
    // through the scene crosses.  Opening pays where instances interpenetrate (a ray inside three or four boxes at once cannot use the
    // hit it found in one mesh to skip the others before it has entered them) and costs where they do not (a ray enters the same
    // instance two or three times).  Measured (profiles/r5_instanced.json): I64, rho = 1.7: 3478 Mrays/s at whole instances, 3182 / 2961
    // at 4 / 64 entry points per instance; its 64 instances pulled into one cluster, rho = 13.6: 2915 at whole instances, 3419 / 3960.
    double sum = 0.0; float ulo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, uhi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (const Open& o : heap) { sum += o.area; for (int r = 0; r < 3; ++r) { ulo[r] = std::min(ulo[r], o.lo[r]); uhi[r] = std::max(uhi[r], o.hi[r]); } }
    const double ex = (double)uhi[0] - ulo[0], ey = (double)uhi[1] - ulo[1], ez = (double)uhi[2] - ulo[2];
    const double rho = sum / std::max(ex * ey + ey * ez + ez * ex, 1.0e-30);
    open_factor = (rho > 4.0) ? (int)std::min(64.0, 4.0 * rho) : 1;
  }
  T.open_factor = std::max(1, open_factor);
  if (!two_sided && open_factor > 1 && tight) {
    auto by_area = [](const Open& a, const Open& b) { return a.area < b.area || (a.area == b.area && (a.inst > b.inst || (a.inst == b.inst && a.entry > b.entry))); };
    std::make_heap(heap.begin(), heap.end(), by_area);
    // (at most 2^20 entry points, or one per instance if those are more: the table is 128 bytes per entry point and the instance tree is built on the host)
    const size_t target = std::min<size_t>((size_t)open_factor * T.inst.size(), std::max<size_t>(T.inst.size(), (size_t)1 << 20));
    std::vector<int32_t> walk;
    while (!heap.empty() && heap.size() + closed.size() < target) {
      std::pop_heap(heap.begin(), heap.end(), by_area);
      const Open o = heap.back(); heap.pop_back();
      const InstRec& R = T.inst[(size_t)o.inst];
      if ((o.entry & 15) != 0 || transforms > 2.0e8) { closed.push_back(o); continue; }      // a leaf, or the budget is spent: stays as it is
      const float* nd = &T.blas_nodes[((size_t)R.node_base + (size_t)(o.entry >> 4)) * (size_t)node_floats(4)];
      for (int j = 0; j < 4; ++j) {
        const int32_t rj = __builtin_bit_cast(int32_t, nd[4 * j + 3]);
        if (rj < 0) continue;
        Open c; c.inst = o.inst; c.entry = (rj << 4) | __builtin_bit_cast(int32_t, nd[16 + 4 * j + 3]);
        world_box(inst_m[(size_t)o.inst], [&](auto&& take) {       // the corners of every triangle below the child
          walk.assign(1, c.entry);
          while (!walk.empty()) {
            const int32_t e = walk.back(); walk.pop_back();
            if (e & 15) {
              for (int q = 0; q < (e & 15); ++q) {
                const float* tr = &T.blas_tris[((size_t)R.tri_base + (size_t)(e >> 4) + (size_t)q) * kTriFloats];
                take(tr[0], tr[1], tr[2]); take(tr[3], tr[4], tr[5]); take(tr[6], tr[7], tr[8]);
                transforms += 3.0;
              }
            } else {
              const float* n2 = &T.blas_nodes[((size_t)R.node_base + (size_t)(e >> 4)) * (size_t)node_floats(4)];
              for (int k = 0; k < 4; ++k) {
                const int32_t rk = __builtin_bit_cast(int32_t, n2[4 * k + 3]);
                if (rk >= 0) walk.push_back((rk << 4) | __builtin_bit_cast(int32_t, n2[16 + 4 * k + 3]));
              }
            }
          }
        }, c);
        heap.push_back(c); std::push_heap(heap.begin(), heap.end(), by_area);
      }
    }
  }
  closed.insert(closed.end(), heap.begin(), heap.end());
  std::stable_sort(closed.begin(), closed.end(), [](const Open& a, const Open& b) { return a.inst < b.inst || (a.inst == b.inst && a.entry < b.entry); });
  // entry point numbering: instance i's first one is entry i, the others follow behind the instances
  T.entry.assign(T.inst.size(), TwoLevelHost::EntryPoint{-1, 0, 0u});
  std::vector<float> proxy9; std::vector<int32_t> proxy_id;
  for (const Open& o : closed) {
    int32_t id = o.inst;
    if (T.entry[(size_t)o.inst].inst >= 0) { id = (int32_t)T.entry.size(); T.entry.push_back(TwoLevelHost::EntryPoint{-1, 0, 0u}); }
    T.entry[(size_t)id] = TwoLevelHost::EntryPoint{o.inst, o.entry, 0u};
    // proxy: ONE triangle whose corners span exactly the entry point's (padded) world box, prim = the entry point
    const float p[9] = {o.lo[0], o.lo[1], o.lo[2], o.hi[0], o.hi[1], o.hi[2], o.lo[0], o.hi[1], o.lo[2]};
    proxy9.insert(proxy9.end(), p, p + 9);
    proxy_id.push_back(id);
  }
  BvhBuildParams tp; tp.width = 4; tp.max_leaf = 1;
  if (!build_bvh8(proxy9.data(), proxy_id.data(), (int32_t)proxy_id.size(), tp, T.tlas, err)) return false;
  if (T.tlas.max_stack > kInstTopStack) { err = "instance tree stack bound " + std::to_string(T.tlas.max_stack) + " exceeds " + std::to_string(kInstTopStack); return false; }
  if (!two_sided) {
    // ---- the cooperative kernel's form: ONE array of quantised nodes, entry words absolute
    if (T.tlas.qnodes.empty()) { err = "instance tree has no quantised nodes"; return false; }
    const size_t words = kQNodeBytes / 4;
    size_t total_nodes = (size_t)T.tlas.n_nodes;
    T.qnode_base.assign(nm, 0);
    for (size_t mi = 0; mi < nm; ++mi) { T.qnode_base[mi] = (int32_t)total_nodes; total_nodes += mesh_q[mi].size() / words; }
    if (total_nodes * kQNodeBytes >= (1ull << 31) || T.blas_tris.size() / kTriFloats * (size_t)kQTriBytes >= (1ull << 31)) { err = "instanced scene too large for 31-bit node / triangle offsets"; return false; }
    T.qnodes.assign(total_nodes * words, 0u);
    std::memcpy(T.qnodes.data(), T.tlas.qnodes.data(), T.tlas.qnodes.size() * 4);
    for (TwoLevelHost::EntryPoint& E : T.entry) {         // where an entry point enters, as the kernel's entry word (checked: it is followed blindly)
      const InstRec& R = T.inst[(size_t)E.inst];
      const size_t ref = (size_t)(E.root_entry >> 4), cnt = (size_t)(E.root_entry & 15);
      const size_t mesh_nodes = mesh_q[(size_t)R.mesh].size() / words;
      if (cnt ? (cnt > 4 || ref + cnt > (size_t)ntris[(size_t)R.mesh]) : (ref >= mesh_nodes)) { err = "internal: entry point outside its mesh's tree"; return false; }
      E.qroot = cnt ? (kQEntryLeaf | (uint32_t)(((size_t)R.tri_base + ref) * (size_t)kQTriBytes) | (uint32_t)cnt)
                    : (uint32_t)(((size_t)T.qnode_base[(size_t)R.mesh] + ref) * (size_t)kQNodeBytes);
    }
    for (int32_t n = 0; n < T.tlas.n_nodes; ++n)
      for (int j = 0; j < 4; ++j) {
        uint32_t& e = T.qnodes[(size_t)n * words + 4 * (size_t)j + 2];
        if (e == kQEntryEmpty || !(e & kQEntryLeaf)) continue;                   // inner entries: the instance tree sits at node 0
        const uint32_t tri = (e & 0x7ffffff0u) / (uint32_t)kQTriBytes;            // (max_leaf = 1: one proxy per leaf)
        const int32_t inst = __builtin_bit_cast(int32_t, T.tlas.tris[(size_t)tri * kTriFloats + 9]);
        e = kQEntryLeaf | ((uint32_t)inst << 4) | kQCountInstance;
      }
    for (size_t mi = 0; mi < nm; ++mi) {
      const std::vector<uint32_t>& q = mesh_q[mi];
      uint32_t* dst = &T.qnodes[(size_t)T.qnode_base[mi] * words];
      std::memcpy(dst, q.data(), q.size() * 4);
      for (size_t n = 0; n < q.size() / words; ++n)
        for (int j = 0; j < 4; ++j) {
          uint32_t& e = dst[n * words + 4 * (size_t)j + 2];
          if (e == kQEntryEmpty) continue;
          if (e & kQEntryLeaf) e = kQEntryLeaf | ((e & 0x7ffffff0u) + (uint32_t)tri_base[mi] * (uint32_t)kQTriBytes) | (e & 15u);
          else e = e + (uint32_t)T.qnode_base[mi] * (uint32_t)kQNodeBytes;
        }
    }
    // The cooperative kernel follows these entry words without a check: a word that led back into a tree already entered would make a
    // ray walk for ever (a GPU hang, not a wrong pixel).  So the array is walked here once, as the kernel would: every node reached exactly
    // once, the instance tree's nodes only from the instance tree, a mesh's nodes only from that mesh's root, every leaf inside its
    // mesh's triangle records, every instance marker naming a valid instance.
    std::vector<uint8_t> seen(total_nodes, 0);
    std::vector<std::pair<uint32_t, int32_t>> todo;      // (node, owner: -1 instance tree, else mesh)
    todo.emplace_back(0u, -1);
    for (size_t mi = 0; mi < nm; ++mi) todo.emplace_back((uint32_t)T.qnode_base[mi], (int32_t)mi);
    while (!todo.empty()) {
      const auto [n, owner] = todo.back(); todo.pop_back();
      const size_t lo = owner < 0 ? 0 : (size_t)T.qnode_base[(size_t)owner];
      const size_t hi = owner < 0 ? (size_t)T.tlas.n_nodes : (((size_t)owner + 1 < nm) ? (size_t)T.qnode_base[(size_t)owner + 1] : total_nodes);
      if (n < lo || n >= hi || seen[n]) { err = "internal: two-level node array is not a forest (node " + std::to_string(n) + ")"; return false; }
      seen[n] = 1;
      for (int j = 0; j < 4; ++j) {
        const uint32_t e = T.qnodes[(size_t)n * words + 4 * (size_t)j + 2];
        if (e == kQEntryEmpty) continue;
        if (e & kQEntryLeaf) {
          const uint32_t cnt = e & 15u, off = e & 0x7ffffff0u;
          if (owner < 0) { if (cnt != kQCountInstance || (off >> 4) >= T.entry.size()) { err = "internal: bad instance marker in the instance tree"; return false; } }
          else {
            const size_t first = off / (size_t)kQTriBytes, base = (size_t)tri_base[(size_t)owner];
            if (cnt < 1 || cnt > 4 || (off % kQTriBytes) != 0 || first < base || first + cnt > base + (size_t)ntris[(size_t)owner]) { err = "internal: bad leaf in a mesh's tree"; return false; }
          }
        } else {
          if (e % kQNodeBytes) { err = "internal: misaligned node entry"; return false; }
          todo.emplace_back(e / (uint32_t)kQNodeBytes, owner);
        }
      }
    }
  }
  return true;
}

}  // namespace art

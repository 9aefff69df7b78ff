// art_bvh.cpp -- host-side BVH construction for the closest-hit mesh.  Replaces Embree's rtcCommitScene
// behind gcore_commit_scene (embree_connect.cpp:241-244): binned-SAH BVH2 (built with a small thread
// pool), collapsed by surface area into 8-wide nodes laid out as the 256-byte lane packets described
// in art_scene.h.  Boxes are inflated by a few ulps so that the kernel's slab test can never cull a
// triangle that the reference's Moeller-Trumbore arithmetic would accept.
#include "art_bvh.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <future>
#include <numeric>

namespace art {
namespace {

struct Box {
  float lo[3], hi[3];
  void reset() { for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
  void grow(const float* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
  void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
  float half_area() const {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return (dx < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
  }
};

struct Node2 {            // BVH2 node; leaf when count > 0
  Box box;
  int32_t left = -1, right = -1;
  int32_t first = 0, count = 0;
};

struct Builder {
  const float* tri;       // 9 floats per triangle
  int32_t n;
  std::vector<Box> tbox;
  std::vector<float> cen; // 3 per triangle
  std::vector<int32_t> order;
  std::vector<Node2> nodes;
  std::atomic<int32_t> next_node{0};
  BvhBuildParams prm;

  int32_t alloc_node() { return next_node.fetch_add(1); }

  // build subtree over order[first, first+count); returns node index
  int32_t build(int32_t first, int32_t count, int depth, int par_depth) {
    const int32_t id = alloc_node();
    Node2 nd;
    nd.box.reset();
    Box cb; cb.reset();
    for (int32_t i = first; i < first + count; ++i) {
      nd.box.grow(tbox[order[i]]);
      cb.grow(&cen[3 * (size_t)order[i]]);
    }
    auto make_leaf = [&]() { nd.first = first; nd.count = count; nodes[id] = nd; return id; };
    if (count <= 1) return make_leaf();

    constexpr int NB = 32;
    float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
    const float parent_area = nd.box.half_area();
    for (int a = 0; a < 3; ++a) {
      const float ext = cb.hi[a] - cb.lo[a];
      if (!(ext > 0.0f)) continue;
      Box bb[NB]; int32_t bc[NB];
      for (int b = 0; b < NB; ++b) { bb[b].reset(); bc[b] = 0; }
      const float scale = (float)NB / ext;
      for (int32_t i = first; i < first + count; ++i) {
        const int32_t t = order[i];
        int b = (int)((cen[3 * (size_t)t + a] - cb.lo[a]) * scale);
        b = std::min(std::max(b, 0), NB - 1);
        bb[b].grow(tbox[t]); bc[b]++;
      }
      float la[NB]; int32_t lc[NB];
      Box acc; acc.reset(); int32_t c = 0;
      for (int b = 0; b < NB - 1; ++b) { acc.grow(bb[b]); c += bc[b]; la[b] = acc.half_area(); lc[b] = c; }
      acc.reset(); c = 0;
      for (int b = NB - 1; b >= 1; --b) {
        acc.grow(bb[b]); c += bc[b];
        if (lc[b - 1] == 0 || c == 0) continue;
        const float cost = la[b - 1] * prm.leaf_cost((int)lc[b - 1]) + acc.half_area() * prm.leaf_cost((int)c);
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
      }
    }
    const bool can_leaf = count <= prm.max_leaf;
    if (best_axis >= 0 && can_leaf) {
      const float split_cost = prm.node_cost * parent_area + best_cost;
      if (!(split_cost < parent_area * prm.leaf_cost(count))) return make_leaf();
    }
    int32_t mid;
    if (best_axis < 0 || depth > prm.max_sah_depth) {
      if (can_leaf && best_axis < 0) return make_leaf();
      // degenerate centroids or depth guard: median split by index on the widest box axis
      int a = 0;
      for (int k = 1; k < 3; ++k) if (nd.box.hi[k] - nd.box.lo[k] > nd.box.hi[a] - nd.box.lo[a]) a = k;
      mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                       [&](int32_t x, int32_t y) { return cen[3 * (size_t)x + a] < cen[3 * (size_t)y + a] || (cen[3 * (size_t)x + a] == cen[3 * (size_t)y + a] && x < y); });
    } else {
      const float ext = cb.hi[best_axis] - cb.lo[best_axis];
      const float scale = (float)NB / ext;
      auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int32_t t) {
        int b = (int)((cen[3 * (size_t)t + best_axis] - cb.lo[best_axis]) * scale);
        b = std::min(std::max(b, 0), NB - 1);
        return b < best_bin;
      });
      mid = (int32_t)(it - order.begin());
      if (mid == first || mid == first + count) mid = first + count / 2;
    }
    const int32_t nl = mid - first, nr = count - nl;
    if (par_depth > 0 && count > 20000) {
      auto fut = std::async(std::launch::async, [&, first, nl, depth, par_depth]() { return build(first, nl, depth + 1, par_depth - 1); });
      nd.right = build(mid, nr, depth + 1, par_depth - 1);
      nd.left = fut.get();
    } else {
      nd.left = build(first, nl, depth + 1, 0);
      nd.right = build(mid, nr, depth + 1, 0);
    }
    nodes[id] = nd;
    return id;
  }
};

inline float next_down(float v) { return std::nextafterf(v, -INFINITY); }
inline float next_up(float v) { return std::nextafterf(v, INFINITY); }

}  // namespace

bool build_bvh8(const float* tri9, const int32_t* prim_ids, int32_t n, const BvhBuildParams& prm, Bvh8& out, std::string& err) {
  out = Bvh8();
  if (n <= 0) return true;
  Builder B;
  B.tri = tri9; B.n = n; B.prm = prm;
  B.tbox.resize(n); B.cen.resize(3 * (size_t)n); B.order.resize(n);
  for (int32_t i = 0; i < n; ++i) {
    Box b; b.reset();
    const float* t = tri9 + 9 * (size_t)i;
    b.grow(t); b.grow(t + 3); b.grow(t + 6);
    for (int a = 0; a < 3; ++a) {
      if (!std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) { err = "non-finite triangle vertex"; return false; }
      B.cen[3 * (size_t)i + a] = 0.5f * b.lo[a] + 0.5f * b.hi[a];
    }
    B.tbox[i] = b;
  }
  std::iota(B.order.begin(), B.order.end(), 0);
  B.nodes.resize(2 * (size_t)n);
  const int32_t root = B.build(0, n, 0, prm.parallel_depth);
  (void)root;

  // ---- collapse to 8-wide
  std::vector<float>& N = out.nodes;
  auto new_node8 = [&]() { const int32_t id = (int32_t)(N.size() / kNodeFloats); N.resize(N.size() + kNodeFloats, 0.0f); return id; };
  out.tris.reserve((size_t)n * kTriFloats);
  const float inflate_rel = prm.inflate_rel, inflate_abs = prm.inflate_abs;

  auto emit_leaf_tris = [&](const Node2& lf) -> int32_t {
    const int32_t first = (int32_t)(out.tris.size() / kTriFloats);
    // triangles inside a leaf are stored in ascending prim order (deterministic, and the order the leaf is tested in)
    std::vector<int32_t> ids(B.order.begin() + lf.first, B.order.begin() + lf.first + lf.count);
    std::sort(ids.begin(), ids.end());
    for (int32_t t : ids) {
      float rec[kTriFloats];
      std::memcpy(rec, tri9 + 9 * (size_t)t, 9 * sizeof(float));
      const int32_t pid = prim_ids ? prim_ids[t] : t;
      std::memcpy(&rec[9], &pid, 4); rec[10] = 0.0f; rec[11] = 0.0f;
      out.tris.insert(out.tris.end(), rec, rec + kTriFloats);
    }
    return first;
  };

  const int32_t root8 = new_node8();   // a single-leaf tree still gets a root node with one child

  int32_t max_stack = 1;
  struct Pending { int32_t n2, n8, stack_before; };
  std::vector<Pending> todo; todo.push_back({0, root8, 0});
  while (!todo.empty()) {
    const Pending p = todo.back(); todo.pop_back();
    // gather up to 8 children by repeatedly opening the inner child with the largest area
    std::vector<int32_t> ch;
    const Node2& top = B.nodes[p.n2];
    if (top.count > 0) ch.push_back(p.n2);
    else { ch.push_back(top.left); ch.push_back(top.right); }
    while ((int)ch.size() < 8) {
      int best = -1; float best_a = -1.0f;
      for (int i = 0; i < (int)ch.size(); ++i) {
        const Node2& c = B.nodes[ch[i]];
        if (c.count > 0) continue;
        const float a = c.box.half_area();
        if (a > best_a) { best_a = a; best = i; }
      }
      if (best < 0) break;
      const Node2 c = B.nodes[ch[best]];
      ch[best] = c.left; ch.push_back(c.right);
    }
    // a BVH2 leaf with more than kMaxLeafTris triangles cannot occur (max_leaf <= 8 is enforced below)
    float* nd = &N[(size_t)p.n8 * kNodeFloats];
    const int nch = (int)ch.size();
    const int32_t stack_here = p.stack_before + (nch - 1);
    max_stack = std::max(max_stack, stack_here + 1);
    for (int j = 0; j < 8; ++j) {
      int32_t ref = -1, cnt = 0;
      float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
      if (j < nch) {
        const Node2& c = B.nodes[ch[j]];
        for (int a = 0; a < 3; ++a) {
          const float m = std::max(std::fabs(c.box.lo[a]), std::fabs(c.box.hi[a]));
          const float pad = inflate_abs + inflate_rel * m;
          lo[a] = next_down(c.box.lo[a] - pad); hi[a] = next_up(c.box.hi[a] + pad);
        }
        if (c.count > 0) { ref = emit_leaf_tris(c); cnt = c.count; }
        else {
          ref = new_node8(); cnt = 0;
          nd = &N[(size_t)p.n8 * kNodeFloats];      // N may have been reallocated
          todo.push_back({ch[j], ref, stack_here});
        }
      }
      nd[4 * j + 0] = lo[0]; nd[4 * j + 1] = lo[1]; nd[4 * j + 2] = lo[2]; std::memcpy(&nd[4 * j + 3], &ref, 4);
      nd[32 + 4 * j + 0] = hi[0]; nd[32 + 4 * j + 1] = hi[1]; nd[32 + 4 * j + 2] = hi[2]; std::memcpy(&nd[32 + 4 * j + 3], &cnt, 4);
    }
  }
  out.n_nodes = (int32_t)(N.size() / kNodeFloats);
  out.n_tris = (int32_t)(out.tris.size() / kTriFloats);
  out.max_stack = max_stack;
  if (out.n_tris != n) { err = "internal: triangle count mismatch after collapse"; return false; }
  return true;
}

}  // namespace art

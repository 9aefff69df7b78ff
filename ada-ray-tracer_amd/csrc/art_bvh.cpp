// art_bvh.cpp -- host-side BVH construction for the closest-hit mesh.  Replaces Embree's rtcCommitScene
// behind gcore_commit_scene (embree_connect.cpp:241-244): binned-SAH BVH2 (built with a small thread
// pool), collapsed by surface area into 8-wide nodes laid out as the 256-byte lane packets described
// in art_scene.h.  Boxes are inflated by a few ulps so that the kernel's slab test can never cull a
// triangle that the reference's Moeller-Trumbore arithmetic would accept.
#include "art_bvh.h"
#include "art_qnode.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <future>
#include <mutex>
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

struct Ref { Box box; int32_t tri; };   // a (possibly clipped) reference to a triangle

// bounds of (triangle  intersected with  the slab p0 <= x[a] <= p1), rounded outwards to binary32.  The clipped polygon's
// corners are the triangle corners inside the slab plus the edge/plane crossings, so their bounds are the polygon's bounds.
static bool clip_bounds(const float* tri, int a, double p0, double p1, Box& out) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  auto grow = [&](const double* p) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } };
  for (int i = 0; i < 3; ++i) {
    const float* u = tri + 3 * i; const float* v = tri + 3 * ((i + 1) % 3);
    const double ua = u[a], va = v[a];
    if (ua >= p0 && ua <= p1) { const double p[3] = {u[0], u[1], u[2]}; grow(p); }
    for (int s = 0; s < 2; ++s) {
      const double pl = s ? p1 : p0;
      if (!std::isfinite(pl)) continue;
      if ((ua < pl && va > pl) || (ua > pl && va < pl)) {
        const double t = (pl - ua) / (va - ua);
        double p[3] = {u[0] + t * ((double)v[0] - u[0]), u[1] + t * ((double)v[1] - u[1]), u[2] + t * ((double)v[2] - u[2])};
        p[a] = pl;
        grow(p);
      }
    }
  }
  if (!(lo[0] <= hi[0])) return false;
  for (int k = 0; k < 3; ++k) {
    float l = (float)lo[k], h = (float)hi[k];
    if ((double)l > lo[k]) l = std::nextafterf(l, -INFINITY);
    if ((double)h < hi[k]) h = std::nextafterf(h, INFINITY);
    out.lo[k] = l; out.hi[k] = h;
  }
  return true;
}

static bool intersect(Box& b, const Box& o) {
  for (int k = 0; k < 3; ++k) { b.lo[k] = std::max(b.lo[k], o.lo[k]); b.hi[k] = std::min(b.hi[k], o.hi[k]); if (b.lo[k] > b.hi[k]) return false; }
  return true;
}

struct Builder {
  const float* tri;       // 9 floats per triangle
  int32_t n;
  std::vector<Node2> nodes;
  std::vector<int32_t> leaf_ids;   // triangle ids of all leaves, appended leaf by leaf
  std::mutex leaf_mu;
  std::atomic<int32_t> next_node{0};
  std::atomic<int64_t> spatial_budget{0};   // extra references spatial splits may still create
  float root_area = 0.0f;
  BvhBuildParams prm;

  int32_t alloc_node() { return next_node.fetch_add(1); }

  // build the subtree over `refs` (consumed); returns the node index
  int32_t build(std::vector<Ref>& refs, int depth, int par_depth) {
    const int32_t id = alloc_node();
    const int32_t count = (int32_t)refs.size();
    Node2 nd;
    nd.box.reset();
    Box cb; cb.reset();
    for (const Ref& r : refs) {
      nd.box.grow(r.box);
      const float c[3] = {0.5f * r.box.lo[0] + 0.5f * r.box.hi[0], 0.5f * r.box.lo[1] + 0.5f * r.box.hi[1], 0.5f * r.box.lo[2] + 0.5f * r.box.hi[2]};
      cb.grow(c);
    }
    if (depth == 0) root_area = nd.box.half_area();
    auto make_leaf = [&]() {
      std::vector<int32_t> ids; ids.reserve(refs.size());
      for (const Ref& r : refs) ids.push_back(r.tri);
      std::sort(ids.begin(), ids.end());                       // ascending prim order: deterministic, and the order the leaf is tested in
      ids.erase(std::unique(ids.begin(), ids.end()), ids.end());   // two pieces of one triangle may meet again in a leaf
      {
        std::lock_guard<std::mutex> lk(leaf_mu);
        nd.first = (int32_t)leaf_ids.size(); nd.count = (int32_t)ids.size();
        leaf_ids.insert(leaf_ids.end(), ids.begin(), ids.end());
      }
      nodes[id] = nd; return id;
    };
    if (count <= 1) return make_leaf();
    auto cen_of = [](const Ref& r, int a) { return 0.5f * r.box.lo[a] + 0.5f * r.box.hi[a]; };

    // ---- object split: binned SAH over the reference centroids
    constexpr int NBMAX = 128; const int NB = prm.sah_bins;
    float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
    Box best_lbox, best_rbox; best_lbox.reset(); best_rbox.reset();
    const float parent_area = nd.box.half_area();
    for (int a = 0; a < 3; ++a) {
      const float ext = cb.hi[a] - cb.lo[a];
      if (!(ext > 0.0f)) continue;
      Box bb[NBMAX]; int32_t bc[NBMAX];
      for (int b = 0; b < NB; ++b) { bb[b].reset(); bc[b] = 0; }
      const float scale = (float)NB / ext;
      for (const Ref& r : refs) {
        int b = (int)((cen_of(r, a) - cb.lo[a]) * scale);
        b = std::min(std::max(b, 0), NB - 1);
        bb[b].grow(r.box); bc[b]++;
      }
      float la[NBMAX]; int32_t lc[NBMAX]; Box lb[NBMAX];
      Box acc; acc.reset(); int32_t c = 0;
      for (int b = 0; b < NB - 1; ++b) { acc.grow(bb[b]); c += bc[b]; la[b] = acc.half_area(); lc[b] = c; lb[b] = acc; }
      acc.reset(); c = 0;
      for (int b = NB - 1; b >= 1; --b) {
        acc.grow(bb[b]); c += bc[b];
        if (lc[b - 1] == 0 || c == 0) continue;
        const float cost = la[b - 1] * prm.leaf_cost((int)lc[b - 1]) + acc.half_area() * prm.leaf_cost((int)c);
        if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; best_lbox = lb[b - 1]; best_rbox = acc; }
      }
    }

    // ---- spatial split (Stich et al. 2009): only where the children of the object split overlap noticeably
    int sp_axis = -1; double sp_plane = 0.0; float sp_cost = INFINITY; int32_t sp_nl = 0, sp_nr = 0;
    if (prm.spatial_alpha >= 0.0f && best_axis >= 0 && spatial_budget.load(std::memory_order_relaxed) > 0 && depth <= prm.max_sah_depth) {
      Box ov = best_lbox;
      const bool overlap = intersect(ov, best_rbox);
      if (overlap && ov.half_area() > prm.spatial_alpha * root_area) {
        const int NS = std::min(std::max(prm.spatial_bins, 2), 32);
        for (int a = 0; a < 3; ++a) {
          const double lo = nd.box.lo[a], ext = (double)nd.box.hi[a] - lo;
          if (!(ext > 0.0)) continue;
          Box bb[32]; int32_t en[32], ex[32];
          for (int b = 0; b < NS; ++b) { bb[b].reset(); en[b] = 0; ex[b] = 0; }
          const double scale = NS / ext;
          auto plane = [&](int k) { return lo + ext * ((double)k / NS); };
          for (const Ref& r : refs) {
            int b0 = std::min(std::max((int)(((double)r.box.lo[a] - lo) * scale), 0), NS - 1);
            int b1 = std::min(std::max((int)(((double)r.box.hi[a] - lo) * scale), 0), NS - 1);
            if (b1 < b0) b1 = b0;
            if (b0 == b1) bb[b0].grow(r.box);
            else
              for (int b = b0; b <= b1; ++b) {
                Box cbx;
                if (clip_bounds(tri + 9 * (size_t)r.tri, a, plane(b), plane(b + 1), cbx) && intersect(cbx, r.box)) bb[b].grow(cbx);
              }
            en[b0]++; ex[b1]++;
          }
          float la[32]; int32_t lc[32];
          Box acc; acc.reset(); int32_t c = 0;
          for (int b = 0; b < NS - 1; ++b) { acc.grow(bb[b]); c += en[b]; la[b] = acc.half_area(); lc[b] = c; }
          acc.reset(); c = 0;
          for (int b = NS - 1; b >= 1; --b) {
            acc.grow(bb[b]); c += ex[b];
            if (lc[b - 1] == 0 || c == 0 || lc[b - 1] >= count || c >= count) continue;
            const float cost = la[b - 1] * prm.leaf_cost((int)lc[b - 1]) + acc.half_area() * prm.leaf_cost((int)c);
            if (cost < sp_cost) { sp_cost = cost; sp_axis = a; sp_plane = plane(b); sp_nl = lc[b - 1]; sp_nr = c; }
          }
        }
      }
    }
    const bool use_spatial = sp_axis >= 0 && sp_cost < best_cost;
    const float chosen_cost = use_spatial ? sp_cost : best_cost;

    const bool can_leaf = count <= prm.max_leaf;
    if (best_axis >= 0 && can_leaf) {
      const float split_cost = prm.node_cost * parent_area + chosen_cost;
      if (!(split_cost < parent_area * prm.leaf_cost(count))) return make_leaf();
    }
    std::vector<Ref> left, right;
    bool done = false;
    if (use_spatial && spatial_budget.fetch_sub((int64_t)(sp_nl + sp_nr - count), std::memory_order_relaxed) > 0) {
      const int a = sp_axis;
      left.reserve(sp_nl + 8); right.reserve(sp_nr + 8);
      for (const Ref& r : refs) {
        if ((double)r.box.hi[a] <= sp_plane) left.push_back(r);
        else if ((double)r.box.lo[a] >= sp_plane) right.push_back(r);
        else {
          Ref l = r, q = r;
          const bool hl = clip_bounds(tri + 9 * (size_t)r.tri, a, -INFINITY, sp_plane, l.box) && intersect(l.box, r.box);
          const bool hr = clip_bounds(tri + 9 * (size_t)r.tri, a, sp_plane, INFINITY, q.box) && intersect(q.box, r.box);
          if (hl) left.push_back(l);
          if (hr) right.push_back(q);
          if (!hl && !hr) left.push_back(r);   // cannot happen for a finite triangle; keep the reference rather than lose it
        }
      }
      done = !left.empty() && !right.empty() && (int32_t)left.size() < count && (int32_t)right.size() < count;
      if (!done) { left.clear(); right.clear(); }
    }
    if (!done) {
      int32_t mid;
      if (best_axis < 0 || depth > prm.max_sah_depth) {
        if (can_leaf && best_axis < 0) return make_leaf();
        // degenerate centroids or depth guard: median split by index on the widest box axis
        int a = 0;
        for (int k = 1; k < 3; ++k) if (nd.box.hi[k] - nd.box.lo[k] > nd.box.hi[a] - nd.box.lo[a]) a = k;
        mid = count / 2;
        std::nth_element(refs.begin(), refs.begin() + mid, refs.end(),
                         [&](const Ref& x, const Ref& y) { const float cx = cen_of(x, a), cy = cen_of(y, a); return cx < cy || (cx == cy && x.tri < y.tri); });
      } else {
        const float ext = cb.hi[best_axis] - cb.lo[best_axis];
        const float scale = (float)NB / ext;
        auto it = std::partition(refs.begin(), refs.end(), [&](const Ref& r) {
          int b = (int)((cen_of(r, best_axis) - cb.lo[best_axis]) * scale);
          b = std::min(std::max(b, 0), NB - 1);
          return b < best_bin;
        });
        mid = (int32_t)(it - refs.begin());
        if (mid == 0 || mid == count) mid = count / 2;
      }
      left.assign(refs.begin(), refs.begin() + mid);
      right.assign(refs.begin() + mid, refs.end());
    }
    std::vector<Ref>().swap(refs);
    if (par_depth > 0 && count > 20000) {
      auto fut = std::async(std::launch::async, [&, depth, par_depth]() { return build(left, depth + 1, par_depth - 1); });
      nd.right = build(right, depth + 1, par_depth - 1);
      nd.left = fut.get();
    } else {
      nd.left = build(left, depth + 1, 0);
      nd.right = build(right, depth + 1, 0);
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
  if (prm.width != 4 && prm.width != 8) { err = "BVH width must be 4 or 8"; return false; }
  out.width = prm.width;
  const int W = prm.width, NF = node_floats(W);
  if (n <= 0) return true;
  Builder B;
  B.tri = tri9; B.n = n; B.prm = prm;
  B.prm.max_leaf = std::min(B.prm.max_leaf, W);      // a leaf is one W-lane packet of triangle tests
  std::vector<Ref> refs((size_t)n);
  for (int32_t i = 0; i < n; ++i) {
    Box b; b.reset();
    const float* t = tri9 + 9 * (size_t)i;
    b.grow(t); b.grow(t + 3); b.grow(t + 6);
    for (int a = 0; a < 3; ++a)
      if (!std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) { err = "non-finite triangle vertex"; return false; }
    for (int a = 0; a < 3; ++a)
      if (std::fabs(b.lo[a]) > 1.0e18f || std::fabs(b.hi[a]) > 1.0e18f) { err = "triangle coordinate beyond 1e18: box extents and the node quantisation need headroom in binary32"; return false; }
    refs[i].box = b; refs[i].tri = i;
  }
  const int64_t budget = (prm.spatial_alpha >= 0.0f) ? (int64_t)((double)prm.spatial_budget * n) : 0;
  B.spatial_budget = budget;
  B.nodes.resize(2 * ((size_t)n + (size_t)budget + 64) + 2);   // every split makes progress, so nodes <= 2 * references
  B.leaf_ids.reserve((size_t)n + (size_t)budget);
  const int32_t root = B.build(refs, 0, prm.parallel_depth);
  (void)root;

  // ---- collapse to 8-wide
  std::vector<float>& N = out.nodes;
  auto new_node8 = [&]() { const int32_t id = (int32_t)(N.size() / NF); N.resize(N.size() + NF, 0.0f); return id; };
  out.tris.reserve((size_t)n * kTriFloats);
  const float inflate_rel = prm.inflate_rel, inflate_abs = prm.inflate_abs;

  auto emit_leaf_tris = [&](const Node2& lf) -> int32_t {
    const int32_t first = (int32_t)(out.tris.size() / kTriFloats);
    // triangles inside a leaf are stored in ascending prim order (deterministic, and the order the leaf is tested in)
    for (int32_t k = lf.first; k < lf.first + lf.count; ++k) {
      const int32_t t = B.leaf_ids[k];
      float rec[kTriFloats];
      std::memcpy(rec, tri9 + 9 * (size_t)t, 9 * sizeof(float));
      const int32_t pid = prim_ids ? prim_ids[t] : t;
      std::memcpy(&rec[9], &pid, 4); rec[10] = 0.0f; rec[11] = 0.0f;
      out.tris.insert(out.tris.end(), rec, rec + kTriFloats);
    }
    return first;
  };

  const int32_t root8 = new_node8();   // a single-leaf tree still gets a root node with one child

  // prm.collapse == 1: cost-optimal collapse (the dynamic programme of Ylitie, Karras, Laine 2017, section 4.1, restricted to the
  // leaves the SAH build chose).  cst[n][i-1] = least expected number of wide-node visits below BVH2 node n when n's subtree may take up
  // to i slots of the wide node above it: either n becomes a wide node itself (one slot: area(n) + the best distribution of W slots
  // over its two children), or n is dissolved and its children share the i slots.  The greedy rule (open the child with the largest
  // area) is the usual approximation of exactly this minimum.
  std::vector<float> cst; std::vector<uint8_t> cut;     // cut[n][i-1]: slots given to the left child when n is dissolved into i slots; 0 = n is a wide node
  std::vector<uint8_t> root_cut;                        // [n]: left child's share of the W slots when n is a wide node
  const bool optimal = prm.collapse == 1 && B.nodes[0].count == 0;
  if (optimal) {
    const size_t nn = (size_t)B.next_node.load();
    cst.assign(nn * W, 0.0f); cut.assign(nn * W, 0); root_cut.assign(nn, 0);
    // children have larger... no ordering guarantee on ids (parallel build): explicit post-order
    std::vector<int32_t> order; order.reserve(nn);
    { std::vector<int32_t> stk; stk.push_back(0);
      while (!stk.empty()) { const int32_t n = stk.back(); stk.pop_back(); order.push_back(n); const Node2& nd2 = B.nodes[n]; if (nd2.count == 0) { stk.push_back(nd2.left); stk.push_back(nd2.right); } } }
    for (size_t q = order.size(); q-- > 0;) {
      const int32_t n = order[q]; const Node2& nd2 = B.nodes[n];
      float* c = &cst[(size_t)n * W];
      if (nd2.count > 0) { for (int i = 0; i < W; ++i) c[i] = 0.0f; continue; }          // fixed leaves: their cost is the same in every collapse
      const float* cl = &cst[(size_t)nd2.left * W]; const float* cr = &cst[(size_t)nd2.right * W];
      auto distribute = [&](int slots, uint8_t& k_best) {
        float best = INFINITY; k_best = 1;
        for (int k = 1; k < slots; ++k) { const float v = cl[k - 1] + cr[slots - k - 1]; if (v < best) { best = v; k_best = (uint8_t)k; } }
        return best;
      };
      const float as_node = nd2.box.half_area() + distribute(W, root_cut[n]);
      c[0] = as_node; cut[(size_t)n * W] = 0;
      for (int i = 2; i <= W; ++i) {
        uint8_t k; const float d = distribute(i, k);
        if (d < as_node) { c[i - 1] = d; cut[(size_t)n * W + i - 1] = k; } else { c[i - 1] = as_node; cut[(size_t)n * W + i - 1] = 0; }
      }
    }
  }

  int32_t max_stack = 1;
  struct Pending { int32_t n2, n8, stack_before; };
  std::vector<Pending> todo; todo.push_back({0, root8, 0});
  while (!todo.empty()) {
    const Pending p = todo.back(); todo.pop_back();
    // gather up to 8 children by repeatedly opening the inner child with the largest area
    std::vector<int32_t> ch;
    const Node2& top = B.nodes[p.n2];
    if (top.count > 0) ch.push_back(p.n2);
    else if (optimal) {
      // hand out the W slots top-down along the recorded minimum: in order, so BVH2 siblings stay neighbours
      struct Give { int32_t n; int slots; };
      std::vector<Give> st2;
      const int kl = root_cut[p.n2];
      st2.push_back({top.right, W - kl}); st2.push_back({top.left, kl});
      while (!st2.empty()) {
        const Give g = st2.back(); st2.pop_back();
        const Node2& c = B.nodes[g.n];
        const int k = (c.count > 0) ? 0 : cut[(size_t)g.n * W + g.slots - 1];
        if (c.count > 0 || k == 0) { ch.push_back(g.n); continue; }
        st2.push_back({c.right, g.slots - k}); st2.push_back({c.left, k});
      }
    }
    else { ch.push_back(top.left); ch.push_back(top.right); }
    while (!optimal && (int)ch.size() < W) {
      int best = -1; float best_a = -1.0f;
      for (int i = 0; i < (int)ch.size(); ++i) {
        const Node2& c = B.nodes[ch[i]];
        if (c.count > 0) continue;
        const float a = c.box.half_area();
        if (a > best_a) { best_a = a; best = i; }
      }
      if (best < 0) break;
      const Node2 c = B.nodes[ch[best]];
      ch[best] = c.left; ch.insert(ch.begin() + best + 1, c.right);   // in-order: BVH2 siblings stay neighbours (slots, hence node indices: two 64-byte nodes share an L2 line)
    }
    // a BVH2 leaf with more than kMaxLeafTris triangles cannot occur (max_leaf <= 8 is enforced below)
    float* nd = &N[(size_t)p.n8 * NF];
    const int nch = (int)ch.size();
    const int32_t stack_here = p.stack_before + (nch - 1);
    max_stack = std::max(max_stack, stack_here + 1);
    for (int j = 0; j < W; ++j) {
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
          nd = &N[(size_t)p.n8 * NF];      // N may have been reallocated
          todo.push_back({ch[j], ref, stack_here});
        }
      }
      nd[4 * j + 0] = lo[0]; nd[4 * j + 1] = lo[1]; nd[4 * j + 2] = lo[2]; std::memcpy(&nd[4 * j + 3], &ref, 4);
      nd[4 * W + 4 * j + 0] = hi[0]; nd[4 * W + 4 * j + 1] = hi[1]; nd[4 * W + 4 * j + 2] = hi[2]; std::memcpy(&nd[4 * W + 4 * j + 3], &cnt, 4);
    }
  }
  out.n_nodes = (int32_t)(N.size() / NF);
  out.n_tris = (int32_t)(out.tris.size() / kTriFloats);
  if (W == 4 && prm.quantise) {                     // 64-byte form for k_trace_coop; N becomes the tree with the dequantised boxes
    out.qnodes.resize((size_t)out.n_nodes * (kQNodeBytes / 4));
    for (int32_t i = 0; i < out.n_nodes; ++i) {
      QNode q;
      quantise_node(&N[(size_t)i * NF], q);
      std::memcpy(&out.qnodes[(size_t)i * (kQNodeBytes / 4)], &q, sizeof q);
    }
  }
  out.max_stack = max_stack;
  if (out.n_tris != (int32_t)B.leaf_ids.size() || out.n_tris < n) { err = "internal: triangle count mismatch after collapse"; return false; }
  return true;
}

}  // namespace art

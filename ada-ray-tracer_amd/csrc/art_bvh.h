// art_bvh.h -- host BVH8 builder interface (see art_bvh.cpp).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "art_scene.h"

namespace art {

struct BvhBuildParams {
  int   width = 4;               // children per node: 8 or 4 (= lanes per ray in the trace kernel); leaves hold <= width triangles
  int   max_leaf = 8;            // <= kMaxLeafTris: one leaf = one 8-lane packet of triangle tests
  // SAH costs in units of one cooperative traversal step: a leaf of <= 8 triangles is ONE 8-lane packet whatever
  // its size, and a BVH2 split only costs a fraction of a BVH8 node step after the collapse.
  float node_cost = 0.4f;        // cost of one BVH2 inner node
  float leaf_base = 1.0f;        // fixed cost of visiting a leaf
  float tri_cost = -1.0f;        // cost per triangle in a leaf; < 0: 0.05 for width 8, 0.2 for width 4 (measured optima on C4)
  int   sah_bins = 32;           // bins of the object split (<= 128); 16 / 64 / 128 bins move the node visits per ray on C4 by +0.5 / -0.3 / -0.4 %
  int   max_sah_depth = 48;      // beyond this BVH2 depth fall back to median splits
  int   parallel_depth = 3;      // top levels built by std::async tasks
  float inflate_rel = 8.0e-6f;   // conservative padding of child boxes (relative to |coordinate|)
  float inflate_abs = 1.0e-6f;
  // spatial splits (SBVH): references whose boxes straddle a split plane are cut in two, each half bounded by the clipped
  // triangle.  The same triangle may then sit in several leaves (harmless: the closest hit is a minimum over (t, key)).
  float spatial_alpha = -1.0f;   // try a spatial split where overlap area of the object split > alpha * root area; < 0: off
  float spatial_budget = 0.5f;   // extra references allowed, as a fraction of the triangle count
  int   spatial_bins = 16;
  int   builder = 3;             // 3 (default since round 3): binned SAH on the GPU (art_sah.hip) -- the tree of builder 0, built in milliseconds;
                                 // 0: binned SAH on the host (art_bvh.cpp; also used below 2 triangles and for spatial splits); 1 LBVH, 2 PLOC on the GPU (art_lbvh.hip)
  int   collapse = 0;            // BVH2 -> wide: 0 open the child with the largest area until the node is full; 1 cost-optimal (fewest expected wide-node visits)
  int   ploc_radius = 8;         // PLOC: neighbours searched on either side
  int   inst_open = 0;           // instanced scenes: entry points per instance the instance tree ends at, on average (art_instanced_build.h).  1: whole
                                 // instances; 0 (default): 1 where the instances' boxes overlap little, up to 64 where they interpenetrate -- the rule and
                                 // the measurements behind it are in art_instanced_build.cpp
  int   gpu_max_leaf = 0;        // GPU builders: a subtree of at most this many triangles becomes one leaf; 0 = measured optimum
                                 // (1 for width 4, 2 for width 8: their trees have no SAH leaf term, small leaves cull better)
  int   quantise = 1;            // width 4: child boxes snapped outwards to the 8-bit grid of the 64-byte node (art_qnode.h); 0 keeps binary32 boxes
                                 // (then the trace kernel cannot run the tree: test-only, to measure what the snapping costs in visits)
  float leaf_cost(int n) const { return leaf_base + (tri_cost >= 0.0f ? tri_cost : (width == 4 ? 0.2f : 0.05f)) * (float)n; }
};

struct Bvh8 {
  std::vector<float> nodes;      // node_floats(width) per node, node 0 = root
  int32_t width = 8;
  std::vector<uint32_t> qnodes;  // width 4: the same nodes in the 64-byte quantised form (art_qnode.h), 16 words per node
  std::vector<float> tris;       // kTriFloats per triangle, in leaf order
  int32_t n_nodes = 0, n_tris = 0;
  int32_t max_stack = 1;         // worst-case traversal stack entries for this tree
};

// tri9: 9 floats per triangle (A,B,C); prim_ids: id written into each triangle record (nullptr = 0..n-1)
bool build_bvh8(const float* tri9, const int32_t* prim_ids, int32_t n, const BvhBuildParams& prm, Bvh8& out, std::string& err);

}  // namespace art

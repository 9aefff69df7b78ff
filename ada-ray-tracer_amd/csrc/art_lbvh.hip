// art_lbvh.hip -- BVH construction ON the GPU (replaces Embree's rtcCommitScene,
// embree_connect.cpp:241-244, for scenes that change or are too large to wait for the host SAH build).
//
//   1. triangle bounds + centroids, scene bounds (workgroup reduction, one atomic per workgroup and bound)
//   2. 30-bit Morton code of the centroid, radix sort (hipCUB) of (code, triangle)
//   3. Karras 2012 binary radix tree: one thread per internal node, ties broken by sorted position
//   4. bottom-up fit of the boxes (second arrival at a node computes it)
//   5. level-by-level collapse into the 8-wide 256-byte node packets of art_scene.h: a subtree holding <= max_leaf
//      triangles becomes one leaf (its triangles are contiguous in Morton order), otherwise the child with the largest
//      surface area is opened until 8 children are reached -- the same rule as the host builder (art_bvh.cpp)
//   6. triangle records in Morton order
// The tree is of lower quality than the host's binned-SAH build (LBVH), but the search result does not depend on the
// tree: the closest hit is an order-independent minimum, and the child boxes get the same conservative inflation.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "art_lbvh.h"
#include "art_qnode.h"

namespace art {
namespace {

#define LB_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return false; } \
  } while (0)

__device__ __forceinline__ int enc(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }   // order-preserving
__host__ __device__ __forceinline__ float dec(int i) { const int j = i >= 0 ? i : i ^ 0x7fffffff; return __builtin_bit_cast(float, j); }

__global__ __launch_bounds__(256) void k_prep(const float* __restrict__ tri9, int n, float4* __restrict__ blo, float4* __restrict__ bhi, int* scene /*6 encoded*/) {
  __shared__ int s[6];
  if (threadIdx.x < 6) s[threadIdx.x] = (threadIdx.x < 3) ? 0x7fffffff : (int)0x80000000;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float* t = tri9 + 9 * (size_t)i;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(fminf(t[a], t[3 + a]), t[6 + a]); hi[a] = fmaxf(fmaxf(t[a], t[3 + a]), t[6 + a]); }
    blo[i] = make_float4(lo[0], lo[1], lo[2], 0.f); bhi[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
    for (int a = 0; a < 3; ++a) {
      const float c = 0.5f * lo[a] + 0.5f * hi[a];
      atomicMin(&s[a], enc(c)); atomicMax(&s[3 + a], enc(c));
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicMin(&scene[threadIdx.x], s[threadIdx.x]);
  else if (threadIdx.x < 6) atomicMax(&scene[threadIdx.x], s[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {
  v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
  v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
  return v;
}

__global__ __launch_bounds__(256) void k_morton(const float4* __restrict__ blo, const float4* __restrict__ bhi, int n, const int* __restrict__ scene,
                                                uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float lo[3] = {dec(scene[0]), dec(scene[1]), dec(scene[2])}, hi[3] = {dec(scene[3]), dec(scene[4]), dec(scene[5])};
  const float c[3] = {0.5f * blo[i].x + 0.5f * bhi[i].x, 0.5f * blo[i].y + 0.5f * bhi[i].y, 0.5f * blo[i].z + 0.5f * bhi[i].z};
  uint32_t q[3];
  for (int a = 0; a < 3; ++a) {
    const float ext = hi[a] - lo[a];
    const float u = ext > 0.0f ? (c[a] - lo[a]) / ext : 0.0f;
    q[a] = (uint32_t)fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
  }
  keys[i] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
  vals[i] = (uint32_t)i;
}

// node ids: >= 0 internal node, < 0 leaf at sorted position ~id
struct Lbvh {
  int n;
  const uint32_t* keys;     // sorted
  int2* child;              // [n-1]
  int* parent;              // [n-1] internal, then [n] leaves at offset n-1
  int2* range;              // [n-1] first,last (sorted positions)
  float4* nlo; float4* nhi; // [n-1] internal boxes
  const float4* llo; const float4* lhi;   // [n] leaf boxes in sorted order
  int* flag;                // [n-1]
};

__device__ __forceinline__ int delta(const Lbvh& T, int i, int j) {
  if (j < 0 || j >= T.n) return -1;
  const uint32_t a = T.keys[i], b = T.keys[j];
  return (a == b) ? 32 + __clz((uint32_t)i ^ (uint32_t)j) : __clz(a ^ b);
}

__global__ __launch_bounds__(256) void k_karras(Lbvh T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T.n - 1) return;
  const int d = (delta(T, i, i + 1) - delta(T, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = delta(T, i, i - d);
  int lmax = 2;
  while (delta(T, i, i + lmax * d) > dmin) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (delta(T, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(T, i, j);
  int s = 0;
  for (int div = 2, t = (l + 1) >> 1;; div <<= 1, t = (l + div - 1) / div) {
    if (delta(T, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const int left = (first == gamma) ? ~gamma : gamma;
  const int right = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
  T.child[i] = make_int2(left, right);
  T.range[i] = make_int2(first, last);
  if (left >= 0) T.parent[left] = i; else T.parent[T.n - 1 + ~left] = i;
  if (right >= 0) T.parent[right] = i; else T.parent[T.n - 1 + ~right] = i;
  if (i == 0) T.parent[0] = -1;
}

__global__ __launch_bounds__(256) void k_fit(Lbvh T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T.n) return;
  int node = T.parent[T.n - 1 + i];
  while (node >= 0) {
    if (atomicAdd(&T.flag[node], 1) == 0) return;        // first arrival: the sibling subtree is not finished yet
    __threadfence();
    const int2 c = T.child[node];
    const float4 alo = c.x >= 0 ? T.nlo[c.x] : T.llo[~c.x], ahi = c.x >= 0 ? T.nhi[c.x] : T.lhi[~c.x];
    const float4 blo = c.y >= 0 ? T.nlo[c.y] : T.llo[~c.y], bhi = c.y >= 0 ? T.nhi[c.y] : T.lhi[~c.y];
    T.nlo[node] = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.f);
    T.nhi[node] = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
    __threadfence();
    node = T.parent[node];
  }
}

__global__ __launch_bounds__(256) void k_gather_boxes(const float4* __restrict__ blo, const float4* __restrict__ bhi, const uint32_t* __restrict__ vals, int n,
                                                      float4* __restrict__ llo, float4* __restrict__ lhi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { llo[i] = blo[vals[i]]; lhi[i] = bhi[vals[i]]; }
}

__global__ __launch_bounds__(256) void k_emit_tris(const float* __restrict__ tri9, const uint32_t* __restrict__ vals, int n, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t t = vals[i];
  const float* s = tri9 + 9 * (size_t)t;
  float* o = out + (size_t)kTriFloats * i;
  for (int k = 0; k < 9; ++k) o[k] = s[k];
  o[9] = __int_as_float((int)t); o[10] = 0.0f; o[11] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// PLOC (parallel locally-ordered clustering, Meister & Bittner 2018) as an alternative to the radix tree: the Morton-sorted
// clusters are merged bottom-up; in every round each cluster picks, among its `radius` neighbours on either side, the one whose
// union with it has the smallest surface area, and mutual choices merge.  Quality is close to a SAH build at LBVH-like cost.
// One round = nearest-neighbour kernel, keep flags, one exclusive scan (compaction AND node numbering), merge + scatter kernel.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float union_area(float4 alo, float4 ahi, float4 blo, float4 bhi) {
  const float dx = fmaxf(ahi.x, bhi.x) - fminf(alo.x, blo.x), dy = fmaxf(ahi.y, bhi.y) - fminf(alo.y, blo.y), dz = fmaxf(ahi.z, bhi.z) - fminf(alo.z, blo.z);
  return dx * dy + dy * dz + dz * dx;
}

__global__ __launch_bounds__(256) void k_ploc_init(int n, const float4* __restrict__ llo, const float4* __restrict__ lhi, int* cid, float4* clo, float4* chi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { cid[i] = ~i; clo[i] = llo[i]; chi[i] = lhi[i]; }
}

__global__ __launch_bounds__(256) void k_ploc_nn(int m, int radius, const float4* __restrict__ clo, const float4* __restrict__ chi, int* __restrict__ nn) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const float4 lo = clo[i], hi = chi[i];
  // ties go to the partner i ^ 1 (then to the lower index): equal boxes then pair up (0,1), (2,3), ... and every round halves
  // them, instead of all pointing at the lowest index of their window (one merge per round)
  int best = (i ^ 1) < m ? (i ^ 1) : i - 1;
  float best_a = union_area(lo, hi, clo[best], chi[best]);
  const int j0 = max(0, i - radius), j1 = min(m - 1, i + radius);
  for (int j = j0; j <= j1; ++j) {
    if (j == i) continue;
    const float a = union_area(lo, hi, clo[j], chi[j]);
    if (a < best_a) { best_a = a; best = j; }
  }
  nn[i] = best;
}

// keep[i] = 0 for the higher-indexed partner of a mutual pair (it creates the merged node and puts it into its partner's place)
__global__ __launch_bounds__(256) void k_ploc_keep(int m, const int* __restrict__ nn, int* __restrict__ keep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  keep[i] = (j >= 0 && nn[j] == i && i > j) ? 0 : 1;
}

struct PlocOut { int2* child; int* parent; int* count; float4* nlo; float4* nhi; int n; };

__global__ __launch_bounds__(256) void k_ploc_merge(int m, int base, const int* __restrict__ nn, const int* __restrict__ keep, const int* __restrict__ pos,
                                                    const int* __restrict__ cid, const float4* __restrict__ clo, const float4* __restrict__ chi,
                                                    int* __restrict__ cid2, float4* __restrict__ clo2, float4* __restrict__ chi2, PlocOut O) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nn[i];
  const bool mutual = j >= 0 && nn[j] == i;
  if (mutual && i < j) return;                            // the partner writes the merged cluster into this slot's new position
  if (!mutual) { const int p = pos[i]; cid2[p] = cid[i]; clo2[p] = clo[i]; chi2[p] = chi[i]; return; }
  // i > j: new internal node; every dropped slot before i belongs to exactly one earlier merge, so the merges are numbered by i - pos[i]
  const int id = base + (i - pos[i]);
  const int a = cid[j], b = cid[i];                       // left = the lower slot
  const float4 alo = clo[j], ahi = chi[j], blo = clo[i], bhi = chi[i];
  const float4 lo = make_float4(fminf(alo.x, blo.x), fminf(alo.y, blo.y), fminf(alo.z, blo.z), 0.f);
  const float4 hi = make_float4(fmaxf(ahi.x, bhi.x), fmaxf(ahi.y, bhi.y), fmaxf(ahi.z, bhi.z), 0.f);
  O.child[id] = make_int2(a, b);
  O.nlo[id] = lo; O.nhi[id] = hi;
  O.count[id] = (a >= 0 ? O.count[a] : 1) + (b >= 0 ? O.count[b] : 1);
  if (a >= 0) O.parent[a] = id; else O.parent[O.n - 1 + ~a] = id;
  if (b >= 0) O.parent[b] = id; else O.parent[O.n - 1 + ~b] = id;
  const int p = pos[j];
  cid2[p] = id; clo2[p] = lo; chi2[p] = hi;
}

// depth-first numbering of the leaves: a node's leaves are contiguous, so a subtree of <= width triangles can become one leaf.
// start(x) = sum over the ancestors of x reached from their right child of count(left sibling).
__device__ __forceinline__ int ploc_start(const PlocOut& O, int x /* node id, or ~leaf */) {
  int start = 0;
  int p = x >= 0 ? O.parent[x] : O.parent[O.n - 1 + ~x];
  while (p >= 0) {
    const int2 c = O.child[p];
    if (c.y == x) start += (c.x >= 0 ? O.count[c.x] : 1);
    x = p; p = O.parent[p];
  }
  return start;
}

__global__ __launch_bounds__(256) void k_ploc_number_leaves(PlocOut O, const uint32_t* __restrict__ order_in, const float4* __restrict__ llo, const float4* __restrict__ lhi,
                                                            int* __restrict__ leaf_pos, uint32_t* __restrict__ order_out, float4* __restrict__ llo2, float4* __restrict__ lhi2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= O.n) return;
  const int p = ploc_start(O, ~k);
  leaf_pos[k] = p; order_out[p] = order_in[k]; llo2[p] = llo[k]; lhi2[p] = lhi[k];
}

__global__ __launch_bounds__(256) void k_ploc_number_nodes(PlocOut O, const int* __restrict__ leaf_pos, int2* __restrict__ range) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= O.n - 1) return;
  const int s = ploc_start(O, id);
  range[id] = make_int2(s, s + O.count[id] - 1);
  int2 c = O.child[id];
  if (c.x < 0) c.x = ~leaf_pos[~c.x];
  if (c.y < 0) c.y = ~leaf_pos[~c.y];
  O.child[id] = c;                                        // ploc_start of OTHER threads compares child ids: leaves are renamed in a second pass
}

struct Item { int n2, n8, stack_before; };

__device__ __forceinline__ float next_dn(float v) { return (v == 0.0f) ? -1.401298464e-45f : __int_as_float(__float_as_int(v) + (v > 0.0f ? -1 : 1)); }
__device__ __forceinline__ float next_up(float v) { return (v == 0.0f) ? 1.401298464e-45f : __int_as_float(__float_as_int(v) + (v > 0.0f ? 1 : -1)); }

// node_cap: capacity of `nodes` in wide nodes.  A child that would need node >= node_cap is written as an empty slot and *overflow is set
// (the host then fails the build): the kernel never writes outside its allocation.
__global__ __launch_bounds__(128) void k_collapse(Lbvh T, const Item* __restrict__ in, int n_in, Item* __restrict__ out, int* out_count, int* node_count,
                                                  int* max_stack, int* overflow, int node_cap, float* __restrict__ nodes, int width, int max_leaf,
                                                  float inflate_rel, float inflate_abs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_in) return;
  const Item it = in[q];
  int ch[8]; int nc;
  auto count_of = [&](int id) { return id >= 0 ? (T.range[id].y - T.range[id].x + 1) : 1; };
  auto first_of = [&](int id) { return id >= 0 ? T.range[id].x : ~id; };
  auto lo_of = [&](int id) { return id >= 0 ? T.nlo[id] : T.llo[~id]; };
  auto hi_of = [&](int id) { return id >= 0 ? T.nhi[id] : T.lhi[~id]; };
  if (it.n2 >= 0 && count_of(it.n2) > max_leaf) { const int2 c = T.child[it.n2]; ch[0] = c.x; ch[1] = c.y; nc = 2; }
  else { ch[0] = it.n2; nc = 1; }                       // tiny tree: a root with a single leaf child
  while (nc < width) {
    int best = -1; float best_a = -1.0f;
    for (int k = 0; k < nc; ++k) {
      const int id = ch[k];
      if (id < 0 || count_of(id) <= max_leaf) continue;
      const float4 lo = lo_of(id), hi = hi_of(id);
      const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
      const float a = dx * dy + dy * dz + dz * dx;
      if (a > best_a) { best_a = a; best = k; }
    }
    if (best < 0) break;
    const int2 c = T.child[ch[best]];
    ch[best] = c.x; ch[nc++] = c.y;
  }
  const int stack_here = it.stack_before + nc - 1;
  atomicMax(max_stack, stack_here + 1);
  float* nd = nodes + (size_t)it.n8 * (size_t)node_floats(width);
  const int hb = 4 * width;
  for (int j = 0; j < width; ++j) {
    int ref = -1, cnt = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (j < nc) {
      const int id = ch[j];
      const float4 l4 = lo_of(id), h4 = hi_of(id);
      const float l[3] = {l4.x, l4.y, l4.z}, h[3] = {h4.x, h4.y, h4.z};
      for (int a = 0; a < 3; ++a) {
        const float pad = inflate_abs + inflate_rel * fmaxf(fabsf(l[a]), fabsf(h[a]));
        lo[a] = next_dn(l[a] - pad); hi[a] = next_up(h[a] + pad);
      }
      const int c = count_of(id);
      if (c <= max_leaf) { ref = first_of(id); cnt = c; }
      else {
        ref = atomicAdd(node_count, 1); cnt = 0;
        if (ref < node_cap) {
          Item nx; nx.n2 = id; nx.n8 = ref; nx.stack_before = stack_here;
          out[atomicAdd(out_count, 1)] = nx;
        } else { ref = -1; atomicExch(overflow, 1); }
      }
    }
    nd[4 * j + 0] = lo[0]; nd[4 * j + 1] = lo[1]; nd[4 * j + 2] = lo[2]; nd[4 * j + 3] = __int_as_float(ref);
    nd[hb + 4 * j + 0] = hi[0]; nd[hb + 4 * j + 1] = hi[1]; nd[hb + 4 * j + 2] = hi[2]; nd[hb + 4 * j + 3] = __int_as_float(cnt);
  }
}

// width 4: the 64-byte quantised form of every node for k_trace_coop; the binary32 packets become the dequantised tree (art_qnode.h)
__global__ __launch_bounds__(256) void k_quantise_nodes(float* __restrict__ nodes, int n_nodes, QNode* __restrict__ qnodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  float nd[32];
  for (int k = 0; k < 32; ++k) nd[k] = nodes[(size_t)i * 32 + k];
  QNode q;
  quantise_node(nd, q);
  for (int k = 0; k < 32; ++k) nodes[(size_t)i * 32 + k] = nd[k];
  qnodes[i] = q;
}

struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T> bool get(T** p, size_t count, std::string& err) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T));
    if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return false; }
    ptrs.push_back(q); *p = (T*)q; return true;
  }
};

}  // namespace

void launch_quantise_nodes(hipStream_t st, float* nodes, int n_nodes, void* qnodes) {
  hipLaunchKernelGGL(k_quantise_nodes, dim3((n_nodes + 255) / 256), dim3(256), 0, st, nodes, n_nodes, (QNode*)qnodes);
}

bool build_bvh8_gpu(const float* d_tri9, int n, const BvhBuildParams& prm, hipStream_t st, GpuBvh& out, std::string& err) {
  if (prm.builder == 3) return build_bvh_sah_gpu(d_tri9, n, prm, st, out, err);
  if (n < 2) { err = "build_bvh8_gpu needs at least 2 triangles"; return false; }
  if (prm.width != 4 && prm.width != 8) { err = "BVH width must be 4 or 8"; return false; }
  const int max_leaf = std::min(prm.gpu_max_leaf > 0 ? prm.gpu_max_leaf : (prm.width == 4 ? 1 : 2), prm.width);
  Scratch S;
  float4 *blo, *bhi, *llo, *lhi, *nlo, *nhi; uint32_t *keys, *keys2, *vals, *vals2; int *scene, *parent, *flag, *counters; int2 *child, *range; Item *qa, *qb;
  if (!S.get(&blo, n, err) || !S.get(&bhi, n, err) || !S.get(&llo, n, err) || !S.get(&lhi, n, err) || !S.get(&nlo, n, err) || !S.get(&nhi, n, err) ||
      !S.get(&keys, n, err) || !S.get(&keys2, n, err) || !S.get(&vals, n, err) || !S.get(&vals2, n, err) || !S.get(&scene, 8, err) ||
      !S.get(&parent, 2 * (size_t)n, err) || !S.get(&flag, n, err) || !S.get(&counters, 8, err) || !S.get(&child, n, err) || !S.get(&range, n, err) ||
      !S.get(&qa, n, err) || !S.get(&qb, n, err))
    return false;
  struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;   // released on every return path
  LB_TRY(hipEventCreate(&ev.a)); LB_TRY(hipEventCreate(&ev.b));
  const hipEvent_t e0 = ev.a, e1 = ev.b;
  LB_TRY(hipEventRecord(e0, st));
  const int h_scene[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
  LB_TRY(hipMemcpyAsync(scene, h_scene, sizeof h_scene, hipMemcpyHostToDevice, st));
  LB_TRY(hipMemsetAsync(flag, 0, (size_t)n * sizeof(int), st));
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_prep, dim3(nb), dim3(256), 0, st, d_tri9, n, blo, bhi, scene);
  hipLaunchKernelGGL(k_morton, dim3(nb), dim3(256), 0, st, blo, bhi, n, scene, keys, vals);
  size_t tmp_bytes = 0;
  LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, n, 0, 30, st));
  void* tmp = nullptr;
  if (!S.get((char**)&tmp, tmp_bytes, err)) return false;
  LB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, n, 0, 30, st));
  hipLaunchKernelGGL(k_gather_boxes, dim3(nb), dim3(256), 0, st, blo, bhi, vals2, n, llo, lhi);
  Lbvh T; T.n = n; T.keys = keys2; T.child = child; T.parent = parent; T.range = range; T.nlo = nlo; T.nhi = nhi; T.llo = llo; T.lhi = lhi; T.flag = flag;
  const uint32_t* tri_order = vals2;          // triangle of every leaf position
  int root_id = 0;
  if (prm.builder == 2) {
    // ---- PLOC: merge the Morton-ordered clusters bottom-up
    int *cid, *cid2, *nn, *keep, *pos, *count, *leaf_pos; float4 *clo, *chi, *clo2, *chi2, *llo2, *lhi2; uint32_t* order2;
    if (!S.get(&cid, n, err) || !S.get(&cid2, n, err) || !S.get(&nn, n, err) || !S.get(&keep, n, err) || !S.get(&pos, n, err) || !S.get(&count, n, err) ||
        !S.get(&leaf_pos, n, err) || !S.get(&clo, n, err) || !S.get(&chi, n, err) || !S.get(&clo2, n, err) || !S.get(&chi2, n, err) ||
        !S.get(&llo2, n, err) || !S.get(&lhi2, n, err) || !S.get(&order2, n, err))
      return false;
    size_t scan_bytes = 0;
    LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, keep, pos, n, st));
    void* scan_tmp = nullptr;
    if (!S.get((char**)&scan_tmp, scan_bytes, err)) return false;
    LB_TRY(hipMemsetAsync(parent, 0xff, 2 * (size_t)n * sizeof(int), st));          // -1: no parent (the root keeps it)
    PlocOut O; O.child = child; O.parent = parent; O.count = count; O.nlo = nlo; O.nhi = nhi; O.n = n;
    hipLaunchKernelGGL(k_ploc_init, dim3(nb), dim3(256), 0, st, n, llo, lhi, cid, clo, chi);
    const int radius = std::max(1, prm.ploc_radius);
    int m = n, base = 0, rounds = 0;
    while (m > 1) {
      const int mb = (m + 255) / 256;
      hipLaunchKernelGGL(k_ploc_nn, dim3(mb), dim3(256), 0, st, m, radius, clo, chi, nn);
      hipLaunchKernelGGL(k_ploc_keep, dim3(mb), dim3(256), 0, st, m, nn, keep);
      LB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, scan_bytes, keep, pos, m, st));
      int last[2];
      LB_TRY(hipMemcpyAsync(&last[0], pos + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
      LB_TRY(hipMemcpyAsync(&last[1], keep + (m - 1), sizeof(int), hipMemcpyDeviceToHost, st));
      LB_TRY(hipStreamSynchronize(st));
      const int m_new = last[0] + last[1];
      if (m_new >= m || m_new < 1) { err = "internal: PLOC round made no progress"; return false; }   // the closest pair is always mutual
      hipLaunchKernelGGL(k_ploc_merge, dim3(mb), dim3(256), 0, st, m, base, nn, keep, pos, cid, clo, chi, cid2, clo2, chi2, O);
      base += m - m_new; m = m_new;
      std::swap(cid, cid2); std::swap(clo, clo2); std::swap(chi, chi2);
      if (++rounds > 4096) { err = "internal: PLOC did not terminate"; return false; }
    }
    if (base != n - 1) { err = "internal: PLOC node count"; return false; }
    root_id = n - 2;                                                                 // the last merge
    hipLaunchKernelGGL(k_ploc_number_leaves, dim3(nb), dim3(256), 0, st, O, vals2, llo, lhi, leaf_pos, order2, llo2, lhi2);
    hipLaunchKernelGGL(k_ploc_number_nodes, dim3(nb), dim3(256), 0, st, O, leaf_pos, range);
    T.llo = llo2; T.lhi = lhi2; tri_order = order2;
    out.levels = rounds;
  } else {
    hipLaunchKernelGGL(k_karras, dim3(nb), dim3(256), 0, st, T);
    hipLaunchKernelGGL(k_fit, dim3(nb), dim3(256), 0, st, T);
  }
  // triangle records in leaf order
  const size_t tri_bytes = (size_t)n * kTriFloats * sizeof(float);
  LB_TRY(hipMalloc(&out.tris, tri_bytes));
  hipLaunchKernelGGL(k_emit_tris, dim3(nb), dim3(256), 0, st, d_tri9, tri_order, n, out.tris);
  // collapse, one launch per level of the wide tree.  Every wide node opens at least one inner node of the binary tree (n - 1 of them), so
  // at most n - 1 wide nodes can appear; with max_leaf = 1 a balanced tree really needs ~2n/3 (n/2 + 2 was too small: ADVICE r1).
  const size_t node_cap = (size_t)n;
  LB_TRY(hipMalloc(&out.nodes, node_cap * (size_t)node_floats(prm.width) * sizeof(float)));
  const int h_cnt[4] = {0, 1, 1, 0};    // [0] next-queue length, [1] node count (root = 0 taken), [2] max stack, [3] node capacity overflow
  LB_TRY(hipMemcpyAsync(counters, h_cnt, sizeof h_cnt, hipMemcpyHostToDevice, st));
  const Item root = {root_id, 0, 0};
  LB_TRY(hipMemcpyAsync(qa, &root, sizeof root, hipMemcpyHostToDevice, st));
  int n_in = 1, levels = 0;
  Item *in = qa, *nx = qb;
  while (n_in > 0) {
    hipLaunchKernelGGL(k_collapse, dim3((n_in + 127) / 128), dim3(128), 0, st, T, in, n_in, nx, counters, counters + 1, counters + 2, counters + 3, (int)node_cap,
                       out.nodes, prm.width, max_leaf, prm.inflate_rel, prm.inflate_abs);
    int h[4];
    LB_TRY(hipMemcpyAsync(h, counters, sizeof h, hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    n_in = h[0]; out.n_nodes = h[1]; out.max_stack = h[2];
    if (h[3] != 0 || (size_t)out.n_nodes > node_cap) { err = "internal: LBVH node capacity exceeded"; return false; }
    LB_TRY(hipMemsetAsync(counters, 0, sizeof(int), st));
    std::swap(in, nx);
    if (++levels > 64) { err = "internal: LBVH collapse did not terminate"; return false; }
  }
  if (prm.width == 4 && prm.quantise) {
    LB_TRY(hipMalloc(&out.qnodes, (size_t)out.n_nodes * kQNodeBytes));
    launch_quantise_nodes(st, out.nodes, out.n_nodes, out.qnodes);
  }
  LB_TRY(hipEventRecord(e1, st));
  LB_TRY(hipEventSynchronize(e1));
  float ms = 0.0f;
  LB_TRY(hipEventElapsedTime(&ms, e0, e1));
  LB_TRY(hipGetLastError());
  out.n_tris = n; out.build_ms = ms; if (prm.builder != 2) out.levels = levels;
  return true;
}

}  // namespace art

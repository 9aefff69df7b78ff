// art_sah.hip -- binned-SAH BVH construction ON the GPU (option bvh_builder = 3, the default since round 3): the algorithm of the
// host builder (art_bvh.cpp: binned SAH over the reference centroids, leaf / split decision by the same cost model, collapse of the
// binary tree into W-wide nodes by opening the child with the largest area) restated breadth-first for the GPU.  It replaces Embree's
// rtcCommitScene (embree_connect.cpp:241-244) -- the 1.1-1.4 s the host build takes for 1M triangles become a few milliseconds --
// and it builds THE SAME TREE: every quantity a split decision depends on is a minimum, a maximum or an integer count over the SET of
// references of a node (boxes, centroid bounds, bin populations), evaluated with the host's binary32 expressions in the host's order,
// so the decisions do not depend on the order in which the references are visited.  (Only the fallbacks for degenerate inputs differ:
// a node whose centroids all coincide, or one deeper than max_sah_depth, is cut in the middle of its current reference order instead
// of by an nth_element on (centroid, triangle id).)
//
// Level by level over the binary tree, one host round trip per level (the sizes of the next level):
//   k_bin_big    nodes with more than kChunk references: one workgroup per chunk of kChunk references bins them in LDS (3 axes x NB bins x
//                {box, count}, integer atomics on order-preserving encodings) and merges into the node's bins in HBM
//   k_eval       ONE WAVE PER ACTIVE NODE.  Small nodes are binned by the wave itself (LDS); then lanes 0..2 sweep the bins of their axis
//                exactly like art_bvh.cpp:131-140, the wave decides leaf / split / fallback, and a small node is partitioned on the spot
//                (stable, ballot + prefix) into the other reference buffer together with its children's boxes and centroid bounds
//   scan         exclusive sums over the active nodes: children ids, next active list, chunks and bin slots of the next level's big nodes
//   k_commit     child records, next active list, chunk descriptors
//   k_part_*     big nodes: lefts per chunk, scan, stable chunk-wise partition + the children's boxes / centroid bounds by atomics
// No kernel waits on another workgroup and every loop is bounded by a count known at launch.
// Then: triangle records in reference order (a leaf's triangles sorted by id), and the collapse into wide nodes, breadth-first with
// scan-based numbering (children of one wide node get consecutive ids in slot order, BVH2 siblings stay neighbours: two 64-byte
// quantised nodes share a 128-byte line).  The whole build is deterministic.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <string>
#include <vector>

#include "art_lbvh.h"
#include "art_qnode.h"

namespace art {
namespace {

#define SB_TRY(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(_e); return false; } \
  } while (0)

constexpr int kChunk = 2048;              // references per chunk workgroup (256 threads x 8); a node with more references is "big"
constexpr int kRounds = kChunk / 256;
constexpr int kBinWords = 8;              // per bin: lo.xyz, hi.xyz (encoded), count, pad
constexpr int kMaxBins = 64;

// order-preserving float <-> int (min / max of encodings = min / max of the floats; -0 sorts below +0, which no result depends on)
__host__ __device__ __forceinline__ int enc(float f) { const int i = __builtin_bit_cast(int, f); return i >= 0 ? i : i ^ 0x7fffffff; }
__host__ __device__ __forceinline__ float dec(int i) { return __builtin_bit_cast(float, i >= 0 ? i : i ^ 0x7fffffff); }
constexpr int kEncPosInf = 0x7f800000;            // enc(+inf): identity of a minimum
constexpr int kEncNegInf = (int)0x807fffffu;      // enc(-inf): identity of a maximum

struct SahCost { int nb, max_leaf, max_depth, width; float node_cost, leaf_base, tri_cost; };
__device__ __forceinline__ float leaf_cost(const SahCost& P, int n) { return P.leaf_base + P.tri_cost * (float)n; }   // BvhBuildParams::leaf_cost

// art_bvh.cpp Box::half_area
__device__ __forceinline__ float half_area(float lx, float ly, float lz, float hx, float hy, float hz) {
  const float dx = hx - lx, dy = hy - ly, dz = hz - lz;
  return (dx < 0.0f) ? 0.0f : dx * dy + dy * dz + dz * dx;
}

// BVH2 nodes.  blo = {enc lo.xyz, first}, bhi = {enc hi.xyz, count}, clo = {enc centroid lo.xyz, depth}, chi = {enc centroid hi.xyz, 0};
// child = {left, right}, or {-1, -1} for a leaf.  A node's references are positions [first, first + count) of reference buffer depth & 1.
struct Nodes { int4* blo; int4* bhi; int4* clo; int4* chi; int2* child; };
struct Act { int node, big_slot, chunk0, pad; };                 // one entry of a level's active list
struct S4 { int s, a, c, b; };                                   // per active node: splits, active children, chunks and big children it creates
struct S4Sum { __host__ __device__ S4 operator()(const S4& x, const S4& y) const { return S4{x.s + y.s, x.a + y.a, x.c + y.c, x.b + y.b}; } };
enum { DEC_LEAF = 0, DEC_SAH = 1, DEC_POS = 2 };
struct Dec {                                                     // what k_eval decided for an active node
  int kind, axis, bin, nl;
  float cb_lo, scale; int pad0, pad1;                            // binning of the split axis (big nodes: read by the partition kernels)
  int lbox[6], rbox[6], lcb[6], rcb[6];                          // small nodes: the children's boxes and centroid bounds (encoded)
};

struct Args {
  Nodes N;
  const float4* rlo_in; const float4* rhi_in;                    // references of this level: {lo.xyz, triangle}, {hi.xyz, 0}
  float4* rlo_out; float4* rhi_out;                              // ... and of the next
  const Act* act; Act* act_next;
  const int2* chunk; int2* chunk_next;                           // chunk -> (active index, chunk number inside the node)
  Dec* dec; S4* flags; const S4* offs;
  int* gbins;                                                    // [big slot][3][NB][kBinWords]
  int* err;                                                      // device error word: set when a partition does not match its bins (read by the host every level)
  int* chunk_cnt; const int* chunk_off;                          // lefts per chunk, exclusive sums
  S4* totals;
  SahCost P;
  int node_base;                                                 // id of the first child this level creates
};

__device__ __forceinline__ int bin_of(float lo, float hi, float cb_lo, float scale, int nb) {       // art_bvh.cpp:127-128
  int b = (int)((0.5f * lo + 0.5f * hi - cb_lo) * scale);
  return min(max(b, 0), nb - 1);
}

__device__ __forceinline__ float sel3(int a, float x, float y, float z) { return (a == 0) ? x : (a == 1) ? y : z; }
__device__ __forceinline__ int wave_min_i(int v) { for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_max_i(int v) { for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- level 0: reference boxes, root box and centroid bounds ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_refs(const float* __restrict__ tri9, int n, float4* __restrict__ rlo, float4* __restrict__ rhi, Nodes N, int* bad) {
  __shared__ int s[12];
  if (threadIdx.x < 12) s[threadIdx.x] = ((threadIdx.x % 6) < 3) ? kEncPosInf : kEncNegInf;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float* t = tri9 + 9 * (size_t)i;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(fminf(t[a], t[3 + a]), t[6 + a]); hi[a] = fmaxf(fmaxf(t[a], t[3 + a]), t[6 + a]); }
    rlo[i] = make_float4(lo[0], lo[1], lo[2], __int_as_float(i)); rhi[i] = make_float4(hi[0], hi[1], hi[2], 0.0f);
    for (int a = 0; a < 3; ++a) {
      if (!(fabsf(lo[a]) <= 1.0e18f) || !(fabsf(hi[a]) <= 1.0e18f)) atomicExch(bad, 1);      // art_bvh.cpp:268-270 (also catches NaN / inf)
      const float c = 0.5f * lo[a] + 0.5f * hi[a];
      atomicMin(&s[a], enc(lo[a])); atomicMax(&s[3 + a], enc(hi[a]));
      atomicMin(&s[6 + a], enc(c)); atomicMax(&s[9 + a], enc(c));
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) { atomicMin(&((int*)&N.blo[0])[threadIdx.x], s[threadIdx.x]); atomicMin(&((int*)&N.clo[0])[threadIdx.x], s[6 + threadIdx.x]); }
  else if (threadIdx.x < 6) { atomicMax(&((int*)&N.bhi[0])[threadIdx.x - 3], s[threadIdx.x]); atomicMax(&((int*)&N.chi[0])[threadIdx.x - 3], s[6 + threadIdx.x]); }
}

__global__ void k_init_bins(int* g, int n_words) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) { const int w = i & (kBinWords - 1); g[i] = (w < 3) ? kEncPosInf : (w < 6) ? kEncNegInf : 0; }
}

__device__ __forceinline__ void bin_ref(int* bins, int nb, const float4 lo, const float4 hi, const float* cbl, const float* ext, const float* scale) {
  const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
  const int el[3] = {enc(lo.x), enc(lo.y), enc(lo.z)}, eh[3] = {enc(hi.x), enc(hi.y), enc(hi.z)};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (!(ext[a] > 0.0f)) continue;                                              // art_bvh.cpp:122
    int* q = bins + (a * nb + bin_of(l[a], h[a], cbl[a], scale[a], nb)) * kBinWords;
    atomicMin(q + 0, el[0]); atomicMin(q + 1, el[1]); atomicMin(q + 2, el[2]);
    atomicMax(q + 3, eh[0]); atomicMax(q + 4, eh[1]); atomicMax(q + 5, eh[2]);
    atomicAdd(q + 6, 1);
  }
}

// ---- big nodes: one workgroup per chunk bins its references in LDS and merges into the node's bins --------------------------------
__global__ __launch_bounds__(256) void k_bin_big(const Args A) {
  extern __shared__ int lds[];
  const int nb = A.P.nb, words = 3 * nb * kBinWords;
  for (int w = threadIdx.x; w < words; w += 256) { const int k = w & (kBinWords - 1); lds[w] = (k < 3) ? kEncPosInf : (k < 6) ? kEncNegInf : 0; }
  __syncthreads();
  const int2 cd = A.chunk[blockIdx.x];
  const Act act = A.act[cd.x];
  const int4 b0 = A.N.blo[act.node], b1 = A.N.bhi[act.node], c0 = A.N.clo[act.node], c1 = A.N.chi[act.node];
  const int first = b0.w, count = b1.w;
  const float cbl[3] = {dec(c0.x), dec(c0.y), dec(c0.z)}, cbh[3] = {dec(c1.x), dec(c1.y), dec(c1.z)};
  float ext[3], scale[3];
  for (int a = 0; a < 3; ++a) { ext[a] = cbh[a] - cbl[a]; scale[a] = (ext[a] > 0.0f) ? (float)nb / ext[a] : 0.0f; }
  for (int r = 0; r < kRounds; ++r) {
    const int k = cd.y * kChunk + r * 256 + (int)threadIdx.x;
    if (k < count) bin_ref(lds, nb, A.rlo_in[first + k], A.rhi_in[first + k], cbl, ext, scale);
  }
  __syncthreads();
  int* g = A.gbins + (size_t)act.big_slot * words;
  for (int w = threadIdx.x; w < words; w += 256) {
    const int k = w & (kBinWords - 1);
    if (k > 6 || lds[(w & ~(kBinWords - 1)) + 6] == 0) continue;                 // nothing fell into this bin
    if (k < 3) atomicMin(g + w, lds[w]); else if (k < 6) atomicMax(g + w, lds[w]); else atomicAdd(g + w, lds[w]);
  }
}

// ---- one wave per active node: bins (small nodes), the SAH sweep, the decision, and the partition of a small node --------------------
__global__ __launch_bounds__(256) void k_eval(const Args A, int m) {
  extern __shared__ int lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= m) return;                                                             // the kernel has no workgroup barrier
  const int nb = A.P.nb, words = 3 * nb * kBinWords;
  int* bins = lds + wave * (words + 6 * nb);
  float* la = reinterpret_cast<float*>(bins + words);                             // [3][nb] area of bins 0..b
  int* lc = bins + words + 3 * nb;                                                // [3][nb] references in bins 0..b
  const Act act = A.act[i];
  const int node = act.node;
  const int4 b0 = A.N.blo[node], b1 = A.N.bhi[node], c0 = A.N.clo[node], c1 = A.N.chi[node];
  const int first = b0.w, count = b1.w, depth = c0.w;
  const bool big = act.big_slot >= 0;
  const float cbl[3] = {dec(c0.x), dec(c0.y), dec(c0.z)}, cbh[3] = {dec(c1.x), dec(c1.y), dec(c1.z)};
  float ext[3], scale[3];
  for (int a = 0; a < 3; ++a) { ext[a] = cbh[a] - cbl[a]; scale[a] = (ext[a] > 0.0f) ? (float)nb / ext[a] : 0.0f; }
  if (big) {
    const int* g = A.gbins + (size_t)act.big_slot * words;
    for (int w = lane; w < words; w += 64) bins[w] = g[w];
  } else {
    for (int w = lane; w < words; w += 64) { const int k = w & (kBinWords - 1); bins[w] = (k < 3) ? kEncPosInf : (k < 6) ? kEncNegInf : 0; }
    wave_sync();
    for (int k = lane; k < count; k += 64) bin_ref(bins, nb, A.rlo_in[first + k], A.rhi_in[first + k], cbl, ext, scale);
  }
  wave_sync();

  // the sweep of art_bvh.cpp:131-140, lane a = axis a: left to right the area and population of bins 0..b, then right to left the cost of
  // every plane with references on both sides; strict "<" keeps the first minimum in the host's order (axis ascending, bin descending)
  float my_cost = __builtin_inff(); int my_bin = -1;
  if (lane < 3 && sel3(lane, ext[0], ext[1], ext[2]) > 0.0f) {
    const int* bb = bins + lane * nb * kBinWords;
    int lx = kEncPosInf, ly = kEncPosInf, lz = kEncPosInf, hx = kEncNegInf, hy = kEncNegInf, hz = kEncNegInf, c = 0;
    for (int b = 0; b < nb - 1; ++b) {
      const int* q = bb + b * kBinWords;
      lx = min(lx, q[0]); ly = min(ly, q[1]); lz = min(lz, q[2]); hx = max(hx, q[3]); hy = max(hy, q[4]); hz = max(hz, q[5]); c += q[6];
      la[lane * nb + b] = half_area(dec(lx), dec(ly), dec(lz), dec(hx), dec(hy), dec(hz)); lc[lane * nb + b] = c;
    }
    lx = ly = lz = kEncPosInf; hx = hy = hz = kEncNegInf; c = 0;
    for (int b = nb - 1; b >= 1; --b) {
      const int* q = bb + b * kBinWords;
      lx = min(lx, q[0]); ly = min(ly, q[1]); lz = min(lz, q[2]); hx = max(hx, q[3]); hy = max(hy, q[4]); hz = max(hz, q[5]); c += q[6];
      const int nl_b = lc[lane * nb + b - 1];
      if (nl_b == 0 || c == 0) continue;
      const float cost = la[lane * nb + b - 1] * leaf_cost(A.P, nl_b) + half_area(dec(lx), dec(ly), dec(lz), dec(hx), dec(hy), dec(hz)) * leaf_cost(A.P, c);
      if (cost < my_cost) { my_cost = cost; my_bin = b; }
    }
  }
  float best_cost = __builtin_inff(); int best_axis = -1, best_bin = -1;
  for (int a = 0; a < 3; ++a) {
    const float ca = __shfl(my_cost, a); const int ba = __shfl(my_bin, a);
    if (ba >= 0 && ca < best_cost) { best_cost = ca; best_axis = a; best_bin = ba; }
  }
  wave_sync();
  // the decision of art_bvh.cpp:185-189, 212-213
  const bool can_leaf = count <= A.P.max_leaf;
  int kind = DEC_SAH;
  if (best_axis >= 0 && can_leaf) {
    const float parent_area = half_area(dec(b0.x), dec(b0.y), dec(b0.z), dec(b1.x), dec(b1.y), dec(b1.z));
    const float split_cost = A.P.node_cost * parent_area + best_cost;
    if (!(split_cost < parent_area * leaf_cost(A.P, count))) kind = DEC_LEAF;
  }
  if (kind == DEC_SAH && (best_axis < 0 || depth > A.P.max_depth)) kind = (can_leaf && best_axis < 0) ? DEC_LEAF : DEC_POS;
  int nl = 0;
  if (kind == DEC_SAH) nl = lc[best_axis * nb + best_bin - 1];
  else if (kind == DEC_POS) nl = count / 2;
  const int ax = (kind == DEC_SAH) ? best_axis : 0;
  const float s_cbl = sel3(ax, cbl[0], cbl[1], cbl[2]), s_scale = sel3(ax, scale[0], scale[1], scale[2]);

  Dec d;
  d.kind = kind; d.axis = ax; d.bin = best_bin; d.nl = nl; d.cb_lo = s_cbl; d.scale = s_scale; d.pad0 = d.pad1 = 0;
  for (int k = 0; k < 6; ++k) { d.lbox[k] = d.lcb[k] = d.rbox[k] = d.rcb[k] = (k < 3) ? kEncPosInf : kEncNegInf; }
  if (kind != DEC_LEAF && !big) {
    // stable partition into the other buffer; the children's boxes and centroid bounds on the way
    int acc[24];                                                                   // lbox, lcb, rbox, rcb
    for (int k = 0; k < 24; ++k) acc[k] = ((k % 6) < 3) ? kEncPosInf : kEncNegInf;
    int run_l = 0, run_r = 0;
    for (int k0 = 0; k0 < count; k0 += 64) {
      const int k = k0 + lane;
      const bool valid = k < count;
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if (valid) { lo = A.rlo_in[first + k]; hi = A.rhi_in[first + k]; }
      const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
      const bool left = (kind == DEC_SAH) ? (bin_of(sel3(ax, lo.x, lo.y, lo.z), sel3(ax, hi.x, hi.y, hi.z), s_cbl, s_scale, nb) < best_bin) : (k < nl);
      const uint64_t ml = __builtin_amdgcn_ballot_w64(valid && left), mr = __builtin_amdgcn_ballot_w64(valid && !left);
      const uint64_t below = (1ull << lane) - 1ull;
      if (valid) {
        const int dst = left ? first + run_l + (int)__popcll(ml & below) : first + nl + run_r + (int)__popcll(mr & below);
        A.rlo_out[dst] = lo; A.rhi_out[dst] = hi;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int ec = enc(0.5f * l[a] + 0.5f * h[a]), el = enc(l[a]), eh = enc(h[a]);
          if (left) { acc[a] = min(acc[a], el); acc[3 + a] = max(acc[3 + a], eh); acc[6 + a] = min(acc[6 + a], ec); acc[9 + a] = max(acc[9 + a], ec); }
          else { acc[12 + a] = min(acc[12 + a], el); acc[15 + a] = max(acc[15 + a], eh); acc[18 + a] = min(acc[18 + a], ec); acc[21 + a] = max(acc[21 + a], ec); }
        }
      }
      run_l += (int)__popcll(ml); run_r += (int)__popcll(mr);
    }
#pragma unroll
    for (int k = 0; k < 24; ++k) acc[k] = ((k % 6) < 3) ? wave_min_i(acc[k]) : wave_max_i(acc[k]);
    for (int k = 0; k < 6; ++k) { d.lbox[k] = acc[k]; d.lcb[k] = acc[6 + k]; d.rbox[k] = acc[12 + k]; d.rcb[k] = acc[18 + k]; }
    if (run_l != nl && lane == 0) atomicExch(A.err, 1);                            // cannot happen (the bins and the partition apply the same expression); if it ever does the host fails the build
  }
  if (lane == 0) {
    A.dec[i] = d;
    S4 f = {0, 0, 0, 0};
    if (kind != DEC_LEAF) {
      f.s = 1;
      const int cc[2] = {nl, count - nl};
      for (int k = 0; k < 2; ++k)
        if (cc[k] > 1) { f.a += 1; if (cc[k] > kChunk) { f.b += 1; f.c += (cc[k] + kChunk - 1) / kChunk; } }
    }
    A.flags[i] = f;
  }
}

// ---- child records, next active list, chunk descriptors ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_commit(const Args A, int m) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const Act act = A.act[i];
  const Dec d = A.dec[i];
  const S4 o = A.offs[i];
  if (i == m - 1) { const S4 f = A.flags[i]; *A.totals = S4{o.s + f.s, o.a + f.a, o.c + f.c, o.b + f.b}; }
  const int node = act.node;
  if (d.kind == DEC_LEAF) { A.N.child[node] = make_int2(-1, -1); return; }
  const int first = A.N.blo[node].w, count = A.N.bhi[node].w, depth = A.N.clo[node].w;
  const int L = A.node_base + 2 * o.s, R = L + 1;
  A.N.child[node] = make_int2(L, R);
  const int nl = d.nl, nr = count - nl;
  // children of a big node get their boxes from the partition kernels (atomics on these identities)
  A.N.blo[L] = make_int4(d.lbox[0], d.lbox[1], d.lbox[2], first);      A.N.bhi[L] = make_int4(d.lbox[3], d.lbox[4], d.lbox[5], nl);
  A.N.clo[L] = make_int4(d.lcb[0], d.lcb[1], d.lcb[2], depth + 1);     A.N.chi[L] = make_int4(d.lcb[3], d.lcb[4], d.lcb[5], 0);
  A.N.blo[R] = make_int4(d.rbox[0], d.rbox[1], d.rbox[2], first + nl); A.N.bhi[R] = make_int4(d.rbox[3], d.rbox[4], d.rbox[5], nr);
  A.N.clo[R] = make_int4(d.rcb[0], d.rcb[1], d.rcb[2], depth + 1);     A.N.chi[R] = make_int4(d.rcb[3], d.rcb[4], d.rcb[5], 0);
  int pos = o.a, slot = o.b, c0 = o.c;
  const int ids[2] = {L, R}, cnt[2] = {nl, nr};
  for (int k = 0; k < 2; ++k) {
    if (cnt[k] <= 1) { A.N.child[ids[k]] = make_int2(-1, -1); continue; }       // art_bvh.cpp:112
    Act e; e.node = ids[k]; e.big_slot = -1; e.chunk0 = -1; e.pad = 0;
    if (cnt[k] > kChunk) {
      const int nch = (cnt[k] + kChunk - 1) / kChunk;
      e.big_slot = slot++; e.chunk0 = c0;
      for (int q = 0; q < nch; ++q) A.chunk_next[c0 + q] = make_int2(pos, q);
      c0 += nch;
    }
    A.act_next[pos++] = e;
  }
}

// ---- big nodes: stable partition, chunk by chunk ---------------------------------------------------------------------------------
__device__ __forceinline__ bool goes_left(const Dec& d, int nb, const float4 lo, const float4 hi, int k) {
  if (d.kind != DEC_SAH) return k < d.nl;
  const float l = (d.axis == 0) ? lo.x : (d.axis == 1) ? lo.y : lo.z, h = (d.axis == 0) ? hi.x : (d.axis == 1) ? hi.y : hi.z;
  return bin_of(l, h, d.cb_lo, d.scale, nb) < d.bin;
}

__global__ __launch_bounds__(256) void k_part_count(const Args A) {
  __shared__ int s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int2 cd = A.chunk[blockIdx.x];
  const Act act = A.act[cd.x];
  const Dec d = A.dec[cd.x];
  const int first = A.N.blo[act.node].w, count = A.N.bhi[act.node].w;
  int n = 0;
  for (int r = 0; r < kRounds; ++r) {
    const int k = cd.y * kChunk + r * 256 + (int)threadIdx.x;
    if (k < count && goes_left(d, A.P.nb, A.rlo_in[first + k], A.rhi_in[first + k], k)) ++n;
  }
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_n, n);
  __syncthreads();
  if (threadIdx.x == 0) A.chunk_cnt[blockIdx.x] = s_n;
}

__global__ __launch_bounds__(256) void k_part_write(const Args A) {
  __shared__ int s_acc[24];
  __shared__ int s_wl[4], s_wr[4];
  if (threadIdx.x < 24) s_acc[threadIdx.x] = ((threadIdx.x % 6) < 3) ? kEncPosInf : kEncNegInf;
  const int2 cd = A.chunk[blockIdx.x];
  const Act act = A.act[cd.x];
  const Dec d = A.dec[cd.x];
  const int first = A.N.blo[act.node].w, count = A.N.bhi[act.node].w;
  const int2 ch = A.N.child[act.node];
  const int lefts_before = A.chunk_off[blockIdx.x] - A.chunk_off[act.chunk0];
  int run_l = lefts_before, run_r = cd.y * kChunk - lefts_before;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int acc[24];
  for (int k = 0; k < 24; ++k) acc[k] = ((k % 6) < 3) ? kEncPosInf : kEncNegInf;
  for (int r = 0; r < kRounds; ++r) {
    const int k = cd.y * kChunk + r * 256 + (int)threadIdx.x;
    const bool valid = k < count;
    float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
    if (valid) { lo = A.rlo_in[first + k]; hi = A.rhi_in[first + k]; }
    const bool left = valid && goes_left(d, A.P.nb, lo, hi, k);
    const uint64_t ml = __builtin_amdgcn_ballot_w64(left), mr = __builtin_amdgcn_ballot_w64(valid && !left);
    __syncthreads();                                                               // the previous round's counts have been read
    if (lane == 0) { s_wl[wave] = (int)__popcll(ml); s_wr[wave] = (int)__popcll(mr); }
    __syncthreads();
    int wl = 0, wr = 0, tl = 0, tr = 0;
    for (int w = 0; w < 4; ++w) { if (w < wave) { wl += s_wl[w]; wr += s_wr[w]; } tl += s_wl[w]; tr += s_wr[w]; }
    if (valid) {
      const uint64_t below = (1ull << lane) - 1ull;
      const int dst = left ? first + run_l + wl + (int)__popcll(ml & below) : first + d.nl + run_r + wr + (int)__popcll(mr & below);
      A.rlo_out[dst] = lo; A.rhi_out[dst] = hi;
      const float l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int ec = enc(0.5f * l[a] + 0.5f * h[a]), el = enc(l[a]), eh = enc(h[a]);
        if (left) { acc[a] = min(acc[a], el); acc[3 + a] = max(acc[3 + a], eh); acc[6 + a] = min(acc[6 + a], ec); acc[9 + a] = max(acc[9 + a], ec); }
        else { acc[12 + a] = min(acc[12 + a], el); acc[15 + a] = max(acc[15 + a], eh); acc[18 + a] = min(acc[18 + a], ec); acc[21 + a] = max(acc[21 + a], ec); }
      }
    }
    run_l += tl; run_r += tr;
  }
  // the node's last chunk: its lefts must add up to what the bins said (k_eval's d.nl), else children would overlap or lose references
  if (threadIdx.x == 0 && (cd.y + 1) * kChunk >= count && run_l != d.nl) atomicExch(A.err, 1);
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    const int v = ((k % 6) < 3) ? wave_min_i(acc[k]) : wave_max_i(acc[k]);
    if (lane == 0) { if ((k % 6) < 3) atomicMin(&s_acc[k], v); else atomicMax(&s_acc[k], v); }
  }
  __syncthreads();
  if (threadIdx.x < 24) {
    const int k = threadIdx.x, side = k / 12, what = (k % 12) / 3, a = k % 3;     // what: 0 box lo, 1 box hi, 2 centroid lo, 3 centroid hi
    const int child = side ? ch.y : ch.x;
    int4* arr = (what == 0) ? A.N.blo : (what == 1) ? A.N.bhi : (what == 2) ? A.N.clo : A.N.chi;
    int* p = &reinterpret_cast<int*>(&arr[child])[a];
    if ((what & 1) == 0) atomicMin(p, s_acc[k]); else atomicMax(p, s_acc[k]);
  }
}

// ---- triangle records in reference order: record p = the triangle at final position p, a leaf's triangles in ascending id order --------
__global__ __launch_bounds__(256) void k_emit_tris(Nodes N, int n_nodes, const float4* __restrict__ r0, const float4* __restrict__ r1,
                                                   const float* __restrict__ tri9, float* __restrict__ out) {
  const int node = blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n_nodes || N.child[node].x >= 0) return;
  const int first = N.blo[node].w, count = N.bhi[node].w, depth = N.clo[node].w;
  const float4* r = (depth & 1) ? r1 : r0;
  int ids[kMaxLeafTris];
  const int c = min(count, kMaxLeafTris);
  for (int k = 0; k < c; ++k) ids[k] = __float_as_int(r[first + k].w);
  for (int a = 1; a < c; ++a) { const int v = ids[a]; int b = a - 1; while (b >= 0 && ids[b] > v) { ids[b + 1] = ids[b]; --b; } ids[b + 1] = v; }
  for (int k = 0; k < c; ++k) {
    const float* s = tri9 + 9 * (size_t)ids[k];
    float* o = out + (size_t)kTriFloats * (size_t)(first + k);
    for (int q = 0; q < 9; ++q) o[q] = s[q];
    o[9] = __int_as_float(ids[k]); o[10] = 0.0f; o[11] = 0.0f;
  }
}

// ---- collapse into W-wide nodes (art_bvh.cpp:300-349), breadth-first ---------------------------------------------------------------
struct Item { int n2, n8, stack_before, pad; };
struct Kids { int ch[8]; int nc, n_inner; };

__device__ __forceinline__ float node_area(const Nodes& N, int id) {
  const int4 l = N.blo[id], h = N.bhi[id];
  return half_area(dec(l.x), dec(l.y), dec(l.z), dec(h.x), dec(h.y), dec(h.z));
}

__global__ __launch_bounds__(128) void k_col_pick(Nodes N, const Item* __restrict__ items, int n_items, int width, Kids* __restrict__ kids, int* __restrict__ n_inner) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_items) return;
  const Item it = items[q];
  Kids K; K.nc = 0;
  const int2 top = N.child[it.n2];
  if (top.x < 0) { K.ch[0] = it.n2; K.nc = 1; }
  else { K.ch[0] = top.x; K.ch[1] = top.y; K.nc = 2; }
  while (K.nc < width) {
    int best = -1; float best_a = -1.0f;
    for (int k = 0; k < K.nc; ++k) {
      if (N.child[K.ch[k]].x < 0) continue;
      const float a = node_area(N, K.ch[k]);
      if (a > best_a) { best_a = a; best = k; }
    }
    if (best < 0) break;
    const int2 c = N.child[K.ch[best]];
    for (int k = K.nc; k > best + 1; --k) K.ch[k] = K.ch[k - 1];                   // in-order: BVH2 siblings stay neighbours
    K.ch[best] = c.x; K.ch[best + 1] = c.y; ++K.nc;
  }
  int inner = 0;
  for (int k = 0; k < K.nc; ++k) inner += (N.child[K.ch[k]].x >= 0) ? 1 : 0;
  K.n_inner = inner;
  kids[q] = K; n_inner[q] = K.n_inner;
}

__device__ __forceinline__ float next_dn(float v) { return (v == 0.0f) ? -1.401298464e-45f : __int_as_float(__float_as_int(v) + (v > 0.0f ? -1 : 1)); }
__device__ __forceinline__ float next_up(float v) { return (v == 0.0f) ? 1.401298464e-45f : __int_as_float(__float_as_int(v) + (v > 0.0f ? 1 : -1)); }

__global__ __launch_bounds__(128) void k_col_emit(Nodes N, const Item* __restrict__ items, int n_items, int width, const Kids* __restrict__ kids, const int* __restrict__ off,
                                                  int next_base, Item* __restrict__ next_items, int* max_stack, float* __restrict__ nodes,
                                                  float inflate_rel, float inflate_abs) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_items) return;
  const Item it = items[q];
  const Kids K = kids[q];
  const int stack_here = it.stack_before + K.nc - 1;
  atomicMax(max_stack, stack_here + 1);
  float* nd = nodes + (size_t)it.n8 * (size_t)node_floats(width);
  const int hb = 4 * width;
  int inner_k = 0;
  for (int j = 0; j < width; ++j) {
    int ref = -1, cnt = 0;
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    if (j < K.nc) {
      const int id = K.ch[j];
      const int4 l4 = N.blo[id], h4 = N.bhi[id];
      const float l[3] = {dec(l4.x), dec(l4.y), dec(l4.z)}, h[3] = {dec(h4.x), dec(h4.y), dec(h4.z)};
      for (int a = 0; a < 3; ++a) {
        const float pad = inflate_abs + inflate_rel * fmaxf(fabsf(l[a]), fabsf(h[a]));
        lo[a] = next_dn(l[a] - pad); hi[a] = next_up(h[a] + pad);
      }
      if (N.child[id].x < 0) { ref = l4.w; cnt = h4.w; }                           // leaf: first reference position = first triangle record
      else {
        const int slot = off[q] + inner_k; ++inner_k;
        ref = next_base + slot; cnt = 0;
        Item nx; nx.n2 = id; nx.n8 = ref; nx.stack_before = stack_here; nx.pad = 0;
        next_items[slot] = nx;
      }
    }
    nd[4 * j + 0] = lo[0]; nd[4 * j + 1] = lo[1]; nd[4 * j + 2] = lo[2]; nd[4 * j + 3] = __int_as_float(ref);
    nd[hb + 4 * j + 0] = hi[0]; nd[hb + 4 * j + 1] = hi[1]; nd[hb + 4 * j + 2] = hi[2]; nd[hb + 4 * j + 3] = __int_as_float(cnt);
  }
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

bool build_bvh_sah_gpu(const float* d_tri9, int n, const BvhBuildParams& prm, hipStream_t st, GpuBvh& out, std::string& err) {
  if (n < 2) { err = "build_bvh_sah_gpu needs at least 2 triangles"; return false; }
  if (prm.width != 4 && prm.width != 8) { err = "BVH width must be 4 or 8"; return false; }
  if (prm.sah_bins < 2 || prm.sah_bins > kMaxBins) { err = "GPU SAH builder: 2..64 bins"; return false; }
  if (prm.spatial_alpha >= 0.0f) { err = "GPU SAH builder: spatial splits are a host-builder option (bvh_builder=0)"; return false; }
  SahCost P;
  P.nb = prm.sah_bins; P.width = prm.width; P.max_leaf = std::min(prm.max_leaf, prm.width); P.max_depth = prm.max_sah_depth;
  P.node_cost = prm.node_cost; P.leaf_base = prm.leaf_base; P.tri_cost = prm.tri_cost >= 0.0f ? prm.tri_cost : (prm.width == 4 ? 0.2f : 0.05f);
  const size_t N2 = 2 * (size_t)n;                                                 // BVH2 nodes: at most 2n - 1
  const int max_big = n / kChunk + 2, max_chunks = n / kChunk + 2 * max_big + 2;
  const size_t max_active = (size_t)n / 2 + 1;                                     // an active node holds at least two references (Dec alone is 128 B per entry: ADVICE r3)
  const int bin_words = 3 * P.nb * kBinWords;
  Scratch S;
  float4 *rlo[2], *rhi[2]; Nodes N; Act* act[2]; int2* chunk[2]; Dec* dec; S4 *flags, *offs, *totals; int *gbins, *chunk_cnt, *chunk_off, *misc;
  Item* items[2]; Kids* kids; int *n_inner, *inner_off;
  if (!S.get(&rlo[0], n, err) || !S.get(&rlo[1], n, err) || !S.get(&rhi[0], n, err) || !S.get(&rhi[1], n, err) ||
      !S.get(&N.blo, N2, err) || !S.get(&N.bhi, N2, err) || !S.get(&N.clo, N2, err) || !S.get(&N.chi, N2, err) || !S.get(&N.child, N2, err) ||
      !S.get(&act[0], max_active, err) || !S.get(&act[1], max_active, err) || !S.get(&chunk[0], max_chunks, err) || !S.get(&chunk[1], max_chunks, err) ||
      !S.get(&dec, max_active, err) || !S.get(&flags, max_active, err) || !S.get(&offs, max_active, err) || !S.get(&totals, 1, err) ||
      !S.get(&gbins, (size_t)max_big * bin_words, err) || !S.get(&chunk_cnt, max_chunks, err) || !S.get(&chunk_off, max_chunks, err) || !S.get(&misc, 8, err) ||
      !S.get(&items[0], n, err) || !S.get(&items[1], n, err) || !S.get(&kids, n, err) || !S.get(&n_inner, n, err) || !S.get(&inner_off, n, err))
    return false;
  size_t tmp_a = 0, tmp_b = 0;
  SB_TRY(hipcub::DeviceScan::ExclusiveScan(nullptr, tmp_a, flags, offs, S4Sum(), S4{0, 0, 0, 0}, (int)max_active, st));
  SB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_b, n_inner, inner_off, n, st));
  void* tmp = nullptr;
  const size_t tmp_bytes = std::max(tmp_a, tmp_b);
  if (!S.get((char**)&tmp, tmp_bytes, err)) return false;

  struct Events { hipEvent_t a = nullptr, b = nullptr; ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
  SB_TRY(hipEventCreate(&ev.a)); SB_TRY(hipEventCreate(&ev.b));
  SB_TRY(hipEventRecord(ev.a, st));

  // root
  {
    const int4 h_root[4] = {{kEncPosInf, kEncPosInf, kEncPosInf, 0}, {kEncNegInf, kEncNegInf, kEncNegInf, n}, {kEncPosInf, kEncPosInf, kEncPosInf, 0}, {kEncNegInf, kEncNegInf, kEncNegInf, 0}};
    SB_TRY(hipMemcpyAsync(N.blo, &h_root[0], sizeof(int4), hipMemcpyHostToDevice, st)); SB_TRY(hipMemcpyAsync(N.bhi, &h_root[1], sizeof(int4), hipMemcpyHostToDevice, st));
    SB_TRY(hipMemcpyAsync(N.clo, &h_root[2], sizeof(int4), hipMemcpyHostToDevice, st)); SB_TRY(hipMemcpyAsync(N.chi, &h_root[3], sizeof(int4), hipMemcpyHostToDevice, st));
    SB_TRY(hipMemsetAsync(misc, 0, 8 * sizeof(int), st));
    hipLaunchKernelGGL(k_refs, dim3((n + 255) / 256), dim3(256), 0, st, d_tri9, n, rlo[0], rhi[0], N, misc);
    Act a0; a0.node = 0; a0.big_slot = (n > kChunk) ? 0 : -1; a0.chunk0 = (n > kChunk) ? 0 : -1; a0.pad = 0;
    SB_TRY(hipMemcpyAsync(act[0], &a0, sizeof a0, hipMemcpyHostToDevice, st));
    if (n > kChunk) {
      std::vector<int2> c0((size_t)(n + kChunk - 1) / kChunk);
      for (size_t k = 0; k < c0.size(); ++k) c0[k] = make_int2(0, (int)k);
      SB_TRY(hipMemcpyAsync(chunk[0], c0.data(), c0.size() * sizeof(int2), hipMemcpyHostToDevice, st));
      SB_TRY(hipStreamSynchronize(st));                                            // c0 is a local
    }
  }
  int m = 1, n_chunks = (n > kChunk) ? (n + kChunk - 1) / kChunk : 0, n_big = (n > kChunk) ? 1 : 0, n_nodes2 = 1, level = 0;
  const size_t eval_lds = 4 * (size_t)(bin_words + 6 * P.nb) * sizeof(int);
  while (m > 0) {
    const int cur = level & 1, nxt = cur ^ 1;
    Args A;
    A.N = N; A.rlo_in = rlo[cur]; A.rhi_in = rhi[cur]; A.rlo_out = rlo[nxt]; A.rhi_out = rhi[nxt];
    A.act = act[cur]; A.act_next = act[nxt]; A.chunk = chunk[cur]; A.chunk_next = chunk[nxt];
    A.dec = dec; A.flags = flags; A.offs = offs; A.gbins = gbins; A.chunk_cnt = chunk_cnt; A.chunk_off = chunk_off; A.totals = totals;
    A.P = P; A.node_base = n_nodes2; A.err = misc + 2;
    if (n_big > max_big || n_chunks > max_chunks) { err = "internal: GPU SAH chunk plan out of bounds"; return false; }
    if (n_chunks > 0) {
      const int gw = n_big * bin_words;
      hipLaunchKernelGGL(k_init_bins, dim3((gw + 255) / 256), dim3(256), 0, st, gbins, gw);
      hipLaunchKernelGGL(k_bin_big, dim3(n_chunks), dim3(256), (size_t)bin_words * sizeof(int), st, A);
    }
    hipLaunchKernelGGL(k_eval, dim3((m + 3) / 4), dim3(256), eval_lds, st, A, m);
    size_t tb = tmp_bytes;
    SB_TRY(hipcub::DeviceScan::ExclusiveScan(tmp, tb, flags, offs, S4Sum(), S4{0, 0, 0, 0}, m, st));
    hipLaunchKernelGGL(k_commit, dim3((m + 255) / 256), dim3(256), 0, st, A, m);
    if (n_chunks > 0) {
      hipLaunchKernelGGL(k_part_count, dim3(n_chunks), dim3(256), 0, st, A);
      tb = tmp_bytes;
      SB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, chunk_cnt, chunk_off, n_chunks, st));
      hipLaunchKernelGGL(k_part_write, dim3(n_chunks), dim3(256), 0, st, A);
    }
    S4 t; int h_misc[8];
    SB_TRY(hipMemcpyAsync(&t, totals, sizeof t, hipMemcpyDeviceToHost, st));
    SB_TRY(hipMemcpyAsync(h_misc, misc, sizeof h_misc, hipMemcpyDeviceToHost, st));      // the error words travel with every level's round trip
    SB_TRY(hipStreamSynchronize(st));
    if (h_misc[0]) { err = "non-finite triangle vertex, or a coordinate beyond 1e18 (box extents and the node quantisation need headroom in binary32)"; return false; }
    if (h_misc[2]) { err = "internal: GPU SAH partition does not match its bins"; return false; }
    n_nodes2 += 2 * t.s; m = t.a; n_chunks = t.c; n_big = t.b;
    if ((size_t)n_nodes2 > N2 || (size_t)m > max_active) { err = "internal: GPU SAH node count out of bounds"; return false; }
    if (++level > 4096) { err = "internal: GPU SAH build did not terminate"; return false; }
  }
  // triangle records: position p of the final reference order
  SB_TRY(hipMalloc(&out.tris, (size_t)n * kTriFloats * sizeof(float)));
  hipLaunchKernelGGL(k_emit_tris, dim3((n_nodes2 + 255) / 256), dim3(256), 0, st, N, n_nodes2, rlo[0], rlo[1], d_tri9, out.tris);
  // collapse.  Every wide node opens at least one inner BVH2 node, so there are at most n - 1 of them.
  const size_t node_cap = (size_t)n;
  SB_TRY(hipMalloc(&out.nodes, node_cap * (size_t)node_floats(prm.width) * sizeof(float)));
  const Item root = {0, 0, 0, 0};
  SB_TRY(hipMemcpyAsync(items[0], &root, sizeof root, hipMemcpyHostToDevice, st));
  int* max_stack = misc + 1;
  const int one = 1;
  SB_TRY(hipMemcpyAsync(max_stack, &one, sizeof one, hipMemcpyHostToDevice, st));
  int n_items = 1, n_wide = 1, wlevels = 0;
  while (n_items > 0) {
    const int cur = wlevels & 1;
    hipLaunchKernelGGL(k_col_pick, dim3((n_items + 127) / 128), dim3(128), 0, st, N, items[cur], n_items, prm.width, kids, n_inner);
    size_t tb = tmp_bytes;
    SB_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tb, n_inner, inner_off, n_items, st));
    hipLaunchKernelGGL(k_col_emit, dim3((n_items + 127) / 128), dim3(128), 0, st, N, items[cur], n_items, prm.width, kids, inner_off, n_wide, items[cur ^ 1], max_stack,
                       out.nodes, prm.inflate_rel, prm.inflate_abs);
    int last[2];
    SB_TRY(hipMemcpyAsync(&last[0], inner_off + (n_items - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    SB_TRY(hipMemcpyAsync(&last[1], n_inner + (n_items - 1), sizeof(int), hipMemcpyDeviceToHost, st));
    SB_TRY(hipStreamSynchronize(st));
    n_items = last[0] + last[1];
    n_wide += n_items;
    if ((size_t)n_wide > node_cap) { err = "internal: GPU SAH wide-node count out of bounds"; return false; }   // checked before the next level writes
    if (++wlevels > 4096) { err = "internal: GPU SAH collapse did not terminate"; return false; }
  }
  out.n_nodes = n_wide;
  {
    int h_ms = 1;
    SB_TRY(hipMemcpyAsync(&h_ms, max_stack, sizeof(int), hipMemcpyDeviceToHost, st));
    SB_TRY(hipStreamSynchronize(st));
    out.max_stack = h_ms;
  }
  if (prm.width == 4 && prm.quantise) {
    SB_TRY(hipMalloc(&out.qnodes, (size_t)out.n_nodes * kQNodeBytes));
    launch_quantise_nodes(st, out.nodes, out.n_nodes, out.qnodes);
  }
  SB_TRY(hipEventRecord(ev.b, st));
  SB_TRY(hipEventSynchronize(ev.b));
  float ms = 0.0f;
  SB_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
  SB_TRY(hipGetLastError());
  out.n_tris = n; out.build_ms = ms; out.levels = level;
  return true;
}

}  // namespace art

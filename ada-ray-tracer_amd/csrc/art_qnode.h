// art_qnode.h -- the 64-byte quantised node of the 4-wide tree, the form k_trace_coop reads.
//
// Why: a 4-wide node with binary32 child boxes is 128 B; the same node with the child boxes quantised to 8 bits per plane relative to
// the node's own origin is 64 B -- half the lines a ray pulls through L2 (measured when it went in, round 2: 77.5 -> 69.1 ms per trace
// launch on the 1M-triangle scene) and, in the lane-record form below, ONE 16-byte load per lane and node step.  The search result
// cannot depend on the boxes as long as they stay conservative, so the image and every hit stay bit-identical; only the traversal
// counters move (slightly looser boxes: +2.5 % node visits).
//
//   bytes 16j..16j+15   lane record j (j = 0..3):  { qlo.x qlo.y qlo.z qhi.x | qhi.y qhi.z 0 0 | entry | header word j }
//   header words   0: origin.x   1: origin.y   2: origin.z   3: scale        binary32; scale is a power of two, shared by the three axes
//                  Lane j of a ray's quad reads record j with ONE 16-byte load (the quad reads the node's 64 contiguous bytes) and the
//                  four header words are broadcast inside the quad by DPP.  Per wave load instruction and CU (profiles/ta_rate.hip): quads
//                  reading 64 contiguous bytes ~15 clocks of the vector-memory path; a 16-byte header + 12-byte child records at a
//                  12-byte lane stride 15 + 45.
//   entry          inner child: node_index * 64               (bit 31 clear, low 6 bits clear)
//                  leaf:        0x80000000 | first_triangle * 64 | count   (count 1..4 in the low 4 bits; 64-byte padded triangle records)
//                  empty slot:  0x80000000   (a leaf of zero triangles: should rounding ever let an empty slot's inverted planes
//                               pass the slab test, the leaf step it causes loads nothing and accepts nothing)
//   child box      lo = fma(float(qlo), scale, origin)   hi = fma(float(qhi), scale, origin)      one rounding each
//
// "The tree" everybody else sees (art_export_bvh, the one-ray-per-lane kernels, the host simulation, the oracle's walker) is the
// binary32 tree with exactly these dequantised boxes: quantise_node() rewrites the binary32 packet in place, so all walkers test
// the same boxes and count the same B and T.  Host and device run the same code (+, -, exact power-of-two division, fma, float->int).
#pragma once
#include "art_scene.h"

namespace art {

constexpr int kQNodeBytes = 64;
constexpr uint32_t kQEntryEmpty = 0x80000000u, kQEntryLeaf = 0x80000000u;
// instanced scenes (k_trace_coop<.., INST = true>): a leaf entry whose count field is 15 names an INSTANCE (index in bits 4..30) -- the ray
// enters its mesh's tree there; count 14 is the marker the kernel pushes under that tree: popped, the ray is back in world space
constexpr uint32_t kQCountInstance = 15u, kQCountLeaveInstance = 14u;
constexpr uint32_t kQEntryLeaveInstance = kQEntryLeaf | kQCountLeaveInstance;
constexpr int kTriBytes = kTriFloats * 4;
constexpr int kQTriBytes = 64;       // the 4-wide kernel reads triangle records padded to 64 bytes: a record never straddles a 128-byte L2 line

struct QNode { struct { uint32_t c0, c1, entry; float hdr; } rec[4]; };   // rec[j].hdr: origin.x, origin.y, origin.z, scale
static_assert(sizeof(QNode) == kQNodeBytes, "quantised node is 64 bytes");

ART_HD float q_pow2_at_least(float r) {            // smallest power of two >= r  (r > 0, finite, normal)
  const uint32_t b = __builtin_bit_cast(uint32_t, r);
  const uint32_t e = (b & 0x007fffffu) ? ((b >> 23) + 1u) : (b >> 23);
  return __builtin_bit_cast(float, e << 23);
}

// nd: binary32 packet of a 4-wide node (art_scene.h layout, 32 floats).  Fills q and replaces the valid child boxes of nd by their
// dequantised form.  The dequantised box always encloses the input box.
ART_HD void quantise_node(float* nd, QNode& q) {
  constexpr int W = 4;
  float o[3] = {0.0f, 0.0f, 0.0f}, ext = 0.0f;
  bool any = false;
  for (int j = 0; j < W; ++j) {
    if (__builtin_bit_cast(int32_t, nd[4 * j + 3]) < 0) continue;
    for (int a = 0; a < 3; ++a) o[a] = any ? ((nd[4 * j + a] < o[a]) ? nd[4 * j + a] : o[a]) : nd[4 * j + a];
    any = true;
  }
  for (int j = 0; j < W; ++j) {
    if (__builtin_bit_cast(int32_t, nd[4 * j + 3]) < 0) continue;
    for (int a = 0; a < 3; ++a) { const float e = nd[4 * W + 4 * j + a] - o[a]; ext = (e > ext) ? e : ext; }
  }
  float s = q_pow2_at_least(((ext > 1.0e-30f) ? ext : 1.0e-30f) / 255.0f);
  uint32_t ql[W][3], qh[W][3];
  for (;;) {                                        // at most a few rounds: one doubling of the scale halves every q
    bool fits = true;
    for (int j = 0; j < W && fits; ++j) {
      if (__builtin_bit_cast(int32_t, nd[4 * j + 3]) < 0) continue;
      for (int a = 0; a < 3; ++a) {
        const float lo = nd[4 * j + a], hi = nd[4 * W + 4 * j + a];
        int32_t k = (int32_t)((lo - o[a]) / s);
        k = k < 0 ? 0 : (k > 255 ? 255 : k);
        while (k > 0 && __builtin_fmaf((float)k, s, o[a]) > lo) --k;
        int32_t m = (int32_t)((hi - o[a]) / s);
        m = m < 0 ? 0 : m;
        while (m <= 255 && __builtin_fmaf((float)m, s, o[a]) < hi) ++m;
        if (m > 255) { fits = false; break; }
        ql[j][a] = (uint32_t)k; qh[j][a] = (uint32_t)m;
      }
    }
    if (fits) break;
    s = s * 2.0f;
  }
  q.rec[0].hdr = o[0]; q.rec[1].hdr = o[1]; q.rec[2].hdr = o[2]; q.rec[3].hdr = s;
  for (int j = 0; j < W; ++j) {
    const int32_t ref = __builtin_bit_cast(int32_t, nd[4 * j + 3]), cnt = __builtin_bit_cast(int32_t, nd[4 * W + 4 * j + 3]);
    uint32_t w0, w1, w2;
    if (ref < 0) { w0 = 0x00ffffffu; w1 = 0u; w2 = kQEntryEmpty; }   // lo = 255, hi = 0
    else {
      w0 = ql[j][0] | (ql[j][1] << 8) | (ql[j][2] << 16) | (qh[j][0] << 24);
      w1 = qh[j][1] | (qh[j][2] << 8);
      w2 = cnt ? (kQEntryLeaf | ((uint32_t)ref * (uint32_t)kQTriBytes) | (uint32_t)cnt) : ((uint32_t)ref * (uint32_t)kQNodeBytes);
    }
    q.rec[j].c0 = w0; q.rec[j].c1 = w1; q.rec[j].entry = w2;
    if (ref < 0) continue;
    for (int a = 0; a < 3; ++a) {
      nd[4 * j + a] = __builtin_fmaf((float)ql[j][a], s, o[a]);
      nd[4 * W + 4 * j + a] = __builtin_fmaf((float)qh[j][a], s, o[a]);
    }
  }
}

}  // namespace art

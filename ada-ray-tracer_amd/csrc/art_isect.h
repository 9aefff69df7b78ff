// art_isect.h -- ray / primitive intersection with the reference's arithmetic, restructured around a
// single running candidate (t, key, u, v) so that the closest hit is an order-independent
// lexicographic minimum (needed by a BVH whose traversal order differs from the reference's scan).
//   spheres      geometry.adb:48-115      Cornell box geometry.adb:146-229
//   flat light   geometry.adb:118-143     triangle    geometry.adb:231-263
//   brute-force mesh (reference quirk mode) geometry.adb:266-323
#pragma once
#include "art_scene.h"

// Diagnostic builds only (-DART_LANE_PROBE; build line and reader: profiles/lane_probe.py): ART_PROBE(k) adds, for every wave passing the point, its number of
// enabled lanes to g_lane_probe[2k] and 1 to g_lane_probe[2k + 1] -- where a stage loses its lanes (round 4: k_shade_compact ran at 31 of
// 64 lanes per VALU instruction).  Expands to nothing in the product build.
#if defined(ART_TIME_PROBE) && !defined(ART_LANE_PROBE)
#define ART_LANE_PROBE 1
#endif
#if defined(ART_LANE_PROBE) && defined(__HIPCC__)
static __device__ unsigned long long g_lane_probe[2 * 96];
static __device__ int g_lane_probe_on;        // set around the launches of the kernel under study (k_shade_compact)
#endif
#if defined(ART_LANE_PROBE) && defined(__HIP_DEVICE_COMPILE__)
#define ART_PROBE(k) do { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); \
    if (g_lane_probe_on && (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == __builtin_ctzll(m_)) { \
      atomicAdd(&g_lane_probe[2 * (k)], (unsigned long long)__builtin_popcountll(m_)); atomicAdd(&g_lane_probe[2 * (k) + 1], 1ull); } } while (0)
#else
#define ART_PROBE(k) do { } while (0)
#endif
// -DART_TIME_PROBE (round 5, profiles/time_probe.py): ART_TPROBE(tp, k) adds the shader-clock cycles the wave spent since its previous time
// probe to g_lane_probe[2k] (and 1 to [2k + 1]); tp: the wave's own word in LDS holding the time of that previous probe.  Where a wave of
// k_shade_compact spends its wall time -- waiting for its loads, for the store queue, at the barriers, or issuing instructions.
#if defined(ART_TIME_PROBE) && defined(__HIP_DEVICE_COMPILE__)
// (tp[0]: the wave's previous probe time; the sums go to the workgroup's LDS table tp_acc -- flushed to g_lane_probe once per workgroup:
// a global atomic per probe and wave made the probes themselves the slowest thing in the kernel)
#define ART_TPROBE(tp, k) do { if ((tp) != nullptr && g_lane_probe_on) { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); \
    if ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == __builtin_ctzll(m_)) { const unsigned long long now_ = __builtin_readcyclecounter(); \
      unsigned long long* const acc_ = *(unsigned long long* volatile*)((tp) + 1); \
      atomicAdd(&acc_[2 * ((k) - 64)], now_ - *(volatile unsigned long long*)(tp)); atomicAdd(&acc_[2 * ((k) - 64) + 1], 1ull); *(volatile unsigned long long*)(tp) = now_; } } } while (0)
#else
#define ART_TPROBE(tp, k) do { } while (0)
#endif


namespace art {

template <class P> ART_HD f3 ld3(P p) { return mk3(p[0], p[1], p[2]); }      // (P: any pointer to floats, LDS-qualified ones included)
template <class P> ART_HD DevSphere load_sphere(P p, int i) { DevSphere r; r.x = p[i].x; r.y = p[i].y; r.z = p[i].z; r.r = p[i].r; return r; }

struct Cand { float t; uint32_t key; float u, v; };

ART_HD Cand cand_init(float tfar) { Cand c; c.t = tfar; c.key = KEY_MISS; c.u = 0.0f; c.v = 0.0f; return c; }

// strict '<' of scene.adb:73 / geometry.adb:72-78 generalised to (t, key); an equal t can only
// displace an existing hit of higher key, never the initial bound.
ART_HD bool cand_wins(float t, uint32_t key, const Cand& b) {
  return (t < b.t) || (t == b.t && b.key != KEY_MISS && key < b.key);
}

ART_HD void cand_take(Cand& b, float t, uint32_t key, float u, float v) {
  if (cand_wins(t, key, b)) { b.t = t; b.key = key; b.u = u; b.v = v; }
}

// one sphere of IntersectAllSpheres: candidate root = t1 if t1 > 0 else t2 if t2 > 0  (t2 >= t1)
ART_HD void isect_sphere(f3 o, f3 d, const DevSphere& s, uint32_t index, Cand& best) {
  const f3 k = o - mk3(s.x, s.y, s.z);
  const float b = dot(k, d);
  const float c = dot(k, k) - s.r * s.r;
  const float disc = b * b - c;
  ART_PROBE(50);
  if (disc >= 0.0f) {
    ART_PROBE(51);
    const float sq = sqrtf(disc);
    const float t1 = -b - sq, t2 = -b + sq;
    if (t1 > 0.0f) { if (t1 < kInfinity) cand_take(best, t1, KEY_SPHERE | index, 0.0f, 0.0f); }
    else if (t2 > 0.0f) { if (t2 < kInfinity) cand_take(best, t2, KEY_SPHERE | index, 0.0f, 0.0f); }
  }
}

// IntersectBox, verbatim slab arithmetic (compare-select min/max, 1/dir may be +-inf).  rcp = (1/d.x, 1/d.y, 1/d.z): the caller may have
// them already (ray_rcp: a ray's three reciprocals are needed by the Cornell box, by the brute-force mesh's box and by the BVH's slab
// set-up -- the same IEEE divisions of the same operands, done once per ray instead of up to three times)
ART_HD f3 ray_rcp(f3 d) { return mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }
ART_HD bool slab_reference(f3 o, f3 rcp, const float* bmin, const float* bmax, float& tmin, float& tmax) {
  const float ix = rcp.x, iy = rcp.y, iz = rcp.z;
  const float lo = (bmax[0] - o.x) * ix, hi = (bmin[0] - o.x) * ix;
  const float lo1 = (bmax[1] - o.y) * iy, hi1 = (bmin[1] - o.y) * iy;
  const float lo2 = (bmax[2] - o.z) * iz, hi2 = (bmin[2] - o.z) * iz;
  tmin = amin(lo, hi); tmax = amax(lo, hi);
  tmin = amax(tmin, amin(lo1, hi1)); tmax = amin(tmax, amax(lo1, hi1));
  tmin = amax(tmin, amin(lo2, hi2)); tmax = amin(tmax, amax(lo2, hi2));
  return (tmax > 0.0f) && (tmin <= tmax);
}

// IntersectCornellBox: exit distance, face by |p - bound| < 1e-5 with later faces overriding, open face 5
ART_HD void isect_cornell(f3 o, f3 d, f3 rcp, const DevScene& s, Cand& best) {
  float tmin, tmax;
  ART_PROBE(52);
  if (!slab_reference(o, rcp, s.cb_min, s.cb_max, tmin, tmax)) return;
  ART_PROBE(53);
  const f3 p = o + tmax * d;
  const float eps = 1.0e-5f;
  uint32_t plane = 0;
  if (fabsf(p.x - s.cb_min[0]) < eps) plane = 0;
  if (fabsf(p.x - s.cb_max[0]) < eps) plane = 1;
  if (fabsf(p.y - s.cb_min[1]) < eps) plane = 2;
  if (fabsf(p.y - s.cb_max[1]) < eps) plane = 3;
  if (fabsf(p.z - s.cb_min[2]) < eps) plane = 4;
  if (fabsf(p.z - s.cb_max[2]) < eps) plane = 5;
  if (plane != 5) cand_take(best, tmax, KEY_CORNELL | plane, 0.0f, 0.0f);
}

// IntersectFlatLight for rect light `index`
template <class L>      // L: pointer to the light (generic or LDS)
ART_HD void isect_quad(f3 o, f3 d, L l, uint32_t index, Cand& best) {
  const float inv_y = 1.0f / d.y;
  const float t = (l->boxMax[1] - o.y) * inv_y;
  const f3 hp = o + t * d;
  const bool hit = (hp.x > l->boxMin[0]) && (hp.x < l->boxMax[0]) && (hp.z > l->boxMin[2]) && (hp.z < l->boxMax[2]) && (t >= 0.0f);
  if (hit) cand_take(best, t, KEY_QUAD | index, 0.0f, 0.0f);
}

// IntersectTriangle without the window test: returns whether (v>0, u>0, u+v<1) holds and the raw t,u,v.
// invDet = 1/max(det, 1e-25) rejects back faces exactly as the reference does.
ART_HD bool tri_raw(f3 o, f3 d, f3 A, f3 B, f3 C, float& t, float& u, float& v) {
  const f3 e1 = B - A, e2 = C - A;
  const f3 pv = cross(d, e2);
  const f3 tv = o - A;
  const f3 qv = cross(tv, e1);
  const float inv = 1.0f / amax(dot(e1, pv), 1.0e-25f);
  v = dot(tv, pv) * inv;
  u = dot(qv, d) * inv;
  t = dot(e2, qv) * inv;
  return (v > 0.0f) && (u > 0.0f) && (u + v < 1.0f);
}

// IntersectMeshBF with its (tmin, tmax) window semantics: triangles in index order, the window
// collapses to (t, t+1e-6) after every accepted hit ("first hit wins").  One candidate results.
ART_HD void isect_bf_mesh(f3 o, f3 d, f3 rcp, const DevScene& s, Cand& best) {
  if (s.bf_ntris <= 0) return;
  float bt0, bt1;
  if (!slab_reference(o, rcp, s.bf_bbmin, s.bf_bbmax, bt0, bt1)) return;
  float wmin = 0.0f, wmax = 1000000.0f;
  bool any = false; uint32_t tri_id = 0; float ht = 0.0f, hu = 0.0f, hv = 0.0f;
  for (int i = 0; i < s.bf_ntris; ++i) {
    const int32_t* ix = s.bf_idx + 3 * i;
    float t, u, v;
    if (tri_raw(o, d, ld3(s.bf_pos + 3 * ix[0]), ld3(s.bf_pos + 3 * ix[1]), ld3(s.bf_pos + 3 * ix[2]), t, u, v) && t > wmin && t < wmax) {
      any = true; tri_id = (uint32_t)i; ht = t; hu = u; hv = v;
      wmin = t; wmax = t + 1.0e-6f;
    }
  }
  if (any) cand_take(best, ht, KEY_BFTRI | tri_id, hu, hv);
}

// ---- BVH8 traversal, one ray per caller (the cooperative 8-lane kernel lives in art_kernels.hip).
// Published order (DESIGN.md "Traversal order"): pop; drop if entry.tmin > best.t; inner node: test
// the valid children against [max(.,0), min(., best.t)], sort hits ascending by
// key = (bits(tmin) & ~7) | slot, push far-to-near; leaf: test its triangles in storage order.
struct BvhStats { uint64_t box_tests, tri_tests, node_visits, leaf_visits; };

// Slab test of the BVH (this backend's own arithmetic, not reference code; the oracle's walker mirrors it for the
// B/T counters).  Per ray: inv = 1/d with |d| < 1e-30 replaced by +-1e-30 (so inv is finite and an axis-parallel ray
// behaves like the limit of a slightly tilted one), noi = -(o * inv).  Per box plane: t = fma(plane, inv, noi) -- one
// fused op per plane, 6 per box.  The interval is clipped to [0, tbest].
ART_HD void slab_setup(f3 o, f3 d, f3& inv, f3& noi) {
  const float tiny = 1.0e-30f;
  const float dx = (fabsf(d.x) < tiny) ? copysignf(tiny, d.x) : d.x;
  const float dy = (fabsf(d.y) < tiny) ? copysignf(tiny, d.y) : d.y;
  const float dz = (fabsf(d.z) < tiny) ? copysignf(tiny, d.z) : d.z;
  inv = mk3(1.0f / dx, 1.0f / dy, 1.0f / dz);
  noi = mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
}
// the same from the ray's reciprocals (ray_rcp): 1 / +-tiny is the constant +-(1 / tiny), every other component IS the reciprocal -- no division
ART_HD void slab_setup_rcp(f3 o, f3 d, f3 rcp, f3& inv, f3& noi) {
  const float tiny = 1.0e-30f, big = 1.0f / tiny;
  inv = mk3((fabsf(d.x) < tiny) ? copysignf(big, d.x) : rcp.x, (fabsf(d.y) < tiny) ? copysignf(big, d.y) : rcp.y, (fabsf(d.z) < tiny) ? copysignf(big, d.z) : rcp.z);
  noi = mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
}

ART_HD void slab_interval(f3 lo, f3 hi, f3 inv, f3 noi, float tbest, float& tmn, float& tmx) {
  const float t0x = __builtin_fmaf(lo.x, inv.x, noi.x), t1x = __builtin_fmaf(hi.x, inv.x, noi.x);
  const float t0y = __builtin_fmaf(lo.y, inv.y, noi.y), t1y = __builtin_fmaf(hi.y, inv.y, noi.y);
  const float t0z = __builtin_fmaf(lo.z, inv.z, noi.z), t1z = __builtin_fmaf(hi.z, inv.z, noi.z);
  tmn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
  tmx = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tbest));
}

ART_HD void slab_fast(const float* nd, int width, int j, f3 inv, f3 noi, float tbest, float& tmn, float& tmx) {
  slab_interval(ld3(nd + 4 * j), ld3(nd + 4 * width + 4 * j), inv, noi, tbest, tmn, tmx);
}

ART_HD void tri_leaf_test(const float* tr, f3 o, f3 d, Cand& best) {
  float t, u, v;
  if (tri_raw(o, d, ld3(tr), ld3(tr + 3), ld3(tr + 6), t, u, v) && t > 0.0f && t < 1000000.0f)
    cand_take(best, t, KEY_TRI | (uint32_t)__builtin_bit_cast(int32_t, tr[9]), u, v);
}

ART_HD float next_up_pos(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) + 1u); }   // x >= 0, finite

// Visibility rule for shadow rays (shm >= 0): Compute_Shadow only asks whether the closest hit lies in (shm, tfar)
// (ray_tracer.adb:122, shm = 10*eps).  The first hit found beyond shm ("far") already bounds the closest hit from
// above, so from then on only hits with t <= shm can change the answer: the search bound collapses to shm, and the
// first hit found there ("near") ends the ray.  `rep` is the hit reported when no near hit exists.
struct ShadowState { float shm; bool far; Cand rep; };

// returns true when the ray is finished
ART_HD bool shadow_rule(ShadowState& sh, Cand& best) {
  if (sh.shm < 0.0f || best.key == KEY_MISS) return false;
  if (best.t <= sh.shm) return true;                       // near hit: not in shadow, done
  if (!sh.far) { sh.far = true; sh.rep = best; best = cand_init(next_up_pos(sh.shm)); }
  return false;
}

// The published walk over a tree of binary32 node packets, with the leaf left to the caller: leaf(ref, cnt) tests triangle records
// [ref, ref + cnt) and updates `best` (bvh_closest: the records ARE the triangles; the instanced render search of art_instanced.h walks a
// mesh's tree with the object-space ray and tests the triangles in world space).
template <bool STATS, class Leaf>
ART_HD void bvh_walk(const float* nodes, int W, f3 o, f3 d, Cand& best, BvhStats* st, ShadowState& sh, int32_t root, Leaf leaf) {
  f3 inv, noi;
  slab_setup(o, d, inv, noi);
  int32_t stk_ref[kStackEntries]; float stk_t[kStackEntries];
  int sp = 0;
  stk_ref[sp] = root; stk_t[sp] = 0.0f; ++sp;     // entry = (ref << 4) | count; root: 0 = node 0, or where an instance's entry point starts (a subtree, even a leaf)
  while (sp > 0) {
    --sp;
    const int32_t e = stk_ref[sp];
    if (stk_t[sp] > best.t) continue;
    const int32_t ref = e >> 4, cnt = e & 15;
    if (cnt == 0) {
      const float* nd = nodes + (size_t)ref * (size_t)node_floats(W);
      uint32_t key[8]; int32_t ent[8]; float tm[8]; int nh = 0;
      if (STATS) st->node_visits++;
      for (int j = 0; j < W; ++j) {
        const int32_t rj = __builtin_bit_cast(int32_t, nd[4 * j + 3]);
        if (rj < 0) continue;
        if (STATS) st->box_tests++;
        float tmn, tmx;
        slab_fast(nd, W, j, inv, noi, best.t, tmn, tmx);
        if (tmn <= tmx) {
          key[nh] = (__builtin_bit_cast(uint32_t, tmn) & ~7u) | (uint32_t)j;
          ent[nh] = (rj << 4) | __builtin_bit_cast(int32_t, nd[4 * W + 4 * j + 3]);
          tm[nh] = tmn; ++nh;
        }
      }
      for (int a = 1; a < nh; ++a) {                // ascending insertion sort by key
        const uint32_t k = key[a]; const int32_t ee = ent[a]; const float tt = tm[a];
        int b = a - 1;
        while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; ent[b + 1] = ent[b]; tm[b + 1] = tm[b]; --b; }
        key[b + 1] = k; ent[b + 1] = ee; tm[b + 1] = tt;
      }
      for (int a = nh - 1; a >= 0; --a) { stk_ref[sp] = ent[a]; stk_t[sp] = tm[a]; ++sp; }
    } else {
      if (STATS) { st->leaf_visits++; st->tri_tests += (uint64_t)cnt; }
      leaf(ref, cnt);                                                                                      // the leaf is one unit
      if (shadow_rule(sh, best)) return;
    }
  }
}
template <bool STATS>
ART_HD void bvh_closest(const DevScene& s, f3 o, f3 d, Cand& best, BvhStats* st, ShadowState& sh) {
  if (s.n_tris <= 0) return;
  bvh_walk<STATS>(s.nodes, s.node_width, o, d, best, st, sh, 0, [&](int32_t ref, int32_t cnt) {
    for (int j = 0; j < cnt; ++j) tri_leaf_test(s.tris + (size_t)(ref + j) * kTriFloats, o, d, best);
  });
}

// (art_instanced.h: the same search over an instanced scene -- DevScene::n_inst > 0; every user of closest_hit includes it)
template <bool STATS> ART_HD void instanced_render_closest(const DevScene& S, f3 o, f3 d, Cand& best, BvhStats* st, ShadowState& sh);

// Scene.Find_Closest_Hit (scene.adb:56-86) for one ray; tfar clips the search (shadow rays only need
// hits below maxDist - epsilon2, ray_tracer.adb:119-122; camera/bounce rays pass kInfinity).
template <bool STATS>
ART_HD Cand closest_hit(const DevScene& s, f3 o, f3 d, float tfar, BvhStats* st, float shm = -1.0f) {
  Cand best = cand_init(tfar);
  for (int i = 0; i < s.n_spheres; ++i) isect_sphere(o, d, s.spheres[i], (uint32_t)i, best);
  const f3 rcp = ray_rcp(d);
  if (s.has_cornell) isect_cornell(o, d, rcp, s, best);
  for (int i = 0; i < s.n_lights; ++i)
    if (s.lights[i].shape == LIGHT_RECT) isect_quad(o, d, s.lights + i, (uint32_t)i, best);
  isect_bf_mesh(o, d, rcp, s, best);
  ShadowState sh; sh.shm = shm; sh.far = false; sh.rep = best;
  if (shadow_rule(sh, best)) return best;
  if (s.n_inst > 0) instanced_render_closest<STATS>(s, o, d, best, st, sh);
  else bvh_closest<STATS>(s, o, d, best, st, sh);
  return (best.key != KEY_MISS || !sh.far) ? best : sh.rep;
}

}  // namespace art

// art_instanced.h -- two-level closest-hit search for the legacy geometry-core seam: one bottom-level tree per mesh (object space),
// one top-level tree over the instances' world boxes, the ray taken into object space by the instance's inverse 3x4 at the boundary.
// This is what Embree builds for embree_connect.cpp:147-184 (one RTCScene per mesh + rtcNewGeometry(RTC_GEOMETRY_TYPE_INSTANCE)):
// memory and build time are O(meshes' triangles + instances) instead of O(instances x triangles) for the flattened upload.
//
// The direction is transformed without renormalising, so t means the same thing in both spaces and the running bound carries over.
// Both levels reuse the 4-wide binary32 node packets and the published walk of art_isect.h; the triangle test is the reference's
// one-sided Moeller-Trumbore (geometry.adb:231-263) in OBJECT space, each triangle stored in both windings (Embree is two-sided).
// Host + device (ART_HD): the host simulation of the tests runs the same code.
#pragma once
#include "art_isect.h"

namespace art {

struct InstRec {
  float minv[12];              // world -> object, 3x4 row-major
  int32_t node_base;           // first node of the mesh's tree in blas_nodes (in nodes)
  int32_t tri_base;            // first triangle record of the mesh in blas_tris (in records)
  int32_t n_tris;              // records of the mesh (2 per input triangle)
  int32_t mesh;
};

struct InstScene {
  const float* tlas_nodes;     // 4-wide packets; a leaf's triangle records are the proxies below
  const float* tlas_tris;      // one proxy record per instance: prim = instance index (its corners only span the instance's world box)
  const float* blas_nodes;     // every mesh's tree, node references relative to the mesh's node_base
  const float* blas_tris;
  const InstRec* inst;
  int32_t n_inst, width;
};

constexpr int kInstTopStack = 96;   // top-level traversal stack of instanced_closest; build_two_level_host refuses a deeper instance tree

struct InstHit { float t; int32_t inst, prim; float u, v; };   // inst < 0: miss; prim = record index inside the mesh (2k front, 2k+1 back)

ART_HD f3 xform_dir(const float* m, f3 v) { return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z); }

ART_HD InstHit instanced_closest(const InstScene& T, f3 o, f3 d, float tfar) {
  InstHit best; best.t = tfar; best.inst = -1; best.prim = -1; best.u = 0.0f; best.v = 0.0f;
  if (T.n_inst <= 0) return best;
  f3 inv, noi;
  slab_setup(o, d, inv, noi);
  constexpr int kTop = kInstTopStack;   // the build checked tlas.max_stack <= kTop and every mesh tree's bound <= kStackEntries (bvh_closest's private stack)
  int32_t stk_ref[kTop]; float stk_t[kTop];
  int sp = 0;
  stk_ref[sp] = 0; stk_t[sp] = 0.0f; ++sp;
  const int W = T.width;
  while (sp > 0) {
    --sp;
    const int32_t e = stk_ref[sp];
    if (stk_t[sp] > best.t) continue;
    const int32_t ref = e >> 4, cnt = e & 15;
    if (cnt == 0) {
      const float* nd = T.tlas_nodes + (size_t)ref * (size_t)node_floats(W);
      uint32_t key[8]; int32_t ent[8]; float tm[8]; int nh = 0;
      for (int j = 0; j < W; ++j) {
        const int32_t rj = __builtin_bit_cast(int32_t, nd[4 * j + 3]);
        if (rj < 0) continue;
        float tmn, tmx;
        slab_fast(nd, W, j, inv, noi, best.t, tmn, tmx);
        if (tmn <= tmx) {
          key[nh] = (__builtin_bit_cast(uint32_t, tmn) & ~7u) | (uint32_t)j;
          ent[nh] = (rj << 4) | __builtin_bit_cast(int32_t, nd[4 * W + 4 * j + 3]);
          tm[nh] = tmn; ++nh;
        }
      }
      for (int a = 1; a < nh; ++a) {
        const uint32_t k = key[a]; const int32_t ee = ent[a]; const float tt = tm[a];
        int b = a - 1;
        while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; ent[b + 1] = ent[b]; tm[b + 1] = tm[b]; --b; }
        key[b + 1] = k; ent[b + 1] = ee; tm[b + 1] = tt;
      }
      for (int a = nh - 1; a >= 0 && sp < kTop; --a) { stk_ref[sp] = ent[a]; stk_t[sp] = tm[a]; ++sp; }
    } else {
      for (int j = 0; j < cnt; ++j) {
        const int32_t ii = __builtin_bit_cast(int32_t, T.tlas_tris[(size_t)(ref + j) * kTriFloats + 9]);
        const InstRec& R = T.inst[ii];
        const f3 oo = xform_point(R.minv, o), dd = xform_dir(R.minv, d);
        DevScene view;                                          // only these four members are read by bvh_closest
        view.node_width = W; view.n_tris = R.n_tris;
        view.nodes = T.blas_nodes + (size_t)R.node_base * (size_t)node_floats(W);
        view.tris = T.blas_tris + (size_t)R.tri_base * kTriFloats;
        Cand c = cand_init(best.t);
        ShadowState sh; sh.shm = -1.0f; sh.far = false; sh.rep = c;
        bvh_closest<false>(view, oo, dd, c, nullptr, sh);
        if (c.key != KEY_MISS && (c.t < best.t || (c.t == best.t && best.inst >= 0 && ii < best.inst))) {
          best.t = c.t; best.inst = ii; best.prim = (int32_t)(c.key & KEY_INDEX_MASK); best.u = c.u; best.v = c.v;
        }
      }
    }
  }
  return best;
}

// ---- the render loop's search over an instanced scene (round 5).  Same two-level walk, but with the semantics of the FLATTENED scene, bit
// for bit: a mesh's tree is walked with the ray taken into object space (boxes only -- they are padded for it at build time), and every
// triangle that survives is tested in WORLD space, its corners transformed by the instance's matrix with the arithmetic the flattening
// uses (xform_point), by the reference's one-sided Moeller-Trumbore on the untransformed ray.  t, u, v are therefore the flattened
// scene's; the closest hit is the lexicographic minimum over (t, key) as everywhere, key index = instance << inst_shift | triangle --
// the order of the flattened triangle list.  `best` comes in as the starting bound (analytic primitives); the shadow rule as in bvh_closest.
template <bool STATS>
ART_HD void instanced_render_closest(const DevScene& S, f3 o, f3 d, Cand& best, BvhStats* st, ShadowState& sh) {
  if (S.n_inst <= 0) return;
  f3 inv, noi;
  slab_setup(o, d, inv, noi);
  constexpr int kTop = kInstTopStack;
  int32_t stk_ref[kTop]; float stk_t[kTop];
  int sp = 0;
  stk_ref[sp] = 0; stk_t[sp] = 0.0f; ++sp;
  constexpr int W = 4;
  while (sp > 0) {
    --sp;
    const int32_t e = stk_ref[sp];
    if (stk_t[sp] > best.t) continue;
    const int32_t ref = e >> 4, cnt = e & 15;
    if (cnt == 0) {
      const float* nd = S.tlas_nodes + (size_t)ref * (size_t)node_floats(W);
      uint32_t key[4]; int32_t ent[4]; float tm[4]; int nh = 0;
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
      for (int a = 1; a < nh; ++a) {
        const uint32_t k = key[a]; const int32_t ee = ent[a]; const float tt = tm[a];
        int b = a - 1;
        while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; ent[b + 1] = ent[b]; tm[b + 1] = tm[b]; --b; }
        key[b + 1] = k; ent[b + 1] = ee; tm[b + 1] = tt;
      }
      for (int a = nh - 1; a >= 0 && sp < kTop; --a) { stk_ref[sp] = ent[a]; stk_t[sp] = tm[a]; ++sp; }
    } else {
      for (int j = 0; j < cnt; ++j) {
        const int32_t ee = __builtin_bit_cast(int32_t, S.tlas_tris[(size_t)(ref + j) * kTriFloats + 9]);      // the proxy names an ENTRY POINT (art_scene.h DevInstance)
        const DevInstance& R = S.inst[ee];
        const f3 oo = xform_point(R.minv, o), dd = xform_dir(R.minv, d);
        const float* const tris = S.blas_tris + (size_t)R.tri_base * kTriFloats;
        const uint32_t key_base = KEY_TRI | ((uint32_t)R.inst << S.inst_shift);
        bvh_walk<STATS>(S.blas_nodes + (size_t)R.node_base * (size_t)node_floats(W), W, oo, dd, best, st, sh, R.root_entry, [&](int32_t r0, int32_t c0) {
          for (int q = 0; q < c0; ++q) {
            const float* tr = tris + (size_t)(r0 + q) * kTriFloats;
            const f3 A = xform_point(R.m, ld3(tr)), B = xform_point(R.m, ld3(tr + 3)), C = xform_point(R.m, ld3(tr + 6));
            float t, u, v;
            if (tri_raw(o, d, A, B, C, t, u, v) && t > 0.0f && t < 1000000.0f)
              cand_take(best, t, key_base | (uint32_t)__builtin_bit_cast(int32_t, tr[9]), u, v);
          }
        });
        if (sh.shm >= 0.0f && best.key != KEY_MISS && best.t <= sh.shm) return;      // shadow rule: a near hit ended the ray inside the mesh's tree
      }
    }
  }
}

}  // namespace art

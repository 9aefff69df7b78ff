// tree_lab.cpp -- EXPERIMENT harness (profiles/tree_lab.py): walks an exported 4-wide tree in the published order and counts what a
// variant of the trace kernel's leaf phase would do.  Not product code, not test code.
//   lab_pair_stats: leaf steps when a popped leaf may take the NEXT stack entry with it if that is a leaf too, is not culled, and both
//   together hold <= width triangles (k_trace_coop round 3: "two leaves per leaf step").
#include <cstdint>
#include <cstring>
#include "../ada-ray-tracer_amd/csrc/art_isect.h"
using namespace art;

extern "C" void lab_pair_stats(const float* nodes, const float* tris, int n_tris, int W, const float* o, const float* d, const float* tfar, long long n, unsigned long long* out /*8*/) {
  unsigned long long node_v = 0, leaf_v = 0, leaf_steps = 0, pairs = 0, tri_t = 0, wasted = 0, next_is_leaf = 0, hist[5] = {0, 0, 0, 0, 0};
#pragma omp parallel for reduction(+ : node_v, leaf_v, leaf_steps, pairs, tri_t, wasted, next_is_leaf) schedule(dynamic, 256)
  for (long long i = 0; i < n; ++i) {
    const f3 oo = mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), dd = mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Cand best = cand_init(tfar[i]);
    f3 inv, noi; slab_setup(oo, dd, inv, noi);
    int32_t stk_ref[kStackEntries]; float stk_t[kStackEntries]; int sp = 0;
    stk_ref[sp] = 0; stk_t[sp] = 0.0f; ++sp;
    while (sp > 0) {
      --sp;
      const int32_t e = stk_ref[sp];
      if (stk_t[sp] > best.t) continue;
      const int32_t ref = e >> 4, cnt = e & 15;
      if (cnt == 0) {
        const float* nd = nodes + (size_t)ref * (size_t)node_floats(W);
        uint32_t key[8]; int32_t ent[8]; float tm[8]; int nh = 0;
        ++node_v;
        for (int j = 0; j < W; ++j) {
          const int32_t rj = __builtin_bit_cast(int32_t, nd[4 * j + 3]);
          if (rj < 0) continue;
          float tmn, tmx; slab_fast(nd, W, j, inv, noi, best.t, tmn, tmx);
          if (tmn <= tmx) { key[nh] = (__builtin_bit_cast(uint32_t, tmn) & ~7u) | (uint32_t)j; ent[nh] = (rj << 4) | __builtin_bit_cast(int32_t, nd[4 * W + 4 * j + 3]); tm[nh] = tmn; ++nh; }
        }
        for (int a = 1; a < nh; ++a) { const uint32_t k = key[a]; const int32_t ee = ent[a]; const float tt = tm[a]; int b = a - 1;
          while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; ent[b + 1] = ent[b]; tm[b + 1] = tm[b]; --b; } key[b + 1] = k; ent[b + 1] = ee; tm[b + 1] = tt; }
        for (int a = nh - 1; a >= 0; --a) { stk_ref[sp] = ent[a]; stk_t[sp] = tm[a]; ++sp; }
      } else {
        ++leaf_v; ++leaf_steps; tri_t += (unsigned long long)cnt;
        // the kernel's peek: the entry on top of the stack as it stands
        bool paired = false; int32_t e2 = 0;
        if (sp > 0) {
          e2 = stk_ref[sp - 1];
          const int cnt2 = e2 & 15;
          if (cnt2 != 0) ++next_is_leaf;
          if (cnt2 != 0 && !(stk_t[sp - 1] > best.t) && cnt + cnt2 <= W) paired = true;
        }
        for (int j = 0; j < cnt; ++j) tri_leaf_test(tris + (size_t)(ref + j) * kTriFloats, oo, dd, best);
        if (paired) {
          --sp; ++pairs; ++leaf_v;
          const int32_t ref2 = e2 >> 4, cnt2 = e2 & 15;
          if (stk_t[sp] > best.t) ++wasted;                   // the first leaf's hit would have culled it: its tests are wasted work, not wrong
          tri_t += (unsigned long long)cnt2;
          for (int j = 0; j < cnt2; ++j) tri_leaf_test(tris + (size_t)(ref2 + j) * kTriFloats, oo, dd, best);
        }
      }
    }
  }
  (void)hist; (void)n_tris;
  out[0] = node_v; out[1] = leaf_v; out[2] = leaf_steps; out[3] = pairs; out[4] = tri_t; out[5] = wasted; out[6] = next_is_leaf; out[7] = 0;
}

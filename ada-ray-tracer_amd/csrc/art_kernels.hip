// art_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the render backend.
//
//   k_trace_coop   the hot kernel (about 88 % of the GPU time).  Persistent workgroups; a wave runs 64 / G independent rays, one per
//                  G-lane group (G = 4 by default: a DPP quad; G = 8 is the option bvh_width=8).  G = 4: a node of the 4-wide tree is one
//                  64-byte packet of four 16-byte lane records (art_qnode.h): lane j reads record j with ONE load, the quad reads 64
//                  contiguous bytes, the node's origin / scale words are broadcast inside the quad by DPP, all 64 lanes run one slab test,
//                  the hits are ranked by a DPP borrow chain and pushed far-to-near onto the ray's traversal stack in LDS.  A leaf holds
//                  <= G triangles: one Moeller-Trumbore test (reference arithmetic) per lane, then a DPP min-reduction over (t, key).
//                  Rays come as 64-byte trace records in queue order (prepared by k_analytic), pulled in chunks of `ray_chunk` records
//                  (48 by default) from one atomic cursor per XCD segment, the whole chunk prefetched (one load per lane per 16 records),
//                  handed to idle groups by ballot.
//   k_analytic     one ray per lane: spheres, Cornell box, rect lights, the reference's brute-force mesh; writes the starting bound of
//                  the BVH search and the trace record of every ray that needs it (incl. slab_setup's three divisions).
//   k_trace_simple one ray per lane, private stack -- the traversal used to cross-check; k_trace_overflow finishes the rays whose
//                  stack did not fit the capped LDS stack of k_trace_coop<.., OVF = true>.
//   k_raygen / k_shade_compact / k_resolve_last / k_fold / k_accumulate / k_resolve / k_debug   wavefront path-tracing stages over
//                  compacted work sets (per-item functions in art_shade.h); k_shade / k_finish: the plain in-place schedule.
//
// Replaces Scene.Find_Closest_Hit (scene.adb:56-86), the Embree bridge (embree_connect.cpp:196-238),
// PathTrace (ray_tracer-integrators.adb:82-301) and the DoPass pixel loop (integrators.adb:25-71).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <type_traits>
#include "art_kernels.h"

namespace art {

// ------------------------------------------------------------------------------------------------
// cross-lane helpers for 8-lane groups (DPP: quad_perm + row_half_mirror stay inside 8 lanes)
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }   // old = 0 + bound_ctrl: no tie to v, so no copy and the move can fold into its user (v_min_u32_dpp)

constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_XOR3 = 0x1B;          // quad_perm [3,2,1,0]
constexpr int DPP_HALF_MIRROR = 0x141;  // lane i <- lane 7-i  (= xor 7 inside an 8-lane group)

// value of lane (j ^ 4) of the same group: half mirror (xor 7) followed by quad reverse (xor 3)
__device__ __forceinline__ int xor4_i(int v) { return dpp_i<DPP_XOR3>(dpp_i<DPP_HALF_MIRROR>(v)); }

// number of lanes in my 8-lane group whose key is smaller than mine (keys are < 2^31).  (A hand-scheduled
// v_sub_co_u32_dpp / v_addc chain has fewer instructions but measured 4 % slower: its carry chain serialises.)
template <int G> __device__ __forceinline__ int group_rank_g(int key);
template <> __device__ __forceinline__ int group_rank_g<4>(int key) {       // 4-lane groups = DPP quads: three quad_perm compares
  int r = 0;
  r += (dpp_i<DPP_XOR1>(key) < key);
  r += (dpp_i<DPP_XOR2>(key) < key);
  r += (dpp_i<DPP_XOR3>(key) < key);
  return r;
}
__device__ __forceinline__ int group_rank(int key) {
  const int m = dpp_i<DPP_HALF_MIRROR>(key);
  int r = 0;
  r += (dpp_i<DPP_XOR1>(key) < key);
  r += (dpp_i<DPP_XOR2>(key) < key);
  r += (dpp_i<DPP_XOR3>(key) < key);
  r += (m < key);
  r += (dpp_i<DPP_XOR1>(m) < key);
  r += (dpp_i<DPP_XOR2>(m) < key);
  r += (dpp_i<DPP_XOR3>(m) < key);
  return r;
}

__device__ __forceinline__ uint64_t pack_tk(uint32_t tbits, uint32_t key) { return ((uint64_t)tbits << 32) | key; }

// 32-bit unsigned minimum across the 8-lane group (every lane gets it); min ops take the DPP operand directly
__device__ __forceinline__ uint32_t group_min_u32(uint32_t k) {
  k = min(k, (uint32_t)dpp_i<DPP_XOR1>((int)k));
  k = min(k, (uint32_t)dpp_i<DPP_XOR2>((int)k));
  k = min(k, (uint32_t)xor4_i((int)k));
  return k;
}

template <> __device__ __forceinline__ int group_rank_g<8>(int key) { return group_rank(key); }
template <int G> __device__ __forceinline__ uint32_t group_min_u32_g(uint32_t k);
template <> __device__ __forceinline__ uint32_t group_min_u32_g<8>(uint32_t k) { return group_min_u32(k); }
template <> __device__ __forceinline__ uint32_t group_min_u32_g<4>(uint32_t k) {
  k = min(k, (uint32_t)dpp_i<DPP_XOR1>((int)k));
  k = min(k, (uint32_t)dpp_i<DPP_XOR2>((int)k));
  return k;
}

// sum of h over the lanes of my group (every lane gets it): DPP adds inside the quad / the 8-lane group
template <int G> __device__ __forceinline__ int group_sum_g(int h);
template <> __device__ __forceinline__ int group_sum_g<4>(int h) {
  h += dpp_i<DPP_XOR1>(h);
  h += dpp_i<DPP_XOR2>(h);
  return h;
}
template <> __device__ __forceinline__ int group_sum_g<8>(int h) {
  h += dpp_i<DPP_XOR1>(h);
  h += dpp_i<DPP_XOR2>(h);
  h += xor4_i(h);
  return h;
}

// hip's __ballot() round-trips the predicate through a VGPR (v_cndmask + v_cmp); the builtin keeps it a lane mask
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// lane masks as explicit 64-bit scalars.  The step code below is branch-free and runs with all 64 lanes enabled, so a per-lane
// boolean IS a 64-bit mask in an SGPR pair.  Written with the compiler's i1 values, every `ballot` of a combined predicate costs a
// v_cndmask + v_cmp round trip through a VGPR; written as masks, compares produce them (v_cmp -> SGPR pair), the logic is s_and /
// s_andn2, a population count is s_bcnt1, and a select takes the mask as its SGPR operand.
// ------------------------------------------------------------------------------------------------
using mask_t = uint64_t;
__device__ __forceinline__ mask_t vcmp(bool direct_compare) { return __builtin_amdgcn_ballot_w64(direct_compare); }   // argument: ONE compare
__device__ __forceinline__ uint32_t sel(mask_t m, uint32_t if_set, uint32_t if_clear) {
  uint32_t r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
  return r;
}
__device__ __forceinline__ float self(mask_t m, float if_set, float if_clear) {
  return __builtin_bit_cast(float, sel(m, __builtin_bit_cast(uint32_t, if_set), __builtin_bit_cast(uint32_t, if_clear)));
}
__device__ __forceinline__ bool lane_of(mask_t m) { return sel(m, 1u, 0u) != 0u; }        // mask -> this lane's bit (off the hot path)

// the three DPP rank compares of a quad as one borrow chain starting from `start`: returns start - #{other lanes of my quad whose
// key is smaller than mine}.  Each compare is a DPP subtract (other lane's key - mine) whose borrow is "other < mine" (keys are
// below 2^31), taken up by v_subbrev: 6 instructions (mov_dpp + v_cmp + v_subbrev per compare: 9, measured 0.2-0.7 % slower).
__device__ __forceinline__ int quad_sub_rank(int key, int start) {
  int r, t;
  asm("s_nop 1\n\t"
      "v_sub_co_u32_dpp %1, vcc, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_subbrev_co_u32 %0, vcc, 0, %3, vcc\n\t"
      "v_sub_co_u32_dpp %1, vcc, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_subbrev_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_sub_co_u32_dpp %1, vcc, %2, %2 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_subbrev_co_u32 %0, vcc, 0, %0, vcc"
      : "=&v"(r), "=&v"(t) : "v"(key), "v"(start) : "vcc");
  return r;
}

// interval of a box given by its NEAR and FAR planes per axis (already chosen by the ray's direction signs), clipped to [0, tbest]:
// the same values slab_interval() gets from min / max of the two plane distances, in 4 instead of 10 min / max instructions
__device__ __forceinline__ void slab_near_far(f3 tn, f3 tf, float tbest, float& tmn, float& tmx) {
  float a, b;
  asm("v_max_f32 %0, 0, %1" : "=v"(a) : "v"(tn.z));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmn) : "v"(tn.x), "v"(tn.y), "v"(a));
  asm("v_min_f32 %0, %1, %2" : "=v"(b) : "v"(tf.z), "v"(tbest));
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmx) : "v"(tf.x), "v"(tf.y), "v"(b));
}

// LDS by 32-bit address (the stack pointer is kept as an LDS byte address: no base + offset add per access)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2 lds_u32x2;
__device__ __forceinline__ uint2 lds_load(uint32_t addr) { const u32x2 v = *reinterpret_cast<lds_u32x2*>(addr); return make_uint2(v.x, v.y); }
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ void lds_store(uint32_t addr, uint2 v) {       // two dwords from any two registers (ds_write2_b32): no pair to assemble
  asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(addr), "v"(v.x), "v"(v.y) : "memory");
}

// ---- "x = mask ? y : x" as ONE vector instruction under EXEC = mask (round 3).  The kernel's VALU is saturated (issue fraction 1.0,
// profiles/r3_final) while its scalar unit is 40 % busy: a v_cndmask holds the SIMD for 1.85 ns, the v_mov / v_add that does the same
// under a lane mask for 0.94 / 1.1 ns, and the two s_mov that set and restore EXEC issue beside the other waves' vector instructions.
// Only used where all 64 lanes are enabled (the step code of k_trace_coop).  ART_EXECM selects which selects are replaced (bit 0 node
// load, 1 pop, 2 push + sort key, 3 stack pointer = address of the top entry).  Measured on C4, trace launch average (two boxes, A/B in
// one call each): 0 -> 57.35 / 57.58 ms, 13 -> 56.73 / 56.85 (-1.2 %), 15 -> 57.15 / 57.18, 5 -> 57.06; alone, bit 1 (the pop) is 0.5 %
// slower, bits 0 / 2 / 3 are within noise: the step answers to the shape of its dependency chains more than to its instruction count.
#ifndef ART_EXECM
#define ART_EXECM 13
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Contract of the four helpers below: they are entered with ALL 64 lanes enabled and leave EXEC = -1.  They are `volatile`
// (never duplicated, hoisted or sunk across each other; EXEC cannot be named as a clobber -- the compiler rejects reserved registers
// there).  A caller that wrapped them in a per-lane branch would switch masked-off lanes on: k_trace_coop checks its side of the
// contract in builds with -DART_CHECK_EXEC (EXEC == -1 at the head of every node step, else the kernel traps): libart_hip_check.so,
// run by tests/test_gpu_widths.py::test_trace_kernel_keeps_its_exec_contract.
__device__ __forceinline__ void pop_masked(mask_t v1, mask_t ok1, uint32_t& sa, uint32_t& pend, uint32_t entry) {     // sa -= 8 in v1, pend = entry in ok1
  asm volatile("s_mov_b64 exec, %2\n\tv_add_u32 %0, -8, %0\n\ts_mov_b64 exec, %3\n\tv_mov_b32 %1, %4\n\ts_mov_b64 exec, -1" : "+v"(sa), "+v"(pend) : "s"(v1), "s"(ok1), "v"(entry));
}
__device__ __forceinline__ u32x4 load_node_masked(mask_t m, const char* base, uint32_t off) {                        // lanes outside m keep garbage: their results are masked
  u32x4 r;
  asm volatile("s_mov_b64 exec, %1\n\tglobal_load_dwordx4 %0, %2, %3\n\ts_mov_b64 exec, -1\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "s"(m), "v"(off), "s"(base) : "memory");
  return r;
}
__device__ __forceinline__ void lds_store_masked(mask_t m, uint32_t addr, uint2 v) {
  asm volatile("s_mov_b64 exec, %0\n\tds_write2_b32 %1, %2, %3 offset1:1\n\ts_mov_b64 exec, -1" :: "s"(m), "v"(addr), "v"(v.x), "v"(v.y) : "memory");
}
__device__ __forceinline__ uint32_t key_masked(mask_t hit, uint32_t tmn_bits, uint32_t j, uint32_t miss) {           // hit ? (tmn & ~7) | j : miss
  uint32_t k;
  asm volatile("v_mov_b32 %0, %4\n\ts_mov_b64 exec, %1\n\tv_and_or_b32 %0, %2, -8, %3\n\ts_mov_b64 exec, -1" : "=&v"(k) : "s"(hit), "v"(tmn_bits), "v"(j), "v"(miss));
  return k;
}

// scalar base + 32-bit byte offset: the address costs no vector instruction (a 64-bit pointer per array costs a v_lshl_add_u64 each)
// (ld_off / st_off: art_math.h)

// ------------------------------------------------------------------------------------------------
// cooperative persistent trace kernel
// ------------------------------------------------------------------------------------------------
#ifndef ART_COOP_WAVES_PER_SIMD
#define ART_COOP_WAVES_PER_SIMD 8   // <= 64 VGPRs: 8 waves per SIMD = 512 rays in flight per CU at 4 lanes per ray
#endif

// G = lanes per ray = children per node = triangles per leaf (4 or 8); 64 / G rays per wave.  OVF: the LDS stack holds fewer entries
// than the tree's worst-case bound, so a push is checked and a ray that would not fit is handed to k_trace_overflow.
//
// G = 4 walks the 64-byte quantised nodes (art_qnode.h).  A stack
// entry is { child entry word, bits(tmin) }: the entry word is the node's byte offset (inner child) or 0x80000000 | triangle byte
// offset | count (leaf), exactly as stored in the node, so a pop needs no decoding.  G = 8 walks the binary32 256-byte nodes; its
// entry word is (ref << 4) | count.
// INST (round 5, G = 4): the quantised node array holds a TWO-LEVEL tree (art_instanced_build.cpp: the instance tree first, then every mesh's
// tree in object space, entry words absolute).  A leaf entry with count 15 names an entry point of an instance (a whole instance, or a subtree of its mesh: art_scene.h DevInstance): the group takes its ray into the mesh's
// space (inv, noi and the plane selectors are replaced; o and d stay the world ray), pushes a "leave" marker (count 14) and goes on
// with the mesh's root; popped, the marker restores the world-space inv / noi / selectors from the ray's trace record.  t means the same in
// both spaces (the direction is not renormalised), so stack entries and the running bound carry over.  A mesh triangle is tested in WORLD
// space -- its corners through the instance's matrix with xform_point's arithmetic, the reference's Moeller-Trumbore on the world ray --
// so t, u, v are the flattened scene's, bit for bit; key index = instance << inst_shift | triangle of the mesh.
template <bool STATS, int G, bool OVF, bool INST = false>
__global__ __launch_bounds__(256, ART_COOP_WAVES_PER_SIMD) void k_trace_coop(const DevScene* __restrict__ Sp, const TraceArgs A) {
  static_assert(!INST || G == 4, "instanced scenes walk the 4-wide quantised nodes");
  extern __shared__ uint2 lds_stack[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NG = 64 / G;                                   // ray groups per wave
  const int j = lane & (G - 1), g = lane / G;
  const int gbase = lane & ~(G - 1);
  // Stack of one ray in LDS (bytes from sb):  0, 8: guard entries | 16 + 8 i: entry i | 16 + 8 entries: sink of masked pushes.
  // The stack pointer is the LDS address `sa = sb + 8 sp`: the top entry is at sa + 8 (an empty stack reads a guard, no clamping),
  // and a push goes to sa + 8 (nh - rank) + 8.
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds_stack;
  const uint32_t sb = lds0 + (uint32_t)(wave * NG + g) * (uint32_t)(A.stack_entries + 3) * 8u;
  const uint32_t sink = sb + 16u + (uint32_t)A.stack_entries * 8u;
  const uint32_t slimit = sb + (uint32_t)A.stack_entries * 8u;      // sa + 8 nh > slimit: the push does not fit
  const mask_t leaders = (G == 8) ? 0x0101010101010101ull : 0x1111111111111111ull;
  // kernel-argument bases stay in SGPRs; per-lane addressing is a 32-bit byte offset (scalar base + vector offset loads).
  // art_upload_scene guarantees that node and triangle byte offsets fit (31 bits for G = 4, 32 for G = 8).
  const char* const nodes_b = reinterpret_cast<const char*>(G == 4 ? (const void*)A.qnodes : (const void*)A.nodes);
  const char* const tris_b = reinterpret_cast<const char*>(G == 4 ? (const void*)A.qtris : (const void*)A.tris);
  const uint32_t jrec = 16u * (uint32_t)j;     // this lane's child record inside a node
  const uint32_t jtri = (uint32_t)j * (uint32_t)(G == 4 ? kQTriBytes : kTriBytes);
  const int n_queue = (A.queue_fixed >= 0) ? A.queue_fixed : (A.queue_items ? *A.queue_items * A.queue_mul : *A.queue_count);

  int chunk_pos = 0, chunk_end = 0;   // wave-uniform
  bool exhausted = false;             // wave-uniform
  // The live-ray queue is in ray order (neighbouring rays start at neighbouring surface points), so it is cut into one
  // contiguous segment per XCD: workgroups are dealt round-robin to the XCDs, each XCD has its own L2, and rays that walk
  // the same part of the tree then share an L2.  A wave whose segment is drained helps with the next one.
  const int n_seg = A.segments, seg_shift = __builtin_ctz((unsigned)n_seg);
  int seg = (int)blockIdx.x & (n_seg - 1), segs_left = n_seg;   // wave-uniform
  mask_t has_ray = 0, pend_valid = 0;                 // group-uniform bits: the group holds a ray / a popped entry waiting for its phase
  constexpr uint32_t kSaBias = (ART_EXECM & 8) ? 8u : 0u;      // the stack pointer register holds sb + 8 sp + kSaBias (bit 3: the address of the top entry itself)
  const uint32_t se = sb + kSaBias;                            // its value for an empty stack
  uint32_t sa = se; int ray = 0;
  int rec_i = 0;                                      // OVF / INST: the ray's record (k_trace_overflow; the world-space state to restore on leaving an instance)
  int cur_inst = -1;                                  // INST: the instance whose mesh the ray is in (-1: the instance tree, world space)
  f3 o = mk3(0, 0, 0), d = o, inv = o, noi = o;
  float best_t = 0.0f; uint32_t best_key = KEY_MISS;
  uint32_t pend = 0;                                  // the popped entry word
  uint32_t sel_near = 0, sel_far = 0;                 // G = 4: v_perm selectors that pick each axis' near / far plane byte by the ray's direction sign
  uint32_t held_key = KEY_MISS; float held_u = 0.0f, held_v = 0.0f;   // lane-local: barycentrics of the hit this lane found
  float shm = -1.0f; bool far_found = false;          // shadow-ray visibility rule (art_isect.h shadow_rule)
  uint64_t st_box = 0, st_tri = 0, st_node = 0, st_leaf = 0, st_it_node = 0, st_it_leaf = 0, st_it_all = 0;

  for (;;) {
    if (STATS) st_it_all += (lane == 0);
    // ---------------- refill idle groups from the wave's chunk of the live-ray queue
    float pf_keep = 0.0f;
    while (!exhausted) {
      const mask_t need_mask = ~has_ray & leaders;
      if (need_mask == 0) break;
      // a refill costs the whole wave its instructions (four record loads, ~30 VALU) whatever the number of rays it brings in: wait until
      // refill_min groups are idle, unless the wave has nothing else to do
      if (__popcll(need_mask) < A.refill_min && has_ray != 0) break;
      if (chunk_pos == chunk_end) {
        const int seg_lo = (int)(((int64_t)n_queue * seg) >> seg_shift), seg_hi = (int)(((int64_t)n_queue * (seg + 1)) >> seg_shift);   // n_seg is a power of two
        int base = 0;
        if (lane == 0) base = atomicAdd(A.cursor + 32 * (seg + 1), A.chunk);
        base = __builtin_amdgcn_readfirstlane(base) + seg_lo;
        chunk_pos = base; chunk_end = min(base + A.chunk, seg_hi);
        if (chunk_pos >= seg_hi) {
          chunk_pos = chunk_end = 0;
          seg = (seg + 1) & (n_seg - 1);
          if (--segs_left == 0) { exhausted = true; break; }
          continue;
        }
        // the chunk's records (16 x 64 B = one load per lane, per 16 records) on their way into the cache before the first group asks for its own
        // (the value is only "used" after the refill's own loads below, so that nothing waits for it alone)
        // (round 5: fetching the records with non-temporal loads -- they are read once -- made the launch 2 % (C4) to 9 % (C5) SLOWER,
        // profiles/r5_shade/ab11_trace_nt_loads.txt: the prefetch below relies on the lines staying in the cache until the refill reads them)
        for (int c16 = 0; c16 < A.chunk; c16 += 16) pf_keep += reinterpret_cast<const float*>(A.rec)[((size_t)(chunk_pos + c16) * 4 + lane) * 4];
      }
      const int avail = chunk_end - chunk_pos;
      const int n_need = __popcll(need_mask);
      const int my_rank = __popcll(need_mask & ((1ull << gbase) - 1ull));
      const bool got = !lane_of(has_ray) && (my_rank < avail);
      if (got) {
        // the ray's 64-byte trace record, prepared by k_analytic (analytic primitives already intersected: the starting bound; slab_setup's
        // three correctly rounded divisions; the shadow rule's starting state).  Records are in queue order: the address depends on
        // nothing but the wave's cursor, and the whole chunk was fetched by the prefetch above.
        const char* rb = reinterpret_cast<const char*>(A.rec) + (size_t)chunk_pos * (size_t)kTraceRecBytes;      // wave-uniform
        const uint32_t ro = (uint32_t)my_rank * (uint32_t)kTraceRecBytes;
        const float4 r0 = ld_off(reinterpret_cast<const float4*>(rb), ro), r1 = ld_off(reinterpret_cast<const float4*>(rb), ro + 16u),
                     r2 = ld_off(reinterpret_cast<const float4*>(rb), ro + 32u), r3 = ld_off(reinterpret_cast<const float4*>(rb), ro + 48u);
        o = mk3(r0.x, r0.y, r0.z); d = mk3(r1.x, r1.y, r1.z); inv = mk3(r2.x, r2.y, r2.z);
        noi = mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
        best_t = r0.w; best_key = __builtin_bit_cast(uint32_t, r1.w); shm = r2.w;
        sel_near = __builtin_bit_cast(uint32_t, r3.x); sel_far = 0x18070503u - sel_near;      // per byte: near + far = 3, 5, 7, 0x18 (0x0c + 0x0c)
        ray = __builtin_bit_cast(int, r3.y); far_found = __builtin_bit_cast(uint32_t, r3.z) != 0u;
        if (OVF) rec_i = chunk_pos + my_rank;
        if (INST) cur_inst = -1;
        held_key = KEY_MISS;
        sa = se; pend = 0u;                                       // entry word 0 = root node (both encodings)
      }
      asm volatile("" :: "v"(pf_keep));
      const mask_t got_mask = __builtin_amdgcn_ballot_w64(got);
      has_ray |= got_mask; pend_valid |= got_mask;
      chunk_pos += min(avail, n_need);
    }
    if (has_ray == 0) break;

    // The step code below is branch-free on purpose: on CDNA a divergent `if` costs three scalar instructions (save/restore exec +
    // skip branch).  Inactive lanes compute on clamped operands and are masked by selects; LDS pushes of non-hit lanes go to the
    // group's sink slot.  Bookkeeping (retire / refill / leaf vote) is kept out of the inner node loop.

    // ---------------- inner loop: pop + node phase, as long as enough of the wave's groups have a node to expand
    mask_t want_leaf = 0;
    // The refill above may have been refused (fewer than refill_min idle groups).  Then "not exhausted" is no reason to leave the inner
    // loop with node work pending: nothing outside it could make progress either (node_min = 8 or refill_min >= 6 at width 8
    // spun forever).  Only a refill that would really happen ends the node phase early.
    const bool refill_possible = !exhausted && (__popcll(~has_ray & leaders) >= A.refill_min);
    for (;;) {
      // next stack entry; an entry culled by the current hit is dropped (its group then sits this step out).  Popping two entries
      // per iteration to skip a culled one costs 12 more instructions in every iteration and saves 0.6 % of the steps: slower.
      {
        const uint2 e1 = lds_load(sa + (8u - kSaBias));
        const mask_t v1 = has_ray & ~pend_valid & vcmp(sa > se);
        const mask_t ok1 = v1 & vcmp(!(__builtin_bit_cast(float, e1.y) > best_t));
        if (ART_EXECM & 2) pop_masked(v1, ok1, sa, pend, e1.x);
        else { pend = sel(ok1, e1.x, pend); sa = sel(v1, sa - 8u, sa); }
        pend_valid |= ok1;
      }
      if (INST) {
        // A popped "leave" marker is dealt with here, not in the leaf phase: two LDS reads and eight instructions for the groups that hold
        // one.  Waiting for the next leaf phase cost a pass of the outer loop per instance left: 2.2 x the leaf-phase passes of the
        // flattened scene at equal node visits; consumed here 1.7 x and +11 % Mrays/s.  (The ENTRY handled here too brought the passes down to
        // the flattened scene's and was 3 % slower: ~100 instructions and four loads inside the node loop for one or two groups;
        // it stays in the leaf phase, which serves every waiting group at once.  profiles/r5_instanced.json)
        const mask_t leaving = has_ray & pend_valid & vcmp(pend == kQEntryLeaveInstance);
        if (leaving != 0) {
          if (lane_of(leaving)) {
            const uint2 s1 = lds_load(sa + (8u - kSaBias)), s0 = lds_load(sa - kSaBias);      // under the marker: world-space inv.xy | inv.z, near-plane selector
            sa -= 16u;
            inv = mk3(__builtin_bit_cast(float, s0.x), __builtin_bit_cast(float, s0.y), __builtin_bit_cast(float, s1.x));
            noi = mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z));
            sel_near = s1.y; sel_far = 0x18070503u - sel_near;
            cur_inst = -1;
          }
          pend_valid &= ~leaving;                          // those groups pop their next entry in the next iteration
        }
      }
      const mask_t active = has_ray & pend_valid;
      const mask_t is_leaf = (G == 4) ? vcmp((int)pend < 0) : vcmp((pend & 15u) != 0u);
      want_leaf = active & is_leaf;
      const mask_t want_node = active & ~is_leaf;
      // leave when fewer than node_min groups still expand nodes (the others wait on a leaf, are finished, or idle)
      if (__popcll(want_node) < 8 * A.node_min) {
        if (want_node == 0 || refill_possible || (want_leaf | (has_ray & ~active & vcmp(sa == se))) != 0) break;
      }
      // ---- node phase: lane j tests child j; groups not taking part read the root node and discard the result
#if defined(ART_CHECK_EXEC)
      if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();      // the exec-masked helpers restore EXEC to -1: they must be entered with -1
#endif
      float tmn, tmx; uint32_t entry; mask_t valid;
#if defined(ART_DIAG_LOAD)
      uint32_t diag_x[ART_DIAG_LOAD] = {};
#endif
      if (G == 4) {
        const uint32_t noff = (ART_EXECM & 1) ? pend : sel(want_node, pend, 0u);
        uint4 rec;
        if (ART_EXECM & 1) { const u32x4 rr = load_node_masked(want_node, nodes_b, noff + jrec); rec = make_uint4(rr.x, rr.y, rr.z, rr.w); }
        else rec = *reinterpret_cast<const uint4*>(nodes_b + (noff + jrec));              // ONE load: the quad reads the node's 64 contiguous bytes
        const uint32_t c0 = rec.x, c1 = rec.y; entry = rec.z;
        asm volatile("s_setprio 1" :: "v"(c0));           // a wave whose node has arrived goes ahead of the waves that pop, refill or test leaves (+0.7 % on C4, +3 % on C5)
        const float4 h = make_float4(__builtin_bit_cast(float, dpp_i<0x00>((int)rec.w)), __builtin_bit_cast(float, dpp_i<0x55>((int)rec.w)),
                                     __builtin_bit_cast(float, dpp_i<0xAA>((int)rec.w)), __builtin_bit_cast(float, dpp_i<0xFF>((int)rec.w)));
        // near / far plane bytes by direction sign (one v_perm each), dequantised (one fma per plane: the very binary32 boxes of the
        // exported tree), then the slab distances.  An empty slot holds lo = 255, hi = 0 on every axis: its near plane lies behind
        // its far plane for every ray, so it needs no validity test (and its entry word is a leaf of zero triangles: harmless
        // even if a degenerate node's planes round onto each other).
        const uint32_t nb = __builtin_amdgcn_perm(c1, c0, sel_near), fb = __builtin_amdgcn_perm(c1, c0, sel_far);
        const f3 pn = mk3(__builtin_fmaf((float)(nb & 255u), h.w, h.x), __builtin_fmaf((float)((nb >> 8) & 255u), h.w, h.y), __builtin_fmaf((float)((nb >> 16) & 255u), h.w, h.z));
        const f3 pf = mk3(__builtin_fmaf((float)(fb & 255u), h.w, h.x), __builtin_fmaf((float)((fb >> 8) & 255u), h.w, h.y), __builtin_fmaf((float)((fb >> 16) & 255u), h.w, h.z));
        slab_near_far(mk3(__builtin_fmaf(pn.x, inv.x, noi.x), __builtin_fmaf(pn.y, inv.y, noi.y), __builtin_fmaf(pn.z, inv.z, noi.z)),
                      mk3(__builtin_fmaf(pf.x, inv.x, noi.x), __builtin_fmaf(pf.y, inv.y, noi.y), __builtin_fmaf(pf.z, inv.z, noi.z)), best_t, tmn, tmx);
        // (no validity mask on the walk itself, timed or counted; `valid` only feeds the box-test COUNT of the counting variant below)
        valid = STATS ? vcmp(entry != kQEntryEmpty) : ~0ull;
#if defined(ART_DIAG_VALU)          // diagnostic builds only (profiles/diag.sh): n more dependent VALU instructions per node step
        { float x = tmn;
#pragma unroll
          for (int k = 0; k < ART_DIAG_VALU; ++k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(tmx)); }
#endif
#if defined(ART_DIAG_LOAD)          // ... or n more vector loads per lane and node step from the node's own line (no new L2 traffic: the TA / L1 path alone)
        {
#pragma unroll
          for (int k = 0; k < ART_DIAG_LOAD; ++k) { uint32_t off2 = noff + jrec + 4u * (uint32_t)k; asm volatile("" : "+v"(off2)); diag_x[k] = *reinterpret_cast<const uint32_t*>(nodes_b + off2); } }
#endif
      } else {
        const uint32_t noff = sel(want_node, pend >> 4, 0u) * (uint32_t)(G * 32) + jrec;
        const float4 r0 = *reinterpret_cast<const float4*>(nodes_b + noff);
        const float4 r1 = *reinterpret_cast<const float4*>(nodes_b + noff + (uint32_t)(G * 16));
        const int cref = __builtin_bit_cast(int, r0.w);
        entry = (uint32_t)((cref << 4) | __builtin_bit_cast(int, r1.w));
        slab_interval(mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), inv, noi, best_t, tmn, tmx);
        valid = vcmp(cref >= 0);
      }
      // the counting variant (STATS) walks with exactly the timed kernel's predicate: at G = 4 an empty slot is never masked out, it
      // simply cannot pass the interval test (and if a degenerate node ever let it, both variants would pop the same zero-triangle leaf)
      const mask_t hit = (G == 4) ? (want_node & vcmp(tmn <= tmx)) : (want_node & valid & vcmp(tmn <= tmx));
      const int key = (ART_EXECM & 4) ? (int)key_masked(hit, __builtin_bit_cast(uint32_t, tmn), (uint32_t)j, 0x7fffffffu)
                                      : (int)sel(hit, (__builtin_bit_cast(uint32_t, tmn) & ~7u) | (uint32_t)j, 0x7fffffffu);
      const int nh = group_sum_g<G>((int)sel(hit, 1u, 0u));
      const int nh_minus_rank = (G == 4) ? quad_sub_rank(key, nh) : nh - group_rank_g<G>(key);
      const uint32_t top = sa + (uint32_t)nh * 8u;
      const mask_t ovf = OVF ? (want_node & vcmp(top > slimit + kSaBias)) : 0;        // the ray moves to k_trace_overflow
      const uint32_t dst_hit = (sa + (8u - kSaBias)) + (uint32_t)nh_minus_rank * 8u;
      if (ART_EXECM & 4) lds_store_masked(hit & ~ovf, dst_hit, make_uint2(entry, __builtin_bit_cast(uint32_t, tmn)));
      else lds_store(sel(hit & ~ovf, dst_hit, sink), make_uint2(entry, __builtin_bit_cast(uint32_t, tmn)));
      sa = OVF ? sel(ovf, se, top) : top;
      if (OVF) {
        if (ovf != 0) {
          if (lane_of(ovf) && j == 0) A.ovf_queue[atomicAdd(A.ovf_count, 1)] = rec_i;
          has_ray &= ~ovf;
        }
      }
      pend_valid &= ~want_node;
#if defined(ART_DIAG_LOAD)
      if (G == 4) {
#pragma unroll
        for (int k = 0; k < ART_DIAG_LOAD; ++k) asm volatile("" :: "v"(diag_x[k]));
      }
#endif
      if (STATS) { st_box += lane_of(want_node & valid); st_node += (lane_of(want_node) && j == 0); st_it_node += (lane == 0); }
      __builtin_amdgcn_s_setprio(0);
      wave_lds_sync();
    }

    // ---------------- retire rays whose stack ran dry
    const mask_t done = has_ray & ~pend_valid & vcmp(sa == se);
    if (done != 0) {
      if (lane_of(done)) {
        // Only a hit found by THIS kernel needs storing, by the lane that holds it and as one 16-byte record: the starting bound (no hit,
        // or an analytic / brute-force hit) was stored by k_analytic, a shadow ray's far hit at the moment it was found (below).
        if (best_key != KEY_MISS && held_key == best_key) {
          if (ray < 0) st_off(A.sh_t, ((uint32_t)ray & ~kShadowWord) << 2, best_t);           // a shadow ray of the record schedule: its t is all the stage asks for
          else st_off(A.hit, (uint32_t)ray << 4, DevHit{best_t, best_key, held_u, held_v});
        }
      }
      has_ray &= ~done;
    }

    // ---------------- leaf phase for every group holding a leaf: lane j < cnt tests triangle j; others test triangle 0, masked
    if (want_leaf != 0) {
      const bool wl = lane_of(want_leaf);
      const int cnt = (int)(pend & 15u);
      const bool special = INST && wl && (cnt >= (int)kQCountLeaveInstance);        // an instance to enter (15) or its "leave" marker (14): no triangles
      const bool tri_lane = wl && !special && (j < cnt);
      const uint32_t tbase = (G == 4) ? (pend & 0x7ffffff0u) : (pend >> 4) * (uint32_t)kTriBytes;
      // only the lanes that hold a triangle load (about 12 of 64): the other lanes' result is masked anyway
      float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0, q2 = q0;
      if (tri_lane) {
        const uint32_t toff = tbase + jtri;
        q0 = *reinterpret_cast<const float4*>(tris_b + toff);
        q1 = *reinterpret_cast<const float4*>(tris_b + toff + 16u);
        q2 = *reinterpret_cast<const float4*>(tris_b + toff + 32u);
      }
      float tt, uu, vv;
      f3 tA = mk3(q0.x, q0.y, q0.z), tB = mk3(q0.w, q1.x, q1.y), tC = mk3(q1.z, q1.w, q2.x);
      uint32_t key_hi = KEY_TRI;
      if (INST) {                                          // the mesh's triangle in world space: the flattening's own arithmetic (xform_point)
        if (tri_lane) {
          const char* const ib = reinterpret_cast<const char*>(A.inst) + (size_t)(uint32_t)cur_inst * sizeof(DevInstance);
          const float4 m0 = *reinterpret_cast<const float4*>(ib), m1 = *reinterpret_cast<const float4*>(ib + 16), m2 = *reinterpret_cast<const float4*>(ib + 32);
          const float mm[12] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w};
          tA = xform_point(mm, tA); tB = xform_point(mm, tB); tC = xform_point(mm, tC);
          key_hi = KEY_TRI | ((uint32_t)cur_inst << A.inst_shift);
        }
      }
      const bool pass = tri_raw(o, d, tA, tB, tC, tt, uu, vv);
      const bool valid = tri_lane && pass && (tt > 0.0f) && (tt < 1000000.0f);
      const uint32_t tb = valid ? __builtin_bit_cast(uint32_t, tt) : 0x7f7fffffu;
      const uint32_t key = valid ? (key_hi | (uint32_t)__builtin_bit_cast(int, q2.y)) : KEY_MISS;
      // lexicographic (t, key) minimum as two 32-bit reductions: smallest t, then smallest key among the lanes holding it
      const uint32_t win_tb = group_min_u32_g<G>(tb);
      const uint32_t win_key = group_min_u32_g<G>(tb == win_tb ? key : 0xffffffffu);
      const uint64_t win = pack_tk(win_tb, win_key);
      // cand_wins for t > 0:  (t, key) < (best_t, best_key), where an equal t never displaces the initial bound
      const uint32_t bt = (best_t == 0.0f) ? 0u : __builtin_bit_cast(uint32_t, best_t);
      const uint64_t cur = pack_tk(bt, best_key == KEY_MISS ? 0u : best_key);
      const bool accept = wl && ((uint32_t)win != KEY_MISS) && (win < cur);
      const float win_t = __builtin_bit_cast(float, (uint32_t)(win >> 32));
      best_t = accept ? win_t : best_t;
      best_key = accept ? (uint32_t)win : best_key;
      const bool sh_hit = accept && (shm >= 0.0f);
      if (ballot64(sh_hit) != 0) {                         // shadow_rule, group-uniform
        const bool near = sh_hit && (win_t <= shm);
        const bool far = sh_hit && !near;                  // first far hit (afterwards the bound is <= shm)
        if (far && j == 0) {
          if (ray < 0) st_off(A.sh_t, ((uint32_t)ray & ~kShadowWord) << 2, win_t);
          else st_off(A.hit, (uint32_t)ray << 4, DevHit{win_t, (uint32_t)win, 0.0f, 0.0f});
        }
        far_found = far_found || far;
        best_t = far ? next_up_pos(shm) : best_t;
        best_key = far ? KEY_MISS : best_key;
        sa = near ? se : sa;                               // near hit: nothing left to learn
      }
      const bool mine = accept && valid && (key == (uint32_t)win);
      held_key = mine ? key : held_key; held_u = mine ? uu : held_u; held_v = mine ? vv : held_v;
      mask_t entered = 0;                                  // groups that go on with a mesh's root: their popped entry stays valid
      if (INST) {
        const mask_t sp_mask = ballot64(special);
        if (sp_mask != 0) {                                // rare next to node steps: a few instances per ray
          bool enter = special && (cnt == (int)kQCountInstance);
          if (enter) {
            const uint32_t ii = (pend >> 4) & 0x07ffffffu;       // an ENTRY POINT of an instance (art_scene.h DevInstance)
            const char* const ib = reinterpret_cast<const char*>(A.inst) + (size_t)ii * sizeof(DevInstance);
            const float4 w0 = *reinterpret_cast<const float4*>(ib + 48), w1 = *reinterpret_cast<const float4*>(ib + 64), w2 = *reinterpret_cast<const float4*>(ib + 80);
            const uint2 wi = *reinterpret_cast<const uint2*>(ib + 112);      // DevInstance::qroot, inst
            const float mi[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
            const f3 oo = xform_point(mi, o), dd = mk3(mi[0] * d.x + mi[1] * d.y + mi[2] * d.z, mi[4] * d.x + mi[5] * d.y + mi[6] * d.z, mi[8] * d.x + mi[9] * d.y + mi[10] * d.z);
            const uint32_t top = sa + 24u;                 // three entries: the world-space inv.xy | inv.z, near-plane selector | the "leave" marker on top
            if (OVF && top > slimit + kSaBias) {           // they do not fit the capped stack: the ray moves to k_trace_overflow
              if (j == 0) A.ovf_queue[atomicAdd(A.ovf_count, 1)] = rec_i;
              enter = false;
            } else {
              // what leaving the instance has to restore goes under the marker, on the ray's own LDS stack (re-reading it from the trace
              // record would put a global-memory round trip into every instance left)
              lds_store(sa + (16u - kSaBias), make_uint2(__builtin_bit_cast(uint32_t, inv.x), __builtin_bit_cast(uint32_t, inv.y)));
              lds_store(sa + (24u - kSaBias), make_uint2(__builtin_bit_cast(uint32_t, inv.z), sel_near));
              lds_store(sa + (32u - kSaBias), make_uint2(kQEntryLeaveInstance, 0u));      // under the mesh's tree: popped when that tree is done
              slab_setup(oo, dd, inv, noi);
              const uint32_t sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
              sel_near = (sx ? 3u : 0u) | ((sy ? 4u : 1u) << 8) | ((sz ? 5u : 2u) << 16) | 0x0c000000u; sel_far = 0x18070503u - sel_near;
              sa = top;
              cur_inst = (int)wi.y;                        // the entry point's instance: what a hit's key and the triangles' matrix go by
              pend = wi.x;                                 // DevInstance::qroot: where this entry point enters the mesh's tree (an inner entry, or a leaf: the next pop's business either way)
            }
          }                                                // (a "leave" marker never gets here: the inner loop consumes it as it pops it)
          entered = ballot64(enter);
          if (OVF) { const mask_t gone = ballot64(special && (cnt == (int)kQCountInstance) && !enter); if (gone != 0) { has_ray &= ~gone; sa = lane_of(gone) ? se : sa; } }
        }
      }
      pend_valid &= ~(want_leaf & ~entered);
      if (STATS) { st_tri += tri_lane; st_leaf += (wl && !special && j == 0); st_it_leaf += (lane == 0); }
    }
  }
  if (STATS) {
    atomicAdd(&A.stats[0], (unsigned long long)st_box); atomicAdd(&A.stats[1], (unsigned long long)st_tri);
    atomicAdd(&A.stats[2], (unsigned long long)st_node); atomicAdd(&A.stats[3], (unsigned long long)st_leaf);
    if (lane == 0) { atomicAdd(&A.stats[5], (unsigned long long)st_it_node); atomicAdd(&A.stats[6], (unsigned long long)st_it_leaf); atomicAdd(&A.stats[7], (unsigned long long)st_it_all); }
  }
}

// ------------------------------------------------------------------------------------------------
// overflow path of k_trace_coop
// ------------------------------------------------------------------------------------------------
// rays k_trace_coop<.., OVF = true> gave up on (capped LDS stack): one ray per lane, the BVH search again from the ray's trace record
// (the starting bound and the shadow rule's starting state are in it) with the private full-size stack.  What ends up in the hit record
// is what k_trace_coop would have left there: a hit found in the BVH, else the first far hit of a shadow ray, else the starting bound
// its producer stored.
template <bool STATS>
__global__ __launch_bounds__(256) void k_trace_overflow(const DevScene* __restrict__ Sp, const TraceArgs A) {
  const DevScene& S = *Sp;
  const int n = *A.ovf_count;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const float4* r = A.rec + 4 * (size_t)A.ovf_queue[k];
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    const int i = __builtin_bit_cast(int, r3.y);
    const bool far0 = __builtin_bit_cast(uint32_t, r3.z) != 0u;
    BvhStats st = {0, 0, 0, 0};
    Cand best; best.t = r0.w; best.key = __builtin_bit_cast(uint32_t, r1.w); best.u = 0.0f; best.v = 0.0f;
    ShadowState sh; sh.shm = r2.w; sh.far = far0; sh.rep = best;
    if (S.n_inst > 0) instanced_render_closest<STATS>(S, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), best, &st, sh);      // (instanced scene: the two-level search)
    else bvh_closest<STATS>(S, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), best, &st, sh);
    const bool word = i < 0;                                   // kShadowWord: the result is one float (DevPaths::sh_t)
    const uint32_t wi = (uint32_t)i & ~kShadowWord;
    if (best.key != KEY_MISS && (best.key & ~KEY_INDEX_MASK) == KEY_TRI) { if (word) A.sh_t[wi] = best.t; else A.hit[i] = DevHit{best.t, best.key, best.u, best.v}; }
    else if (sh.far && !far0) { if (word) A.sh_t[wi] = sh.rep.t; else A.hit[i] = DevHit{sh.rep.t, sh.rep.key, 0.0f, 0.0f}; }
  }
}

// ------------------------------------------------------------------------------------------------
// k_trace_inst: the trace kernel of INSTANCED scenes (round 5; DevScene::n_inst > 0): one ray per lane over the trace records of the
// bank, the two-level search of art_instanced.h (a tree over the instances, one tree per mesh walked with the ray in object space, the
// triangles tested in world space: the flattened scene's t, u, v).  Same protocol as k_trace_coop / k_trace_overflow: the record carries
// the starting bound and the shadow rule's state; what ends up in the hit slot is a hit found in the meshes, else the first far hit of a
// shadow ray, else what the producer stored.  It is the cross-check and the option inst_coop = 0: since round 5 an instanced scene
// renders through k_trace_coop<.., INST = true>, which crosses the instance boundary on its LDS stack (3.7 against
// 0.41 Grays/s on I64).
// ------------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(256) void k_trace_inst(const DevScene* __restrict__ Sp, const TraceArgs A) {
  const DevScene& S = *Sp;
  const int n_queue = (A.queue_fixed >= 0) ? A.queue_fixed : (A.queue_items ? *A.queue_items * A.queue_mul : *A.queue_count);
  BvhStats st = {0, 0, 0, 0};
  unsigned long long traced = 0;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_queue; k += gridDim.x * blockDim.x) {
    const float4* r = A.rec + 4 * (size_t)k;
    const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    if (!(r0.w >= 0.0f)) continue;                            // a ray that does not exist, or one the analytic pass decided
    const int i = __builtin_bit_cast(int, r3.y);
    const bool far0 = __builtin_bit_cast(uint32_t, r3.z) != 0u;
    Cand best; best.t = r0.w; best.key = __builtin_bit_cast(uint32_t, r1.w); best.u = 0.0f; best.v = 0.0f;
    ShadowState sh; sh.shm = r2.w; sh.far = far0; sh.rep = best;
    instanced_render_closest<STATS>(S, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), best, &st, sh);
    if (STATS) traced += 1;
    const bool word = i < 0;                                   // kShadowWord: the result is one float (DevPaths::sh_t)
    const uint32_t wi = (uint32_t)i & ~kShadowWord;
    if (best.key != KEY_MISS && (best.key & ~KEY_INDEX_MASK) == KEY_TRI) { if (word) A.sh_t[wi] = best.t; else A.hit[i] = DevHit{best.t, best.key, best.u, best.v}; }
    else if (sh.far && !far0) { if (word) A.sh_t[wi] = sh.rep.t; else A.hit[i] = DevHit{sh.rep.t, sh.rep.key, 0.0f, 0.0f}; }
  }
  if (STATS) {
    atomicAdd(&A.stats[0], (unsigned long long)st.box_tests); atomicAdd(&A.stats[1], (unsigned long long)st.tri_tests);
    atomicAdd(&A.stats[2], (unsigned long long)st.node_visits); atomicAdd(&A.stats[3], (unsigned long long)st.leaf_visits);
    atomicAdd(&A.stats[4], traced);
  }
}

// ------------------------------------------------------------------------------------------------
// k_analytic: one ray per lane.  Intersects the analytic primitives and the reference brute-force mesh
// (scene.adb:62-69 candidates 1-4), stores the result as the starting bound of the BVH search and appends
// the rays that still need the BVH to the live-ray queue (dead rays and decided shadow rays drop out here).
// ------------------------------------------------------------------------------------------------
constexpr int kAnalyticLdsSpheres = 64, kAnalyticLdsLights = 16;
constexpr int kAnalyticChunk = 4096;   // rays per workgroup: ONE global atomic per chunk for the queue (a single hot word
                                       // serves only ~88 atomics/us on this chip, so per-wave atomics were the bottleneck)

__global__ __launch_bounds__(256) void k_analytic(const DevScene S, const TraceArgs A) {   // the scene by value: kernel arguments are invariant, so its fields are fetched once, not once per round
  __shared__ uint16_t s_idx[kAnalyticChunk];            // queued rays of this chunk, as offsets from chunk0
  __shared__ float4 s_rec[4][256 + 1];                  // [piece][thread] staging of the trace records (+1: the read-back of 4 consecutive lanes hits 4 different banks)
  __shared__ int s_count, s_live, s_base;
  __shared__ DevLight s_lgt[kAnalyticLdsLights];       // ... and its lights (the rect ones are intersected here)
  __shared__ DevSphere s_sph[kAnalyticLdsSpheres];     // the scene's spheres, fetched once per workgroup: every ray tests every sphere, and a global
                                                       // load per sphere and ray (the compiler cannot keep them across the stores) is a wait per sphere
  if (threadIdx.x == 0) { s_count = 0; s_live = 0; }
  const int n_sph_lds = S.n_spheres < kAnalyticLdsSpheres ? S.n_spheres : kAnalyticLdsSpheres;
  const int n_lgt_lds = S.n_lights < kAnalyticLdsLights ? S.n_lights : kAnalyticLdsLights;
  if ((int)threadIdx.x < n_sph_lds) s_sph[threadIdx.x] = S.spheres[threadIdx.x];
  if ((int)threadIdx.x < n_lgt_lds) s_lgt[threadIdx.x] = S.lights[threadIdx.x];
  __syncthreads();
  // compacted work set: only items [0, *item_count) exist; their extension rays are [0, n), their shadow rays [shadow_begin, shadow_begin + n)
  const int n_items = A.item_count ? *A.item_count : 0x7fffffff;
  const int chunk0 = blockIdx.x * kAnalyticChunk;
  if (A.item_count && ((chunk0 >= n_items && chunk0 + kAnalyticChunk <= A.shadow_begin) || chunk0 >= A.shadow_begin + n_items)) return;   // nothing exists in this chunk
  // Software pipeline over the rounds: the seven loads of round r + 1 are in flight while round r is intersected (one ray per lane has
  // no other way to overlap its memory round trip with its ~250 instructions of arithmetic).
  auto exists = [&](int i) { return (i < A.n_rays) && (!A.item_count || ((i < A.shadow_begin) ? (i < n_items) : (i - A.shadow_begin < n_items))); };
  struct RayIn { float tfar, ox, oy, oz, dx, dy, dz, shm; };
  auto fetch = [&](int i, bool ok) {
    RayIn r = {-1.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, -1.0f};
    if (ok) {
      r.tfar = A.ray_tfar[i]; r.ox = A.ray_ox[i]; r.oy = A.ray_oy[i]; r.oz = A.ray_oz[i]; r.dx = A.ray_dx[i]; r.dy = A.ray_dy[i]; r.dz = A.ray_dz[i];
      r.shm = (A.sh_min != nullptr && i >= A.shadow_begin) ? A.sh_min[i - A.shadow_begin] : -1.0f;
    }
    return r;
  };
  RayIn nxt = fetch(chunk0 + (int)threadIdx.x, exists(chunk0 + (int)threadIdx.x));
  for (int k0 = 0; k0 < kAnalyticChunk; k0 += 256) {
    const int i = chunk0 + k0 + threadIdx.x;
    bool queue_it = false, live = false;
    const RayIn cur = nxt;
    if (k0 + 256 < kAnalyticChunk) nxt = fetch(i + 256, exists(i + 256));
    {
      const float tfar = cur.tfar;                       // a ray that does not exist reads as dead
      const f3 o = mk3(cur.ox, cur.oy, cur.oz), d = mk3(cur.dx, cur.dy, cur.dz);
      const float shm = cur.shm;
      if (tfar >= 0.0f) {
        live = true;
        Cand best = cand_init(tfar);
        for (int k = 0; k < n_sph_lds; ++k) isect_sphere(o, d, s_sph[k], (uint32_t)k, best);
        for (int k = n_sph_lds; k < S.n_spheres; ++k) isect_sphere(o, d, S.spheres[k], (uint32_t)k, best);
        const f3 rcp = ray_rcp(d);
        if (S.has_cornell) isect_cornell(o, d, rcp, S, best);
        for (int k = 0; k < n_lgt_lds; ++k)
          if (s_lgt[k].shape == LIGHT_RECT) isect_quad(o, d, &s_lgt[k], (uint32_t)k, best);
        for (int k = n_lgt_lds; k < S.n_lights; ++k)
          if (S.lights[k].shape == LIGHT_RECT) isect_quad(o, d, S.lights + k, (uint32_t)k, best);
        isect_bf_mesh(o, d, rcp, S, best);
        A.hit[i] = DevHit{best.t, best.key, best.u, best.v};
        const bool near_done = (shm >= 0.0f) && (best.key != KEY_MISS) && (best.t <= shm);   // shadow_rule: decided
        queue_it = (A.n_tris > 0) && !near_done;
      }
    }
    // wave-aggregated append into the workgroup's LDS list
    const uint64_t m = __builtin_amdgcn_ballot_w64(queue_it);
    const uint64_t lv = __builtin_amdgcn_ballot_w64(live);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) {
      if (m) base = atomicAdd(&s_count, (int)__popcll(m));
      if (lv) atomicAdd(&s_live, (int)__popcll(lv));
    }
    base = __builtin_amdgcn_readfirstlane(base);
    if (queue_it) s_idx[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(i - chunk0);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_base = s_count ? atomicAdd(A.queue_count, s_count) : 0;
    if (s_live) atomicAdd(A.live_rays, (unsigned long long)s_live);
    if (A.stats != nullptr && s_live) atomicAdd(&A.stats[4], (unsigned long long)s_live);
  }
  __syncthreads();
  // the queue entry of a ray is its trace record (art_kernels.h): the trace kernel's per-ray setup, done here at one ray per lane.
  // A lane builds the four 16-byte pieces of its ray's record; they go through LDS so that every store instruction of a wave writes
  // 1 KB of contiguous bytes (64 lanes storing 16 bytes at a 64-byte stride cost the vector-memory path 4x as much, profiles/ta_rate.hip;
  // the kernel's TA units were 81 % busy: 3.1-3.3 -> 2.7 ms per launch on C4).
  const int lane2 = threadIdx.x & 63, wave2 = threadIdx.x >> 6;
  for (int k0 = 0; k0 < s_count; k0 += 256) {
    const int k = k0 + threadIdx.x;
    if (k < s_count) {
      const int i = chunk0 + (int)s_idx[k];
      const f3 o = mk3(A.ray_ox[i], A.ray_oy[i], A.ray_oz[i]), d = mk3(A.ray_dx[i], A.ray_dy[i], A.ray_dz[i]);
      const DevHit h0 = A.hit[i];
      float bt = h0.t; uint32_t bk = h0.key;
      const float shm = (A.sh_min != nullptr && i >= A.shadow_begin) ? A.sh_min[i - A.shadow_begin] : -1.0f;
      f3 inv, noi; slab_setup(o, d, inv, noi);
      const bool far_found = (shm >= 0.0f) && (bk != KEY_MISS);
      bt = far_found ? next_up_pos(shm) : bt;
      bk = far_found ? KEY_MISS : bk;
      const uint32_t sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
      const uint32_t sel_near = (sx ? 3u : 0u) | ((sy ? 4u : 1u) << 8) | ((sz ? 5u : 2u) << 16) | 0x0c000000u;
      s_rec[0][threadIdx.x] = make_float4(o.x, o.y, o.z, bt);
      s_rec[1][threadIdx.x] = make_float4(d.x, d.y, d.z, __builtin_bit_cast(float, bk));
      s_rec[2][threadIdx.x] = make_float4(inv.x, inv.y, inv.z, shm);
      s_rec[3][threadIdx.x] = make_float4(__builtin_bit_cast(float, sel_near), __builtin_bit_cast(float, i), __builtin_bit_cast(float, far_found ? 1u : 0u), 0.0f);
    }
    wave_lds_sync();                                                          // a wave only reads back the 64 records it wrote itself
    const int w0 = k0 + wave2 * 64;
    float4* out = A.rec + 4 * (size_t)(s_base + w0);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int q = it * 64 + lane2;                                          // piece (q & 3) of the wave's record (q >> 2)
      if (w0 + (q >> 2) < s_count) out[q] = s_rec[q & 3][wave2 * 64 + (q >> 2)];
    }
    wave_lds_sync();
  }
}

// rays of a launch that bypasses k_analytic (one-ray-per-lane kernel): count the live ones, one atomic per workgroup chunk
__global__ __launch_bounds__(256) void k_count_live(const TraceArgs A) {
  __shared__ int s_live;
  if (threadIdx.x == 0) s_live = 0;
  __syncthreads();
  int n = 0;
  for (int k0 = 0; k0 < kAnalyticChunk; k0 += 256) {
    const int i = blockIdx.x * kAnalyticChunk + k0 + threadIdx.x;
    n += (i < A.n_rays && A.ray_tfar[i] >= 0.0f) ? 1 : 0;
  }
  for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(&s_live, n);
  __syncthreads();
  if (threadIdx.x == 0 && s_live) atomicAdd(A.live_rays, (unsigned long long)s_live);
}

// ------------------------------------------------------------------------------------------------
// one ray per lane (cross-check / baseline)
// ------------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(256) void k_trace_simple(const DevScene* __restrict__ Sp, const TraceArgs A) {
  const DevScene& S = *Sp;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n_rays) return;
  const float tfar = A.ray_tfar[i];
  if (!(tfar >= 0.0f)) return;
  BvhStats st = {0, 0, 0, 0};
  const float shm = (A.sh_min != nullptr && i >= A.shadow_begin) ? A.sh_min[i - A.shadow_begin] : -1.0f;
  const Cand c = closest_hit<STATS>(S, mk3(A.ray_ox[i], A.ray_oy[i], A.ray_oz[i]), mk3(A.ray_dx[i], A.ray_dy[i], A.ray_dz[i]), tfar, &st, shm);
  A.hit[i] = DevHit{c.t, c.key, c.u, c.v};
  if (STATS) {
    atomicAdd(&A.stats[0], (unsigned long long)st.box_tests); atomicAdd(&A.stats[1], (unsigned long long)st.tri_tests);
    atomicAdd(&A.stats[2], (unsigned long long)st.node_visits); atomicAdd(&A.stats[3], (unsigned long long)st.leaf_visits);
    atomicAdd(&A.stats[4], 1ull);
  }
}

// ------------------------------------------------------------------------------------------------
// two-level search of the legacy geometry-core seam (art_instanced.h): one ray per lane
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_trace_instanced(const InstScene T, const float* __restrict__ o, const float* __restrict__ d,
                                                        const float* __restrict__ tfar, int n, InstHit* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = instanced_closest(T, mk3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), mk3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), tfar[i]);
}

// ------------------------------------------------------------------------------------------------
// wavefront stages
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_raygen(const DevFrame F, const DevScene S, const DevPaths Q) {
  __shared__ DevSphere s_sph[kAnalyticLdsSpheres];
  __shared__ DevLight s_lgt[kAnalyticLdsLights];
  StageCtx cx;
  if (Q.rec != nullptr && S.n_spheres <= kAnalyticLdsSpheres && S.n_lights <= kAnalyticLdsLights) {      // record mode: the camera ray's analytic intersection happens here
    if ((int)threadIdx.x < S.n_spheres) s_sph[threadIdx.x] = S.spheres[threadIdx.x];
    if ((int)threadIdx.x < S.n_lights) s_lgt[threadIdx.x] = S.lights[threadIdx.x];
    __syncthreads();
    cx.lds_spheres = as_lds(&s_sph[0]); cx.lds_lights = as_lds(&s_lgt[0]);
  }
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  // record mode: a wave's 64 records are 4 KB of consecutive bytes; staged through LDS so that every store instruction writes one
  // contiguous kilobyte (as in k_shade_compact)
  constexpr int kPitch = 64 + 1;
  __shared__ Rec4 s_stage[4][4 * kPitch];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool staged = (Q.rec != nullptr) && Q.has_bvh;
  if (staged) { cx.stage = s_stage[wave]; cx.stage_pitch = kPitch; cx.stage_item = lane; }
  cx.hot_layout = (Q.hot != nullptr);                    // record schedule: the hit record goes out through the bank's block, as a streaming store
  if (slot < Q.P) raygen_slot(F, S, Q, slot, cx);
  if (staged) {
    wave_lds_sync();
    const int first = slot - lane;
    const int n_pieces = 4 * max(0, min(64, Q.P - first));
    Rec4* out = Q.rec + 4 * (size_t)first;
    for (int pc = lane; pc < n_pieces; pc += 64) put_s(out, pc, s_stage[wave][(pc & 3) * kPitch + (pc >> 2)]);      // (read once, by the trace kernel: non-temporal)
  }
}

__global__ __launch_bounds__(256) void k_shade(const DevFrame F, const DevScene S, const DevPaths Q, int bounce) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < Q.P) shade_slot(F, S, Q, slot, bounce);
}

// ------------------------------------------------------------------------------------------------
// k_shade_compact: one bounce over a compacted work set.  Every stage reads the items of the paths that still need something (input
// set `Qi`: rays, hits, per-path state, item -> slot map) and writes only the items that will need something at the next stage to the
// output set `Qo`, densely: the kernels that follow (the trace kernel's record fetch, the next shade) then stream full 128-byte lines
// of live data instead of lines in which a quarter of the slots is alive.
//
// A workgroup handles a chunk of 256 x PER items in three steps (round 4):
//   1. classify: every item's class (item_class: its surface material, or CLS_CHEAP when the path ends at this hit), from the flags, the
//      hit key and the material type alone; per class, a wave reserves its items' places in the class with ONE LDS atomic (ballot +
//      prefix inside the wave);
//   2. sort: the chunk's items are laid out class by class in LDS (s_perm), surface classes first -- they are the survivors: ONE global
//      atomic reserves their range of the output set, item r of the sorted chunk goes to output item base + r;
//   3. shade in sorted order: thread t takes sorted items t, t + 256, ...: a wave's 64 items are consecutive in the sorted chunk, so
//      they are of ONE class except where two classes meet.  Before the sort a wave ran every material present in it with that material's
//      lanes only: 31 of 64 lanes enabled per VALU instruction (C3, C4), 27 (C5), e.g. on C4 71 % of the waves walked the Phong code
//      (four binary64 pow) for three lanes on the back wall (profiles/r4_shade/lane_probe_*.txt).
// The survivors of a wave's round are consecutive output items: their trace records are one contiguous piece of the output bank, the
// extension rays first, then the shadow rays (REC_BOTH).  The wave builds the records of one kind (make_record: analytic primitives
// intersected = the starting bound, slab set-up), stages them in LDS and copies them out 16 bytes per lane to consecutive addresses.
// In the shipped configuration (ART_SHADE_DEFER = 0) that copy-out sits inside shade_item's emit_ray: every kept item of a wave is a
// surface item and takes the surface branch together, which rests on the compiler keeping ONE call site for it; it is therefore checked
// at run time -- every record copied must name the hit slot its position implies, else the stage counts a lost path, and a lost path
// fails art_synchronize.  ART_SHADE_DEFER = 1 builds and copies the records after shade_item at a point every lane of the
// wave reaches (30 more VGPRs: slower).
// ------------------------------------------------------------------------------------------------
// items per workgroup = 256 x kShadePerThread, one global atomic per workgroup.  Rounds 1-2 used 16 per thread (a single hot word serves
// ~88 atomics / us); measured in round 3 with the fused stage: 4 per thread is faster at every size -- a thread's items one after the other
// are its critical path (C2, 4 M paths per step: 3.87 -> 2.03 ms per step; C3 +2.8 %, C5 +2.9 %, C4 +0.6 %), and 130 k atomics per launch of
// the largest batch still spread over 6 ms.
#ifndef ART_SHADE_PER
#define ART_SHADE_PER 4
#endif
constexpr int kShadePerThread = ART_SHADE_PER;
#ifndef ART_SHADE_DEFER
#define ART_SHADE_DEFER 0         // 1: the kernel builds and copies the records after shade_item, at a wave-uniform point (costs 30 more VGPRs: spills at 6 waves per SIMD, +25 % launch time); 0: shade_item's own emit_ray does (round 3)
#endif
#ifndef ART_SHADE_HINT
#define ART_SHADE_HINT 1          // 1: the classification's key and material index go along to shade_item (all of an item's loads start at once)
#endif
#ifndef ART_SHADE_SORT
#define ART_SHADE_SORT 1          // 1: sort the items of a round by material class (see above); 0: input order (one class for all surfaces).  A/B in one call on the final stage (profiles/r4_shade/ab_sort_final.txt): C4 5.28 vs 5.70 ms per launch, C3 1.70-1.83 vs 1.88-1.90, C5 33.4 vs 33.7 ms per batch
#endif

#ifndef ART_SHADE_WAVES
#define ART_SHADE_WAVES 6
#endif
// CAMERA: bounce 0 over raygen's bank (DevPaths::synth0: flags, previous pdf and the camera ray are recomputed, not read) -- its own
// instantiation, so that the other bounces' code is exactly what it was (as one kernel the extra branch cost 17 more spilled VGPRs: +7 %)
// The stage's arguments as ONE struct: about 240 dwords of scene header and array pointers.  Taken by value the compiler loads all of them
// into SGPRs at the kernel's entry and keeps them live to its end: 320 of them spilled to VGPR lanes (v_writelane / v_readlane: 15 % of the
// kernel's VALU instructions, and the VGPRs that hold them).  ART_SHADE_KERNARG = 1: the code reads the fields through a pointer to the
// kernarg segment instead, laundered at the head of every round so that a field is a scalar load (scalar cache) where it is used:
// 80 spilled SGPRs instead of 320, 0-2 spilled VGPRs instead of 5-7, 6 % fewer VALU instructions; A/B in one call
// (profiles/r4_shade/ab_kernarg.txt): C4 5.01 vs 5.26 ms per launch (bounce 0: 7.4 vs 8.0), C3 1.58 vs 1.82 (2.98 vs 3.61), C5 3.61 vs 3.67.
// (0: the struct by value -- which as ONE parameter allocates worse than the eleven parameters of before: 29 spilled VGPRs, 6.0 ms.)
struct ShadeKernArgs {
  DevFrame F; DevScene S; DevPaths Qi; DevPaths Qo; int bounce;
  const int* n_in_ptr; int* n_out_ptr; uint32_t* slot_out; unsigned long long* lost; unsigned long long* rays_a; unsigned long long* rays_b;
  uint4* heavy; int* n_heavy_ptr;          // the deferred items of the heavy material classes (SET_LIGHT appends, SET_HEAVY consumes) and their count
};
#ifndef ART_SHADE_KERNARG
#define ART_SHADE_KERNARG 1
#endif
static_assert(sizeof(ShadeKernArgs) <= 4096 && __is_trivially_copyable(ShadeKernArgs), "ShadeKernArgs is the kernel's ONLY parameter: it sits at offset 0 of the kernarg segment");
typedef const __attribute__((address_space(4))) ShadeKernArgs* ShadeKArgs;
__device__ __forceinline__ ShadeKArgs launder_kargs(ShadeKArgs k) { unsigned long long r = (unsigned long long)k; asm volatile("" : "+s"(r)); return (ShadeKArgs)r; }

// ---- one instantiation per REGISTER CLASS (round 5).  Until round 4 ONE kernel carried every material: it sat at the 80-VGPR cap of 6 waves
// per SIMD with spills, for a Lambert majority that needs none of Phong's four binary64 pow evaluations or the glass branch.  Now:
//   SET_LIGHT   shades the classes outside kHeavyClasses (Lambert, mirror, and the paths that end: CLS_CHEAP) and contains no instruction of
//               the heavy materials; the heavy items it meets are appended, { item, hit key, material | class << 24 }, to a queue in HBM
//               (one global atomic per workgroup, 16 B per deferred item);
//   SET_HEAVY   a second launch over that queue: the same three steps (sort by class, reserve, shade), compiled for the heavy materials only.
//               Its survivors follow the light kernel's in the output bank (the order of the items of a bank carries no meaning: item -> slot
//               map, fold records linked by child index).
//   SET_ALL     the round-4 kernel (every class in one instantiation), kept for A/B: -DART_SHADE_SPLIT=0.
enum ShadeSet : int { SET_ALL = 0, SET_LIGHT = 1, SET_HEAVY = 2 };
#ifndef ART_SHADE_SPLIT
#define ART_SHADE_SPLIT 1
#endif
#ifndef ART_HEAVY_CLASSES
#define ART_HEAVY_CLASSES ((1 << CLS_PHONG) | (1 << CLS_GLASS))
#endif
constexpr int kHeavyClasses = ART_HEAVY_CLASSES;
constexpr int kHeavyMats = ((kHeavyClasses >> CLS_PHONG & 1) ? mat_bit(MAT_PHONG) : 0) | ((kHeavyClasses >> CLS_GLASS & 1) ? mat_bit(MAT_GLASS) : 0) |
                           ((kHeavyClasses >> CLS_MIRROR & 1) ? mat_bit(MAT_MIRROR) : 0) | ((kHeavyClasses >> CLS_LAMBERT & 1) ? mat_bit(MAT_LAMBERT) : 0);
constexpr int set_mats(int set) { return set == SET_ALL ? kMatsAll : set == SET_LIGHT ? (kMatsAll & ~kHeavyMats) : (kHeavyMats | mat_bit(MAT_NULL)); }
constexpr bool class_in_set(int set, int q) { return set == SET_ALL ? true : set == SET_LIGHT ? !(kHeavyClasses >> q & 1) : (kHeavyClasses >> q & 1) != 0; }
#ifndef ART_SHADE_LIGHT_HINT
#define ART_SHADE_LIGHT_HINT 1
#endif
#ifndef ART_SHADE_WAVES_LIGHT
#define ART_SHADE_WAVES_LIGHT ART_SHADE_WAVES
#endif
#ifndef ART_SHADE_WAVES_HEAVY
#define ART_SHADE_WAVES_HEAVY ART_SHADE_WAVES
#endif
constexpr int set_waves(int set) { return set == SET_LIGHT ? ART_SHADE_WAVES_LIGHT : set == SET_HEAVY ? ART_SHADE_WAVES_HEAVY : ART_SHADE_WAVES; }

template <int PER, bool CAMERA, int SET>
// SET_ALL: 6 waves per SIMD (80 VGPRs): left to itself the compiler takes 111 VGPRs = 4 waves (6.6 -> 6.2 ms per launch on C4, round 3)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(set_waves(SET)))) void k_shade_compact(const ShadeKernArgs A_by_value) {
#if ART_SHADE_KERNARG
  ShadeKArgs K = launder_kargs((ShadeKArgs)__builtin_amdgcn_kernarg_segment_ptr());
#define ART_KREF(T, field) (*(const T*)&K->field)
#else
  const ShadeKernArgs* const K = &A_by_value;
#define ART_KREF(T, field) (K->field)
#endif
  constexpr int MATS = set_mats(SET);
  const DevFrame& F = ART_KREF(DevFrame, F); const DevScene& S = ART_KREF(DevScene, S); const DevPaths& Qi = ART_KREF(DevPaths, Qi); const DevPaths& Qo = ART_KREF(DevPaths, Qo);
  const int bounce_arg = K->bounce;
  const int* const n_in_ptr = K->n_in_ptr; int* const n_out_ptr = K->n_out_ptr;
  unsigned long long* const rays_a = K->rays_a; unsigned long long* const rays_b = K->rays_b;
  const int bounce = CAMERA ? 0 : bounce_arg;            // (a literal for the camera instantiation: no shadow test can be owed, nothing is pending)
  constexpr int kShadeChunk = 256 * PER;
  __shared__ int s_tot[PER * kItemClasses];                // items of each class in each round (256 items) of the chunk
  __shared__ int s_base, s_rays, s_hbase;
  __shared__ int s_nall[PER], s_nkeep[PER], s_out0[PER];   // per round: items, survivors, first output item (relative to s_base)
  // [round][sorted position] -> { hit key, thread that classified the item | surface flag << 8 | material index << 9 }: what the
  // classification fetched goes along (ItemHint), so that shade_item asks for the triangle's normals and the material record at once
  constexpr bool kHints = (ART_SHADE_HINT != 0) && !(SET == SET_LIGHT && ART_SHADE_LIGHT_HINT == 0);      // SET_LIGHT without hints: one byte per item (the thread that classified it) -- 1 KB instead of 8: with it the workgroup's LDS fits 8 times into a CU
  __shared__ typename std::conditional<kHints, uint2, uint8_t>::type s_hint[kShadeChunk];
  // a wave's trace records of one kind, [quarter][record] (+1: four consecutive records' quarters fall into four banks)
  constexpr int kStagePitch = 64 + 1;
  __shared__ Rec4 s_stage[4][4 * kStagePitch];
  constexpr int kShadeLdsLights = 8;                    // (8, not k_analytic's 16: with the hints the workgroup's LDS must stay below 160 KB / 6)
  __shared__ DevSphere s_sph[kAnalyticLdsSpheres];      // the scene's spheres and lights, fetched once per workgroup (every emitted ray is tested against them)
  __shared__ DevLight s_lgt[kShadeLdsLights];
  // the input set: the bank's items (n_in of them), or -- SET_HEAVY -- the entries of the deferred queue
  const int n_in = (SET == SET_HEAVY) ? *K->n_heavy_ptr : (n_in_ptr ? *n_in_ptr : Qi.P);
  const int c0 = blockIdx.x * kShadeChunk;
  if (c0 >= n_in) return;                                // the grid covers Qi.P items; the work set has shrunk to n_in
  // (the material table stays in global memory: a 40-byte per-lane-indexed record out of LDS measured 7 % slower than the cached global read)
  const bool tables_in_lds = (S.n_spheres <= kAnalyticLdsSpheres) && (S.n_lights <= kShadeLdsLights);
  if (tables_in_lds) {
    if ((int)threadIdx.x < S.n_spheres) s_sph[threadIdx.x] = S.spheres[threadIdx.x];
    if ((int)threadIdx.x < S.n_lights) s_lgt[threadIdx.x] = S.lights[threadIdx.x];
  }
  if (threadIdx.x < PER * kItemClasses) s_tot[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_rays = 0;
  __syncthreads();
  StageCtx tables;
  tables.hot_layout = true;
  if (tables_in_lds) { tables.lds_spheres = as_lds(&s_sph[0]); tables.lds_lights = as_lds(&s_lgt[0]); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t lanes_below = (1ull << lane) - 1ull;
#if defined(ART_TIME_PROBE)
  __shared__ unsigned long long s_tprobe[4][2];        // per wave: { time of the previous probe, address of the workgroup's table }
  __shared__ unsigned long long s_tacc[2 * 32];
  if (threadIdx.x < 64) s_tacc[threadIdx.x] = 0;
  if (lane == 0) { s_tprobe[wave][0] = __builtin_readcyclecounter(); s_tprobe[wave][1] = (unsigned long long)(uintptr_t)s_tacc; }
  __syncthreads();
  tables.tprobe = &s_tprobe[wave][0];
#endif
  // ---- 1. classify; a wave's items of one class take consecutive places in the class (of their round)
  int cls[PER], rank[PER]; ItemHint hint[PER];
  {
    // the PER classifications as ONE batch of loads per step (item_classes: 3 memory round trips; item by item through item_class they were 20)
    int wk[PER]; bool on[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) { wk[k] = c0 + k * 256 + (int)threadIdx.x; on[k] = wk[k] < n_in; hint[k].key = KEY_MISS; hint[k].mat = 0; cls[k] = kItemClasses; }
    if (SET == SET_HEAVY) {
      uint4 e[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) e[k] = K->heavy[on[k] ? wk[k] : 0];
#pragma unroll
      for (int k = 0; k < PER; ++k) if (on[k]) { hint[k].key = e[k].y; hint[k].mat = (int32_t)(e[k].z & 0xffffffu); cls[k] = (int)(e[k].z >> 24); }
    } else item_classes<PER>(S, Qi, wk, on, tables, CAMERA, cls, hint);
  }
  ART_TPROBE(tables.tprobe, 64);      // classification loads
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int c = cls[k];
    const bool surface = c < CLS_CHEAP;
    if (SET == SET_LIGHT && c < kItemClasses && !class_in_set(SET_LIGHT, c)) hint[k].mat |= c << 24;          // (a deferred item's queue entry carries its class)
    else hint[k].mat = (int32_t)(threadIdx.x | (surface ? 256u : 0u) | ((uint32_t)hint[k].mat << 9));
    if (!ART_SHADE_SORT && SET == SET_ALL && c < CLS_CHEAP) c = CLS_LAMBERT;
    cls[k] = c; rank[k] = 0;
#pragma unroll
    for (int q = 0; q < kItemClasses; ++q) {
      const uint64_t m = __builtin_amdgcn_ballot_w64(c == q);
      if (m != 0) {                                        // wave-uniform
        int b = 0;
        if (lane == 0) b = atomicAdd(&s_tot[k * kItemClasses + q], (int)__popcll(m));
        b = __builtin_amdgcn_readfirstlane(b);
        if (c == q) rank[k] = b + (int)__popcll(m & lanes_below);
      }
    }
  }
  ART_TPROBE(tables.tprobe, 65);      // class ranks (ballots + LDS atomics)
  __syncthreads();
  ART_TPROBE(tables.tprobe, 66);      // barrier 1
  // ---- 2. sort every round: class q starts where the classes before it end; the surface classes (the survivors) come first.
  // The sort stays inside a round's 256 items: the four waves of the workgroup then read one 1-KB window of every array at the same time
  // (sorted over the whole chunk, a wave's loads touched lines whose other halves were fetched again three rounds later: +43 % bytes
  // fetched, +22 % written back, 16-27 % slower although it ran 29 % fewer VALU instructions -- profiles/r4_shade/ab_sort_chunk.txt)
  // SET_LIGHT: the classes of kHeavyClasses are left out of the rounds; their items go to the deferred queue, class by class and round by round
  const bool keeps = stage_keeps_surfaces(F, bounce);
  if (SET == SET_LIGHT) {
    if (threadIdx.x == 0) {
      int nh = 0;
      for (int k = 0; k < PER; ++k)
        for (int q = 0; q < kItemClasses; ++q) if (!class_in_set(SET_LIGHT, q)) nh += s_tot[k * kItemClasses + q];
      s_hbase = nh ? atomicAdd(K->n_heavy_ptr, nh) : 0;
    }
  }
  int total_keep = 0, total_heavy = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    int st = 0, mine = 0, nkeep = 0, hv = total_heavy, hmine = 0;
#pragma unroll
    for (int q = 0; q < kItemClasses; ++q) {
      if (q == CLS_CHEAP) nkeep = keeps ? st : 0;
      const int nq = s_tot[k * kItemClasses + q];
      if (SET == SET_LIGHT && !class_in_set(SET_LIGHT, q)) { hmine = (cls[k] == q) ? hv : hmine; hv += nq; continue; }
      mine = (cls[k] == q) ? st : mine;
      st += nq;
    }
    if (threadIdx.x == 0) { s_nall[k] = st; s_nkeep[k] = nkeep; s_out0[k] = total_keep; }
    total_keep += nkeep; total_heavy = hv;
    const bool deferred = (SET == SET_LIGHT) && cls[k] < kItemClasses && !class_in_set(SET_LIGHT, cls[k]);
    if (deferred) rank[k] += hmine - (1 << 30);           // its place in the workgroup's piece of the queue (marked: written once s_hbase is known)
    else if (cls[k] < kItemClasses) {
      if constexpr (kHints) s_hint[k * 256 + mine + rank[k]] = make_uint2(hint[k].key, (uint32_t)hint[k].mat);
      else s_hint[k * 256 + mine + rank[k]] = (uint8_t)threadIdx.x;
    }
  }
  if (threadIdx.x == 0) s_base = total_keep ? atomicAdd(n_out_ptr, total_keep) : 0;
  __syncthreads();
  if (SET == SET_LIGHT) {
    const int hbase = s_hbase;
#pragma unroll
    for (int k = 0; k < PER; ++k)
      if (rank[k] < 0) K->heavy[hbase + rank[k] + (1 << 30)] = make_uint4((uint32_t)(c0 + k * 256 + (int)threadIdx.x), hint[k].key, (uint32_t)hint[k].mat, 0u);
  }
  const int base = s_base;
  ART_TPROBE(tables.tprobe, 67);      // sort, output reservation (global atomic), barrier 2
  // ---- 3. shade in sorted order
  const bool staged = (Qo.rec != nullptr) && Qo.has_bvh;
  const int mode = Qo.rec_mode;
  StageCtx cx = tables;
  int n_rays = 0;
#pragma unroll 1
  for (int k = 0; k < PER; ++k) {
#if ART_SHADE_KERNARG
    const ShadeKArgs K2 = launder_kargs(K);               // the round's own view of the arguments: nothing loaded before survives in a register
    const DevFrame& F = *(const DevFrame*)&K2->F; const DevScene& S = *(const DevScene*)&K2->S; const DevPaths& Qi = *(const DevPaths*)&K2->Qi; const DevPaths& Qo = *(const DevPaths*)&K2->Qo;
    unsigned long long* const lost = K2->lost;            // (the output items' slot words are written by shade_item: Qo's HF_SLOT field IS K->slot_out)
#else
    unsigned long long* const lost = K->lost;
    const ShadeKernArgs* const K2 = K;
#endif
    ART_TPROBE(tables.tprobe, 70);    // round bookkeeping (and, after the first round, whatever followed the last probe of the item before)
    // Which 64 sorted positions of the round this wave takes: tile (wave + k) mod 4 (round 5).  A sorted round is [Lambert ...][Phong][glass]
    // [mirror][cheap]: its last tile is where the classes meet -- on C4 a few Lambert lanes, the round's 13 Phong items and its ~33 ended
    // paths, three code paths one after the other, about three times a one-class tile's time.  With tile = wave that tile went to wave 3 of
    // every workgroup in every round, i.e. to the same SIMD of the CU (a workgroup's four waves sit on its four SIMDs); rotated, every wave
    // gets it once per chunk.
#ifndef ART_SHADE_ROTATE
#define ART_SHADE_ROTATE 0   // measured (profiles/r5_shade/ab8_rotate.txt): no gain on C3 / C5, bounce 0 4 % slower on C4 / S4 -- the hardware does not pin wave 3 to one SIMD
#endif
    const int tile = ART_SHADE_ROTATE ? ((wave + k) & 3) : wave;
    const int r = tile * 64 + lane;                       // position in the sorted round
    const int n_all_k = s_nall[k], n_keep_k = s_nkeep[k], out0_k = s_out0[k];
    const bool keep = r < n_keep_k;
    const int wo = keep ? base + out0_k + r : -1;
    RayOut ro; ro.alive = false; ro.shadow = false; ro.no = ro.nd = ro.so = ro.sd = mk3(0.0f, 0.0f, 0.0f); ro.s_tfar = -1.0f; ro.sh_min = 0.0f;
    if (r < n_all_k) {
      uint2 hk;
      if constexpr (kHints) hk = s_hint[k * 256 + r]; else hk = make_uint2(KEY_MISS, (uint32_t)s_hint[k * 256 + r]);
      int w = c0 + k * 256 + (int)(hk.y & 255u);
      if (SET == SET_HEAVY) w = (int)K2->heavy[w].x;       // the queue entry names the item
      ItemHint ih; ih.key = hk.x; ih.mat = (ART_SHADE_HINT && (hk.y & 256u)) ? (int32_t)(hk.y >> 9) : -1;      // mat < 0: no hint (shade_item)
      const ItemHint* const hp = &ih;
      // (the output item's slot word is written by shade_item with the item's other words: qo.slot_id IS slot_out)
      if (ART_SHADE_DEFER) n_rays += shade_item<MATS>(F, S, Qi, Qo, w, wo, bounce, lost, cx, &ro, hp, CAMERA ? 1 : 0, 1);
      else {
        StageCtx cy = tables;
        if (staged) {
          const int nk = min(64, max(0, n_keep_k - tile * 64)), per = (Qo.rec_mode == REC_BOTH) ? 2 : 1;
          const size_t first = (size_t)per * (size_t)(base + out0_k + tile * 64);
          cy.stage = s_stage[wave]; cy.stage_pitch = kStagePitch; cy.stage_item = lane; cy.stage_count = nk; cy.lost = lost;
          cy.rec_base[0] = first; cy.rec_base[1] = (per == 2) ? first + (size_t)nk : first;
        }
        n_rays += shade_item<MATS>(F, S, Qi, Qo, w, wo, bounce, lost, cy, nullptr, hp, CAMERA ? 1 : 0, 1);
      }
    }
    if (ART_SHADE_DEFER && Qo.rec != nullptr) {
      // every lane of the wave is here.  The wave's survivors of this round are its lanes [0, nk): consecutive output items.
      const int r0 = tile * 64;
      const int nk = min(64, max(0, n_keep_k - r0));
      if (nk > 0) {                                        // wave-uniform
        const int per = (mode == REC_BOTH) ? 2 : 1;
        const size_t first = (size_t)per * (size_t)(base + out0_k + r0);
#pragma unroll
        for (int kind = 0; kind < 2; ++kind) {             // 0: extension rays, 1: shadow rays
          if ((kind == 0 && mode == REC_SHADOW) || (kind == 1 && mode == REC_EXT)) continue;
          TraceRec t;
          if (keep) {
            t = (kind == 0) ? make_record(S, Qo, (size_t)wo, ro.alive, ro.no, ro.nd, kInfinity, -1.0f, cx)
                            : make_record(S, Qo, Qo.sh_t ? (size_t)(kShadowWord | (uint32_t)wo) : (size_t)Qo.P + (size_t)wo, ro.shadow, ro.so, ro.sd, ro.s_tfar, Qo.shadow_rule ? ro.sh_min : -1.0f, cx);
          }
          if (staged) {
            Rec4* const stage = s_stage[wave];
            if (keep) { stage[lane] = t.r0; stage[kStagePitch + lane] = t.r1; stage[2 * kStagePitch + lane] = t.r2; stage[3 * kStagePitch + lane] = t.r3; }
            wave_lds_sync();
            Rec4* out = Qo.rec + 4 * (first + ((kind == 1 && per == 2) ? (size_t)nk : 0));
            for (int g = lane; g < 4 * nk; g += 64) out[g] = stage[(g & 3) * kStagePitch + (g >> 2)];
            wave_lds_sync();
          }
        }
      }
    }
  }
  ART_TPROBE(tables.tprobe, 68);      // end of the last round
#if defined(ART_TIME_PROBE)
  __syncthreads();
  if (threadIdx.x < 64 && s_tacc[threadIdx.x] != 0) atomicAdd(&g_lane_probe[2 * 64 + threadIdx.x], s_tacc[threadIdx.x]);
#endif
  // record mode: the rays just emitted are the closest-hit queries of the next trace launch (k_analytic counts them in the plain layout)
  if (Qo.rec != nullptr && rays_a != nullptr) {
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_xor(n_rays, off);
    if (lane == 0 && n_rays) atomicAdd(&s_rays, n_rays);
    __syncthreads();
    if (threadIdx.x == 0 && s_rays) { atomicAdd(rays_a, (unsigned long long)s_rays); if (rays_b) atomicAdd(rays_b, (unsigned long long)s_rays); }
  }
}

__global__ void k_bump(unsigned long long* a, unsigned long long* b, unsigned long long n) { *a += n; if (b) *b += n; }
__global__ void k_acc_items(const int* live, int depth, int P, unsigned long long* items) {
  unsigned long long in = (unsigned long long)P;
  for (int b = 0; b < depth && b < 16; ++b) { const unsigned long long out = (unsigned long long)live[32 * (b + 1)]; items[b] += in; items[16 + b] += out; in = out; }
}

// the shadow tests still owed after the last trace, over the last work set; then the fold over all slots
__global__ __launch_bounds__(256) void k_resolve_last(const DevPaths Q, const int* __restrict__ n_ptr, int last_level) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < *n_ptr) resolve_last_shadow(Q, w, last_level);
}
__global__ __launch_bounds__(256) void k_fold(const DevFrame F, const DevPaths Q) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < Q.P) fold_slot(F, Q, slot);
}

// dense fold records: one level (art_shade.h fold_level_item).  n_ptr: the level's item count (nullptr: all Q.P slots = level 0)
__global__ __launch_bounds__(256) void k_fold_level(const DevFrame F, const DevPaths Q, int level, int deepest, const int* __restrict__ n_ptr,
                                                    const float* __restrict__ nxt_r, const float* __restrict__ nxt_g, const float* __restrict__ nxt_b,
                                                    float* __restrict__ cur_r, float* __restrict__ cur_g, float* __restrict__ cur_b) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = n_ptr ? *n_ptr : Q.P;
  if (w < n) fold_level_item(F, Q, level, w, deepest != 0, nxt_r, nxt_g, nxt_b, cur_r, cur_g, cur_b);
}

__global__ __launch_bounds__(256) void k_finish(const DevFrame F, const DevPaths Q, int last_level) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot < Q.P) finish_slot(F, Q, slot, last_level);
}

__global__ __launch_bounds__(256) void k_accumulate(const DevFrame F, const DevPaths Q, int samples_in_batch, float* accum) {
  const int pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl < Q.npix) accumulate_pixel(F, Q, pl, samples_in_batch, accum);
}

__global__ __launch_bounds__(256) void k_resolve(const float* accum, int n_pixels, float norm_c, uint32_t* screen) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pixels) screen[i] = resolve_pixel(mk3(accum[3 * (size_t)i], accum[3 * (size_t)i + 1], accum[3 * (size_t)i + 2]), norm_c);
}

// Debug_Ray_Tracing (ray_tracer.adb:208-238): palette colour of the primary hit's matId, plus raw ids
__global__ __launch_bounds__(256) void k_debug(const DevFrame F, const DevScene S, const DevPaths Q, float* accum,
                                               int32_t* prim_index, int32_t* mat_id, int32_t* prim_type) {
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= Q.P) return;
  uint32_t pixel, sample;
  slot_to_sample(Q, slot, pixel, sample);
  const DevHit hq = Q.hit[slot];
  const uint32_t key = hq.key;
  f3 col = mk3(0.0f, 0.0f, 0.0f);
  int32_t pi = -1, mi = -1, pt = -1;
  if (key != KEY_MISS) {
    const f3 o = mk3(Q.ray_ox[slot], Q.ray_oy[slot], Q.ray_oz[slot]), d = mk3(Q.ray_dx[slot], Q.ray_dy[slot], Q.ray_dz[slot]);
    const Surface sf = surface_at(S, o, d, hq.t, key, hq.u, hq.v);
    col = debug_palette(sf.mat_id);
    mi = sf.mat_id; pi = (int32_t)(key & KEY_INDEX_MASK);
    const uint32_t cls = key & ~KEY_INDEX_MASK;   // geometry.ads:55 Primitive'Pos: plane 0, sphere 1, triangle 2, quad 3
    pt = (cls == KEY_CORNELL) ? 0 : (cls == KEY_SPHERE) ? 1 : (cls == KEY_QUAD) ? 3 : 2;
  }
  accum[3 * (size_t)pixel] = col.x; accum[3 * (size_t)pixel + 1] = col.y; accum[3 * (size_t)pixel + 2] = col.z;
  if (prim_index) prim_index[pixel] = pi;
  if (mat_id) mat_id[pixel] = mi;
  if (prim_type) prim_type[pixel] = pt;
}

// layout conversion at the Ada seam: AccumBuff(x,y) / ScreenBufferData(x,y) are x-major (ray_tracer.ads:35,54)
__global__ __launch_bounds__(256) void k_to_xmajor_f3(const float* src_rowmajor, float* dst_xmajor, int w, int h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int x = i / h, y = i - x * h;
  const size_t s = (size_t)y * w + x;
  dst_xmajor[3 * (size_t)i] = src_rowmajor[3 * s]; dst_xmajor[3 * (size_t)i + 1] = src_rowmajor[3 * s + 1]; dst_xmajor[3 * (size_t)i + 2] = src_rowmajor[3 * s + 2];
}
__global__ __launch_bounds__(256) void k_to_xmajor_u32(const uint32_t* src_rowmajor, uint32_t* dst_xmajor, int w, int h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int x = i / h, y = i - x * h;
  dst_xmajor[i] = src_rowmajor[(size_t)y * w + x];
}
__global__ __launch_bounds__(256) void k_from_xmajor_f3(const float* src_xmajor, float* dst_rowmajor, int w, int h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  const int x = i / h, y = i - x * h;
  const size_t s = (size_t)y * w + x;
  dst_rowmajor[3 * s] = src_xmajor[3 * (size_t)i]; dst_rowmajor[3 * s + 1] = src_xmajor[3 * (size_t)i + 1]; dst_rowmajor[3 * s + 2] = src_xmajor[3 * (size_t)i + 2];
}

// 48-byte triangle records -> 64-byte padded records for the 4-wide trace kernel
__global__ __launch_bounds__(256) void k_pad_tris(const float* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * 16) return;
  const int t = i >> 4, k = i & 15;
  dst[i] = (k < kTriFloats) ? src[(size_t)t * kTriFloats + k] : 0.0f;
}

// dst[i] += src[i]: the framebuffer sum between two contexts that live on the same physical GPU (multi-device rehearsal; distinct
// GPUs use the RCCL reduce).  Written as src + dst: adding exact zeros in any order gives the same bits anyway.
__global__ __launch_bounds__(256) void k_add_f32(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] + dst[i];
}

// ------------------------------------------------------------------------------------------------
// launchers (the only symbols art_api.cpp needs from this translation unit)
// ------------------------------------------------------------------------------------------------
static inline int blocks_for(int n) { return (n + 255) / 256; }

void launch_raygen(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Q) {
  hipLaunchKernelGGL(k_raygen, dim3(blocks_for(Q.P)), dim3(256), 0, st, F, S, Q);
}
void launch_shade(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Q, int bounce) {
  hipLaunchKernelGGL(k_shade, dim3(blocks_for(Q.P)), dim3(256), 0, st, F, S, Q, bounce);
}
#if defined(ART_LANE_PROBE)
__global__ void k_probe_on(int v) { g_lane_probe_on = v; }
#endif
void launch_shade_compact(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Qi, const DevPaths& Qo, int bounce,
                          const int* n_in, int* n_out, uint32_t* slot_out, unsigned long long* lost, unsigned long long* rays_a, unsigned long long* rays_b,
                          uint4* heavy, int* n_heavy, int per) {
  const bool camera = Qi.synth0 && Qi.slot_id == nullptr;
  const int per_now = (per == 2 && !(ART_SHADE_SPLIT && heavy != nullptr && n_heavy != nullptr)) ? 2 : kShadePerThread;      // items per thread of this launch (2: the one-kernel form only)
  const dim3 grid((Qi.P + 256 * per_now - 1) / (256 * per_now));
#if defined(ART_LANE_PROBE)
  hipLaunchKernelGGL(k_probe_on, dim3(1), dim3(1), 0, st, 1);
#endif
  ShadeKernArgs A; A.F = F; A.S = S; A.Qi = Qi; A.Qo = Qo; A.bounce = bounce; A.n_in_ptr = n_in; A.n_out_ptr = n_out; A.slot_out = slot_out; A.lost = lost; A.rays_a = rays_a; A.rays_b = rays_b;
  A.heavy = heavy; A.n_heavy_ptr = n_heavy;
  const bool split = ART_SHADE_SPLIT && heavy != nullptr && n_heavy != nullptr;
  if (!split) {
    if (per_now == 2) {
      if (camera) hipLaunchKernelGGL((k_shade_compact<2, true, SET_ALL>), grid, dim3(256), 0, st, A);
      else hipLaunchKernelGGL((k_shade_compact<2, false, SET_ALL>), grid, dim3(256), 0, st, A);
    } else {
      if (camera) hipLaunchKernelGGL((k_shade_compact<kShadePerThread, true, SET_ALL>), grid, dim3(256), 0, st, A);
      else hipLaunchKernelGGL((k_shade_compact<kShadePerThread, false, SET_ALL>), grid, dim3(256), 0, st, A);
    }
  } else {
    // the light classes, then the deferred items of the heavy ones (their count is only known on the device: the grid covers the
    // worst case, workgroups past the queue's end return at once)
    const dim3 hgrid = grid;
    if (camera) {
      hipLaunchKernelGGL((k_shade_compact<kShadePerThread, true, SET_LIGHT>), grid, dim3(256), 0, st, A);
      hipLaunchKernelGGL((k_shade_compact<kShadePerThread, true, SET_HEAVY>), hgrid, dim3(256), 0, st, A);
    } else {
      hipLaunchKernelGGL((k_shade_compact<kShadePerThread, false, SET_LIGHT>), grid, dim3(256), 0, st, A);
      hipLaunchKernelGGL((k_shade_compact<kShadePerThread, false, SET_HEAVY>), hgrid, dim3(256), 0, st, A);
    }
  }
#if defined(ART_LANE_PROBE)
  hipLaunchKernelGGL(k_probe_on, dim3(1), dim3(1), 0, st, 0);
#endif
}
void launch_acc_items(hipStream_t st, const int* live, int depth, int P, unsigned long long* items) { hipLaunchKernelGGL(k_acc_items, dim3(1), dim3(1), 0, st, live, depth, P, items); }
void launch_bump(hipStream_t st, unsigned long long* a, unsigned long long* b, unsigned long long n) { hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, st, a, b, n); }
void launch_resolve_last(hipStream_t st, const DevPaths& Q, const int* n, int last_level) {
  hipLaunchKernelGGL(k_resolve_last, dim3(blocks_for(Q.P)), dim3(256), 0, st, Q, n, last_level);
}
void launch_fold(hipStream_t st, const DevFrame& F, const DevPaths& Q) {
  hipLaunchKernelGGL(k_fold, dim3(blocks_for(Q.P)), dim3(256), 0, st, F, Q);
}
// the whole fold over dense records: levels max_depth - 1 .. 0, ping-pong between the term arrays (odd levels) and rad (even levels: level 0
// ends in rad, where k_accumulate reads).  counts[32 * k] = item count of level k (k >= 1)
void launch_fold_levels(hipStream_t st, const DevFrame& F, const DevPaths& Q, int max_depth, const int* counts) {
  for (int k = max_depth - 1; k >= 0; --k) {
    float* cur[3] = {(k & 1) ? Q.term_r : Q.rad_r, (k & 1) ? Q.term_g : Q.rad_g, (k & 1) ? Q.term_b : Q.rad_b};
    const float* nxt[3] = {(k & 1) ? Q.rad_r : Q.term_r, (k & 1) ? Q.rad_g : Q.term_g, (k & 1) ? Q.rad_b : Q.term_b};
    hipLaunchKernelGGL(k_fold_level, dim3(blocks_for(Q.P)), dim3(256), 0, st, F, Q, k, k == max_depth - 1 ? 1 : 0, k == 0 ? nullptr : counts + 32 * k,
                       nxt[0], nxt[1], nxt[2], cur[0], cur[1], cur[2]);
  }
}
void launch_finish(hipStream_t st, const DevFrame& F, const DevPaths& Q, int last_level) {
  hipLaunchKernelGGL(k_finish, dim3(blocks_for(Q.P)), dim3(256), 0, st, F, Q, last_level);
}
void launch_accumulate(hipStream_t st, const DevFrame& F, const DevPaths& Q, int samples_in_batch, float* accum) {
  hipLaunchKernelGGL(k_accumulate, dim3(blocks_for(Q.npix)), dim3(256), 0, st, F, Q, samples_in_batch, accum);
}
void launch_resolve(hipStream_t st, const float* accum, int n_pixels, float norm_c, uint32_t* screen) {
  hipLaunchKernelGGL(k_resolve, dim3(blocks_for(n_pixels)), dim3(256), 0, st, accum, n_pixels, norm_c, screen);
}
void launch_debug(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Q, float* accum, int32_t* prim_index, int32_t* mat_id, int32_t* prim_type) {
  hipLaunchKernelGGL(k_debug, dim3(blocks_for(Q.P)), dim3(256), 0, st, F, S, Q, accum, prim_index, mat_id, prim_type);
}
void launch_to_xmajor_f3(hipStream_t st, const float* src, float* dst, int w, int h) {
  hipLaunchKernelGGL(k_to_xmajor_f3, dim3(blocks_for(w * h)), dim3(256), 0, st, src, dst, w, h);
}
void launch_to_xmajor_u32(hipStream_t st, const uint32_t* src, uint32_t* dst, int w, int h) {
  hipLaunchKernelGGL(k_to_xmajor_u32, dim3(blocks_for(w * h)), dim3(256), 0, st, src, dst, w, h);
}
void launch_from_xmajor_f3(hipStream_t st, const float* src, float* dst, int w, int h) {
  hipLaunchKernelGGL(k_from_xmajor_f3, dim3(blocks_for(w * h)), dim3(256), 0, st, src, dst, w, h);
}

void launch_trace_instanced(hipStream_t st, const InstScene& T, const float* o, const float* d, const float* tfar, int n, InstHit* out) {
  hipLaunchKernelGGL(k_trace_instanced, dim3((n + 63) / 64), dim3(64), 0, st, T, o, d, tfar, n, out);
}

void launch_pad_tris(hipStream_t st, const float* src, float* dst, int n) {
  hipLaunchKernelGGL(k_pad_tris, dim3((unsigned)(((size_t)n * 16 + 255) / 256)), dim3(256), 0, st, src, dst, n);
}

void launch_add_f32(hipStream_t st, const float* src, float* dst, size_t n) {
  hipLaunchKernelGGL(k_add_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
}

size_t trace_coop_lds_bytes(int stack_entries, int width) { return (size_t)4 * (64 / width) * (stack_entries + 3) * sizeof(uint2); }   // + 2 guards + sink

void launch_analytic(hipStream_t st, const DevScene& S, const TraceArgs& A, bool stats) {
  TraceArgs B = A;
  if (!stats) B.stats = nullptr;
  hipLaunchKernelGGL(k_analytic, dim3((A.n_rays + kAnalyticChunk - 1) / kAnalyticChunk), dim3(256), 0, st, S, B);
}

void launch_trace(hipStream_t st, const DevScene* S, const TraceArgs& A, int kernel, bool stats, int grid_blocks) {
  if (kernel == TRACE_SIMPLE) {
    hipLaunchKernelGGL(k_count_live, dim3((A.n_rays + kAnalyticChunk - 1) / kAnalyticChunk), dim3(256), 0, st, A);
    if (stats) hipLaunchKernelGGL(k_trace_simple<true>, dim3(blocks_for(A.n_rays)), dim3(256), 0, st, S, A);
    else hipLaunchKernelGGL(k_trace_simple<false>, dim3(blocks_for(A.n_rays)), dim3(256), 0, st, S, A);
    return;
  }
  if (A.n_tris <= 0) return;                       // no BVH mesh: the analytic pass (launch_analytic) is the whole search
  if (A.instanced == 2) {                          // instanced scene, option inst_coop = 0: the two-level kernel with one ray per lane over the same records
    const int blocks = std::max(1, std::min(grid_blocks * 4, 65535));
    if (stats) hipLaunchKernelGGL(k_trace_inst<true>, dim3(blocks), dim3(256), 0, st, S, A);
    else hipLaunchKernelGGL(k_trace_inst<false>, dim3(blocks), dim3(256), 0, st, S, A);
    return;
  }
  const size_t lds = trace_coop_lds_bytes(A.stack_entries, A.width);
  if (A.instanced == 1) {                          // instanced scene: the cooperative kernel crosses the instance boundary itself
    if (stats) { if (A.stack_overflow) hipLaunchKernelGGL((k_trace_coop<true, 4, true, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); else hipLaunchKernelGGL((k_trace_coop<true, 4, false, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); }
    else { if (A.stack_overflow) hipLaunchKernelGGL((k_trace_coop<false, 4, true, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); else hipLaunchKernelGGL((k_trace_coop<false, 4, false, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); }
    if (A.stack_overflow) {
      if (stats) hipLaunchKernelGGL(k_trace_overflow<true>, dim3(64), dim3(256), 0, st, S, A);
      else hipLaunchKernelGGL(k_trace_overflow<false>, dim3(64), dim3(256), 0, st, S, A);
    }
    return;
  }
  const int variant = (stats ? 4 : 0) | (A.width == 4 ? 2 : 0) | (A.stack_overflow ? 1 : 0);
  switch (variant) {
    case 0: hipLaunchKernelGGL((k_trace_coop<false, 8, false>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
    case 1: hipLaunchKernelGGL((k_trace_coop<false, 8, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
    case 2: hipLaunchKernelGGL((k_trace_coop<false, 4, false>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
    case 3: hipLaunchKernelGGL((k_trace_coop<false, 4, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
    case 4: hipLaunchKernelGGL((k_trace_coop<true, 8, false>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
    case 5: hipLaunchKernelGGL((k_trace_coop<true, 8, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
    case 6: hipLaunchKernelGGL((k_trace_coop<true, 4, false>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
    default: hipLaunchKernelGGL((k_trace_coop<true, 4, true>), dim3(grid_blocks), dim3(256), lds, st, S, A); break;
  }
  if (A.stack_overflow) {
    if (stats) hipLaunchKernelGGL(k_trace_overflow<true>, dim3(64), dim3(256), 0, st, S, A);
    else hipLaunchKernelGGL(k_trace_overflow<false>, dim3(64), dim3(256), 0, st, S, A);
  }
}

}  // namespace art

#if defined(ART_LANE_PROBE)
// diagnostic build: read (and clear) the lane probes of art_isect.h
extern "C" int art_debug_lane_probe(unsigned long long* out, int n) {
  unsigned long long z[2 * 96] = {};
  if (n > 2 * 96) n = 2 * 96;
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_probe), (size_t)n * 8) != hipSuccess) return 1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_lane_probe), z, sizeof z) != hipSuccess;
}
#endif

namespace art {
int trace_coop_blocks_per_cu(int stack_entries, int width) {
  int nb = 0;
  const size_t lds = trace_coop_lds_bytes(stack_entries, width);
  const hipError_t e = (width == 4) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_coop<false, 4, true>, 256, lds)
                                    : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_coop<false, 8, true>, 256, lds);
  if (e != hipSuccess || nb < 1) nb = 1;
  return nb;
}
}  // namespace art

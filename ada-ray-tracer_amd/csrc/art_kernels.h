// art_kernels.h -- launch interface between art_api.cpp (host driver) and art_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "art_shade.h"
#include "art_qnode.h"
#include "art_instanced.h"

namespace art {

enum { TRACE_COOP = 0, TRACE_SIMPLE = 1 };

struct TraceArgs {
  int32_t n_rays;
  int32_t stack_entries;        // per-ray LDS stack depth; >= Bvh8::max_stack unless stack_overflow
  int32_t width;                // node width = lanes per ray in k_trace_coop (8 or 4)
  int32_t instanced;            // DevScene::n_inst > 0: 1 = k_trace_coop<.., INST> walks the two-level tree (qnodes = art_instanced_build.cpp's array), 2 = k_trace_inst (one ray per lane; option inst_coop = 0)
  int32_t stack_overflow;       // the LDS stack is smaller than the tree's bound: pushes are checked, rays that do not fit go to ovf_queue
  int32_t segments;             // k_trace_coop: the queue is cut into this many contiguous segments (1, 2, 4, 8); workgroup b starts
                                // in segment b % segments (its XCD) and moves on to the next segment when that one is drained
  int32_t node_min;             // a wave keeps expanding nodes while at least this many of its 8 ray groups have one
  int32_t refill_min;           // k_trace_coop: idle groups take new rays only when at least this many of the wave's groups are idle (or nothing else is left to do)
  const float* ray_ox; const float* ray_oy; const float* ray_oz;
  const float* ray_dx; const float* ray_dy; const float* ray_dz;
  const float* ray_tfar;        // < 0: skip
  float4* rec;                  // k_analytic -> k_trace_coop: one 64-byte record per QUEUED ray, in queue order (layout below); n_rays + 4096 records (slack for the chunk prefetch)
  // shadow rays (index >= shadow_begin) only feed Compute_Shadow's test  10*eps < t_closest < tfar  (ray_tracer.adb:122):
  // sh_min[i - shadow_begin] = 10*eps.  nullptr: every ray is a closest-hit query.
  const float* sh_min; int32_t shadow_begin;
  DevHit* hit;
  const DevInstance* inst; int32_t inst_shift;   // instanced scene (k_trace_coop<.., INST>): the instance table; a hit's key index = instance << inst_shift | triangle
  float* sh_t;                  // record schedule: a record whose hit-slot word has kShadowWord set leaves its result as ONE float, sh_t[word & ~kShadowWord] = t of the hit (DevPaths::sh_t)
  const float* nodes; const uint32_t* qnodes; const float* tris; const float* qtris; int32_t n_tris;   // qnodes: 64-byte quantised nodes (width 4, art_qnode.h)   // BVH of the closest-hit mesh (hot-loop operands)
  int32_t chunk;                // trace records a wave claims per atomic on the cursor (a multiple of 16: prefetched 16 records per load)
  int* cursor;                  // work cursors, zeroed before every launch: segment k's cursor is cursor[32 * (k + 1)] (cursor[0] serves the
                                // kernels with a single cursor)
  int* queue; int* queue_count; // queue_count: number of trace records k_analytic queued, zeroed before every launch (queue: the same buffer as rec)
  // round 3: the wavefront stages write the records themselves, in item order (DevPaths::rec).  Then the queue length is
  // queue_fixed (>= 0: known on the host), or *queue_items x queue_mul (the output item count of the stage, 1 or 2 records per item)
  int32_t queue_fixed; const int* queue_items; int32_t queue_mul;
  int* ovf_queue; int* ovf_count;   // stack_overflow: rays handed to k_trace_overflow, count zeroed before every launch
  unsigned long long* stats;    // [box, tri, node, leaf, rays] when counting
  unsigned long long* live_rays;   // += closest-hit queries actually issued by this launch (Mrays/s numerator)
  const int* item_count;           // compacted work set: items [0, *item_count) exist (rays [0, n) and [shadow_begin, shadow_begin + n)); nullptr: all n_rays
};

// trace record (4 x float4) of a queued ray: everything k_trace_coop needs to start it, prepared at one ray per lane by k_analytic
//   [0] origin.xyz, starting bound t      [1] direction.xyz, starting bound key      [2] 1/direction (slab_setup), shadow-rule minimum (< 0: closest hit)
//   [3] near-plane byte selector (width 4), ray index, far_found, 0
constexpr int kTraceRecBytes = 64;

void launch_raygen(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Q);
void launch_bump(hipStream_t st, unsigned long long* a, unsigned long long* b, unsigned long long n);    // *a += n; if (b) *b += n
// end of a batch: items[b] += input items of bounce b, items[16 + b] += the items it kept (live[32 (b + 1)]); bounce 0 reads the P camera paths
void launch_acc_items(hipStream_t st, const int* live, int depth, int P, unsigned long long* items);
void launch_shade(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Q, int bounce);
void launch_finish(hipStream_t st, const DevFrame& F, const DevPaths& Q, int last_level);
// compacted work sets (k_shade_compact): shade the items of Qi, write the survivors densely to Qo (+ their slot ids); n_in nullptr: all Qi.P items
// rays_a / rays_b: += the rays the stage emits as trace records (Qo.rec != nullptr); rays_b may be nullptr
void launch_shade_compact(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Qi, const DevPaths& Qo, int bounce,
                          const int* n_in, int* n_out, uint32_t* slot_out, unsigned long long* lost, unsigned long long* rays_a, unsigned long long* rays_b,
                          uint4* heavy = nullptr, int* n_heavy = nullptr, int per = 0);      // per: items per thread, 2 or 0 = the default (4)      // heavy: the queue of deferred items (16 B x Qi.P), n_heavy: its count (zeroed by the caller); nullptr: one kernel for every material
void launch_resolve_last(hipStream_t st, const DevPaths& Q, const int* n, int last_level);
void launch_fold(hipStream_t st, const DevFrame& F, const DevPaths& Q);
void launch_fold_levels(hipStream_t st, const DevFrame& F, const DevPaths& Q, int max_depth, const int* counts);     // dense fold records: counts[32 k] = items of level k
void launch_accumulate(hipStream_t st, const DevFrame& F, const DevPaths& Q, int samples_in_batch, float* accum);
void launch_resolve(hipStream_t st, const float* accum, int n_pixels, float norm_c, uint32_t* screen);
void launch_debug(hipStream_t st, const DevFrame& F, const DevScene& S, const DevPaths& Q, float* accum, int32_t* prim_index, int32_t* mat_id, int32_t* prim_type);
void launch_to_xmajor_f3(hipStream_t st, const float* src, float* dst, int w, int h);
void launch_to_xmajor_u32(hipStream_t st, const uint32_t* src, uint32_t* dst, int w, int h);
void launch_from_xmajor_f3(hipStream_t st, const float* src, float* dst, int w, int h);
void launch_trace_instanced(hipStream_t st, const InstScene& T, const float* o, const float* d, const float* tfar, int n, InstHit* out);
void launch_pad_tris(hipStream_t st, const float* src12, float* dst16, int n);
void launch_add_f32(hipStream_t st, const float* src, float* dst, size_t n);
size_t trace_coop_lds_bytes(int stack_entries, int width);
void launch_trace(hipStream_t st, const DevScene* d_scene, const TraceArgs& A, int kernel, bool stats, int grid_blocks);
void launch_analytic(hipStream_t st, const DevScene& scene, const TraceArgs& A, bool stats);
int  trace_coop_blocks_per_cu(int stack_entries, int width);

}  // namespace art

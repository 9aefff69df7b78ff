// art_api_internal.h -- process-wide backend state (the reference keeps the same kind of singleton:
// g_data in embree_connect.cpp:12-22 and the package-level variables of ray_tracer.ads:20-33,106-109).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/art_hip.h"
#include "art_host_scene.h"
#include "art_kernels.h"

namespace art {

constexpr int kMaxDevices = 16;       // art_init_devices: GPUs one process may drive (an MI355X node has 8)
constexpr int kCursorInts = 32 * 9;   // d_cursor: 3 scalars + one work cursor per queue segment, each on its own 128-byte line

struct DevBuf {
  void* p = nullptr; size_t bytes = 0;
  // option paths_spread: the buffer is a reserved address range backed by separately created physical chunks (HIP virtual memory management)
  std::vector<hipMemGenericAllocationHandle_t> chunks; size_t chunk_bytes = 0, reserved = 0;
  void release();
};

struct Ctx {
  int device = -1; bool device_ready = false; int num_cus = 0; int lds_per_cu = 160 * 1024; std::string arch;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;            // multi-device mode: the stream this context created (stream == own_stream)
  // scene
  bool scene_ready = false;
  HostScene host_scene;
  DevScene scene;
  DevScene* d_scene = nullptr;                 // the same header in HBM (the trace kernels take it by pointer)
  BvhBuildParams bvh_params;
  DevBuf b_spheres, b_sphere_mat, b_lights, b_materials, b_bf_pos, b_bf_nrm, b_bf_uv, b_bf_idx, b_nodes, b_qnodes, b_tris, b_qtris, b_m_shade;
  int bvh_stack_bound = 8;      // worst-case traversal stack of the uploaded tree
  int lds_stack_cap = 0;        // option: force the LDS stack size (tests of the overflow path)
  // frame
  int width = 0, height = 0, spp = 0;
  int rank = 0, nranks = 1, tile = 32, npix_local = 0;
  DevBuf b_inst, b_tlas_nodes, b_tlas_tris, b_blas_nodes, b_blas_tris;      // instanced scene (DevScene::n_inst > 0)
  DevBuf b_accum, b_screen, b_stage, b_pixmap, b_paths, b_rays, b_ids, b_queue, b_ovf;
  int* d_live = nullptr;                       // item counts per level: d_live[32 k] = items of bounce k's input set (k >= 1); the dense fold walks them again
  DevBuf b_reduced;                            // device 0, multi-device mode: sum of every device's accum (the RCCL reduce target)
  float* ext_accum = nullptr;
  // work distribution / counters
  int* d_cursor = nullptr;
  unsigned long long* d_counters = nullptr;   // [0] rays issued by shade, [3..7] box, tri, node, leaf, rays of counting traces
  uint64_t camera_rays = 0;
  // options
  int trace_kernel = TRACE_COOP;
  int64_t batch_paths = 128ll << 20;  // path slots per batch (328 B each at depth 8 -> up to 44 GB of the 288 GB HBM; buffers are sized to the frame,
                                      // so a small render takes less): big batches keep late bounces wide (8M -> 32M: +7 %, -> 128M: +1.7 %)
  int opt_blocks_per_cu = 0, blocks_per_cu = 0;
  bool count_tests = false;
  int node_min = 0;              // the trace kernel leaves its node loop when fewer than 8 x node_min lanes still expand nodes.  0: 4 (the optimum on C4, profiles/r2_sensitivity.json);
                                 // 2 for an instanced scene -- its groups also wait for instance entries, and leaving the loop for them less often is worth 6 % on I64 (profiles/r5_inst_sweep.py)
  int refill_min = 2;            // idle ray groups of a wave take new rays when two of them are idle (1: at once; measured 0.5-1 % slower)
  int queue_segments = 8;       // the live-ray queue is cut into this many contiguous segments, one per XCD (1 = a single cursor)
  int ray_chunk = 48;            // trace records a wave claims per atomic (and prefetches): 16 -> 48 is worth 1 % on C4, 4 % on C3, 10 % on C5 (the claim stalls the wave)
  bool shadow_anyhit = true;   // shadow rays use the visibility rule instead of a full closest-hit search (same decision)
  // Items per thread of k_shade_compact (256 x per items and one global atomic per workgroup): 4 or 2.  Neither wins everywhere (inside one
  // process 2 is 5 % faster on C4, 3 % slower on C5, equal on S4: profiles/r6_shade/trial_check_*.txt; round 5's "+7 % / -10 %" of ab12 were
  // mostly the mapping of the path state, which differed between its processes), so by default the library MEASURES -- without ever
  // waiting for the GPU (round 6; until round 5 the trial blocked the host twice per trial batch, which serialised the devices of a
  // multi-GPU process during their first two batches): after a scene upload or a resize batch 0 runs warm and unmeasured (code-object load,
  // first touch of the path state), batch 1 with 4 items per thread, batch 2 with 2 -- their shade launches' event pairs carry the trial's
  // tag -- and every batch after that with 4 until a synchronize (art_synchronize, art_get_stats, a download: calls that wait anyway) has
  // read both trials' events; then the faster setting is kept.  Two trials only count when their batches had the same number of paths.
  // The picture does not depend on it (the order of the items of a bank carries no meaning).  Option shade_per: 0 automatic, 2, 4.
  int opt_shade_per = 0, auto_per = 0, auto_phase = 0;      // auto_phase: 0 warm batch next, 1 trial A (4) next, 2 trial B (2) next, 3 both enqueued, 4 decided
  double auto_ms[2] = {0.0, 0.0}; int64_t auto_P[2] = {0, 0}; int auto_redo = 0; unsigned auto_gen = 0;      // auto_gen: a trial's event pairs carry its generation (mod 4); a reset or a re-done trial starts a new one
  uint64_t lost_reported = 0;    // lost paths (d_counters[15], cumulative) already reported by a synchronize: only synchronize_one advances it
  int inject_lost = 0;           // test option: the next render pass bumps the self-check counter once in its first batch (tests/test_gpu_parity.py)
  // multi-device mode: host clock (steady_clock, ms since the process' first use) at which this device's stream reached the start / the end
  // of its passes -- hipLaunchHostFunc on the stream, so the devices' times share one clock (events of different devices cannot be compared)
  struct PassClock { double t0 = 0.0, t1 = 0.0; };
  std::vector<PassClock*> pass_clock;      // one per pass since the last art_get_reduce_info / resize (heap cells: the callbacks write into them)
  double busy_ms = 0.0, idle_ms = 0.0, start_skew_ms = 0.0;   // folded by art_get_reduce_info
  // HOW THE PATH STATE IS MAPPED decides the shade stage's rate (round 6, profiles/r6_bimodal).  The stage runs about forty concurrent streams
  // through the 35-74 GB of path state.  As ONE hipMalloc it took 10.4 or 12.1 ms per batch on C3 (31.7 / 35.9 on C4, 22.2 / 26.0 on C5, 28.0 /
  // 32.5 on S4) depending on the PROCESS -- same binary, same device addresses, same instructions, L2 hits, misses and fabric requests, 16 %
  // more L1 -> L2 read latency -- and 15.0 as physically contiguous memory: the larger the pieces the driver maps the range in, the slower.
  // paths_spread_mb: the path state as one address range over separately created physical chunks of that many MB (HIP virtual memory
  // management, art_api.cpp alloc_spread): the fast mode in every process measured, 13-17 ms to set up, no extra memory; any failure falls
  // back to hipMalloc.  -1 (default): 64 MB chunks when the path state is 1 GB or more; 0: plain hipMalloc; n > 0: n MB chunks always.
  // paths_spread_holes: a spacer chunk between two chunks, released after mapping (twice the memory for a moment; measured: not what helps).
  // paths_contiguous: hipExtMallocWithFlags(hipDeviceMallocContiguous) -- the slowest and the one deterministic placement (A/B tool).
  int paths_spread_mb = -1; bool paths_are_spread = false; bool paths_spread_holes = false;
  int spread_fail_at = -1;       // test option: the creation of this chunk is made to fail, so that the undo + hipMalloc fallback runs (tests/test_gpu_parity.py)
  bool paths_contiguous = false, paths_are_contiguous = false;
  int hot_pad = 0;               // items added to the stride between the fields of a bank's hot block (art_scene.h HotField): the frame sizes make that stride a multiple of 256 KB

  bool skip_null_shadow = false;   // DevFrame::skip_null_shadow: shadow rays that cannot change the picture are not traced (fewer rays than the reference issues: off by default)
  bool inst_coop = true;       // instanced scenes: the cooperative kernel crosses the instance boundary (k_trace_coop<.., INST>); false: k_trace_inst, one ray per lane (A/B, cross-check)
  bool shade_split = false;    // k_shade_compact as one instantiation per register class (light materials / deferred heavy ones); false: the round-4 kernel with every material (A/B)
  // timing
  std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
  std::vector<uint8_t> ev_kind;                // per event pair: 0 trace kernel, 1 shade stage, 2 raygen, 3 fold group (art_get_stage_stats) | trial (1 / 2: A / B of the items-per-thread trial) << 4 | the trial's generation << 6
  ArtStageStats stage = ArtStageStats();
  unsigned long long* d_items = nullptr;       // cumulative items_in[16] | items_out[16] (k_acc_items at the end of every batch)
  std::vector<hipEvent_t> pass_events;
  ArtStats stats = ArtStats();
};

// One context per GPU.  A single-GPU process only ever uses g_devs[0]; art_init_devices(n) fills n of them.  Every entry point holds
// g_mu, so "the current context" is a plain global that the multi-device loops repoint.
extern Ctx g_devs[kMaxDevices];
extern int g_ndev;
extern Ctx* g_cur;
#define g_ctx (*::art::g_cur)
extern std::mutex g_mu;

int fail(const std::string& msg);
int ensure_device();
int upload_scene(const ArtSceneDesc* d);
int resize(int w, int h);
int trace_rays(const float* origins, const float* dirs, const float* tfar, int64_t n, ArtHit* out, int kernel, ArtStats* st);
void shutdown();
int fetch_host_bvh(std::vector<float>& nodes, std::vector<float>& tris, int& width, int& n_tris);   // device 0's tree as host arrays (caller holds g_mu)

}  // namespace art

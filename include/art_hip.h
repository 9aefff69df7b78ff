/*
 * art_hip.h -- C ABI of libart_hip.so, the MI355X (gfx950) render backend for FROL256/ada-ray-tracer.
 *
 * Plain C types only (Interfaces.C.int / float / unsigned, System.Address on the Ada side).
 * Two groups of entry points:
 *
 *  1. Frame-level calls (art_*): what Ray_Tracer.Render_Pass (ray_tracer.adb:240-293) forwards to
 *     instead of waking its Path_Trace_Thread tasks -- the per-pixel DoPass loop
 *     (ray_tracer-integrators.adb:25-71), PathTrace x3 (integrators.adb:82-301),
 *     Scene.Find_Closest_Hit (scene.adb:56-86), Compute_Shadow (ray_tracer.adb:100-132), the accum
 *     and the gamma/tonemap/pack resolve (ray_tracer.adb:281-291) all run on the GPU.
 *
 *  2. The legacy geometry-core seam (gcore_*): the six symbols scene_hydra_embree.adb:37-66 imports
 *     and cpp/embree_connect.cpp:51-244 defines on top of Embree 3.7.  Same names and return types;
 *     counts are element counts (the reference passes 'Size in bits, scene_hydra_embree.adb:259-262).
 *
 * Error convention: art_* return 0 on success, non-zero on failure, text via art_last_error();
 * nothing in this library calls exit() (embree_connect.cpp:28-49 does).  gcore_* keep the reference's
 * return types (0 / false on failure).  Rendering and batched queries have NO CPU fallback: without a usable HIP device
 * every call that needs one fails with a message.  One documented host path exists on the legacy seam: a SINGLE-ray
 * gcore_closest_hit is answered on the calling thread by a walk of the committed tree (the product's own walker, the GPU kernels'
 * boxes and triangle arithmetic: gcore_set_single_ray_on_gpu below; SURVEY 8(b) -- a kernel launch per ray is what that call pattern
 * cannot afford).  The tree itself is always built and committed on the GPU box; nothing routes through the test oracle.
 *
 * Threading: art_* are single-caller (Render_Pass is only called from the environment task,
 * test.adb:50); gcore_closest_hit may be called concurrently (it is in the reference, from up to 28
 * tasks) and is serialised internally.
 */
#ifndef ART_HIP_H
#define ART_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flattened scene description (host pointers, copied during art_upload_scene) ------------- */

/* materials.ads:58-130 flattened: type tag + parameters */
enum { ART_MAT_NULL = 0, ART_MAT_LIGHT = 1, ART_MAT_LAMBERT = 2, ART_MAT_MIRROR = 3, ART_MAT_GLASS = 4, ART_MAT_PHONG = 5 };
typedef struct ArtMaterial {
  int32_t type;
  int32_t light;     /* LIGHT: index into lights (MaterialLight.lref) */
  float   p[8];      /* LAMBERT kd[3] | MIRROR reflection[3] | GLASS reflection[3] transparency[3] ior | PHONG reflection[3] cosPower */
} ArtMaterial;

/* lights.ads:36-55 */
enum { ART_LIGHT_RECT = 0, ART_LIGHT_SPHERE = 1 };
typedef struct ArtLight {
  int32_t shape;
  int32_t mat;       /* material-table index of the MaterialLight referring to this light */
  float boxMin[3], boxMax[3], normal[3];   /* AreaLight   */
  float center[3], radius;                 /* SphereLight */
  float intensity[3];
  float surfaceArea;
} ArtLight;

/* geometry.ads:21-25 */
typedef struct ArtSphere { float pos[3]; float r; int32_t mat; } ArtSphere;

/* geometry.ads:94-101.  mode selects the search semantics:
 *   ART_MESH_REFERENCE_BF : IntersectMeshBF verbatim (geometry.adb:266-323: bbox early-out, index-order
 *                           scan with the (t, t+1e-6) window, matId forced to 2) -- for the reference's own
 *                           8-triangle pyramid; O(N) per ray.
 *   ART_MESH_CLOSEST      : true closest hit through the BVH (the semantics Embree provides at
 *                           gcore_closest_hit), reference Moeller-Trumbore arithmetic, material_ids honoured. */
enum { ART_MESH_REFERENCE_BF = 0, ART_MESH_CLOSEST = 1 };
typedef struct ArtMesh {
  int32_t mode;
  int32_t nverts, ntris;
  const float*   pos;     /* 3*nverts, world space (LoadMeshFromVSGF transforms positions only) */
  const float*   nrm;     /* 3*nverts */
  const float*   uv;      /* 2*nverts, may be NULL (treated as zeros, geometry.adb:565-566) */
  const int32_t* idx;     /* 3*ntris */
  const int32_t* matid;   /* ntris (ignored in REFERENCE_BF mode) */
  float bbmin[3], bbmax[3];   /* used by REFERENCE_BF (geometry.adb:273) */
} ArtMesh;

/* One instance of a mesh (round 5; the reference: gcore_instance_meshes, embree_connect.cpp:147-184 -- RTC_FORMAT_FLOAT3X4_ROW_MAJOR read from
 * the first 12 floats of a 16-float block, :169).  m: object -> world. */
typedef struct ArtInstance { int32_t mesh; float m[12]; } ArtInstance;

typedef struct ArtSceneDesc {
  int32_t n_spheres;   const ArtSphere*   spheres;
  int32_t has_cornell;                              /* scene.ads:75-80 */
  float   cb_min[3], cb_max[3];
  int32_t cb_mat[6];
  float   cb_nrm[6][3];
  int32_t n_lights;    const ArtLight*    lights;   /* the reference has exactly one (scene.adb:45-48) */
  int32_t n_materials; const ArtMaterial* materials;
  int32_t n_meshes;    const ArtMesh*     meshes;   /* at most one per mode -- unless n_instances > 0 */
  float   cam_pos[3];                               /* scene.ads:27-32 */
  float   cam_matrix[16];                           /* row-major float4x4 */
  /* Instanced scenes (n_instances > 0): meshes[] are then object-space prototypes, any number of them, every one ART_MESH_CLOSEST, and
   * the scene's mesh geometry is instances[]: instance i shows mesh instances[i].mesh under its 3x4.  The render loop walks a tree over
   * the instances and one tree per mesh (memory O(meshes + instances)); the picture is that of the FLATTENED scene, bit for bit --
   * corners transformed by m (m[0] x + m[1] y + m[2] z + m[3], evaluated left to right in binary32), vertex normals by the inverse
   * transpose of its 3x3 and normalised, triangles in the order (instance, triangle of the mesh).  Cooperative trace kernel only. */
  int32_t n_instances; const ArtInstance* instances;
} ArtSceneDesc;

/* ---- render control --------------------------------------------------------------------------- */

enum { ART_RT_DEBUG = 0, ART_RT_WHITTED = 1, ART_PT_STUPID = 2, ART_PT_SHADOW = 3, ART_PT_MIS = 4 };  /* ray_tracer.ads:40 */
enum { ART_LAYOUT_ADA_XY = 0,   /* AccumBuff(x,y) / ScreenBufferData(x,y): element (x,y) at x*height + y (ray_tracer.ads:35,54) */
       ART_LAYOUT_ROW_MAJOR = 1 /* element (x,y) at y*width + x (Bitmap.Image.data, test.adb:65) */ };

/* the mutable package variables Render_Pass reads (ray_tracer.ads:20-32), passed per call */
typedef struct ArtPassParams {
  int32_t  render_type;     /* g_rend_type */
  int32_t  aa_on;           /* Anti_Aliasing_On */
  int32_t  max_depth;       /* Max_Trace_Depth, 1..16 */
  int32_t  vthreads;        /* Threads_Num: the pass adds vthreads * (aa_on ? 4 : 1) samples per pixel */
  float    background[3];   /* Background_Color */
  uint64_t seed;            /* keys the counter-based RNG (replaces Float_Random.Reset, ray_tracer.adb:147) */
  int32_t  layout;          /* layout of the host buffers handed to this call */
} ArtPassParams;

typedef struct ArtStats {
  uint64_t rays;            /* closest-hit queries issued (camera + bounce + shadow), cumulative since art_resize */
  uint64_t samples;         /* camera samples, cumulative */
  double   trace_ms;        /* GPU time inside the trace kernel, cumulative (HIP events) */
  double   pass_ms;         /* GPU time of whole passes (all kernels), cumulative */
  uint64_t trace_launches;
  uint64_t box_tests, tri_tests, node_visits, leaf_visits, traced_rays;  /* only filled by art_trace_rays(stats) / option count_tests */
  uint64_t node_phase_iters, leaf_phase_iters, wave_iters;               /* cooperative kernel: wave-level loop counters (count_tests) */
  uint64_t lost_paths;      /* self-check of the compacted work sets: paths that needed an item and had none; must stay 0 */
} ArtStats;

typedef struct ArtHit {          /* geometry.ads:57-67 flattened */
  float   t;
  int32_t is_hit;
  int32_t prim_type;             /* Primitive'Pos: 0 plane, 1 sphere, 2 triangle, 3 quad; -1 miss */
  int32_t prim_index;
  int32_t mat_id;
  int32_t mat;
  float   normal[3];
  float   u, v;                  /* triangle barycentrics (weight of C, weight of B; geometry.adb:245-246) */
} ArtHit;

typedef struct ArtBvhInfo { int32_t n_nodes, n_tris, max_stack, node_width; double build_ms; } ArtBvhInfo;   /* nodes: 8*node_width floats each */

int  art_init(int device_ordinal);                       /* -1: keep the current HIP device */
/* One process, n GPUs of the node (the Ada host calls Render_Pass from one task: ray_tracer.adb:240-293, test.adb:50).  ordinals ==
 * NULL: devices 0..n-1.  Scene and BVH are replicated, device k owns the 32x32 pixel tiles (bx, by) with (bx + skew by) mod n == k
 * (skew = 3, or 5 when 3 divides n, or 7 when 15 divides n: tiles dealt along diagonals, csrc/art_host_scene.cpp build_pixmap -- an
 * integrator that needs the map should not recompute it from this sentence but take the rule from there), and the float3
 * framebuffers are added into device 0 by ONE RCCL reduce over xGMI whenever the image is asked for (art_render_pass with host
 * pointers, art_download, art_reduce).  The image is bit-identical for any n.  Call INSTEAD of art_init; art_set_stream /
 * art_set_shard / art_bind_accum are single-device calls and fail afterwards.  Repeating one ordinal n times rehearses the whole
 * path on a single GPU (the reduce is then a local sum). */
int  art_init_devices(int32_t n, const int32_t* ordinals);
int32_t art_device_count(void);
int  art_reduce(void);                                   /* enqueue the framebuffer reduce now (no-op with one device) */
/* What the multi-device path really did (round 5: a bench line has to show how many ranks RCCL saw, not be taken on trust).
 * Filled after art_synchronize: rccl_ranks = ncclCommCount of the communicator the reduces ran on (0: no communicator -- one device
 * without ART_FORCE_RCCL, or n contexts on one GPU, whose sum is a chain of local adds); reduce_ms = GPU time of the reduces on device
 * 0's stream (HIP events around the grouped ncclReduce / the adds; it INCLUDES the time device 0's stream waits in the collective for
 * the slowest device), cumulative since art_resize; device_pass_ms[k] = GPU time of device k's render passes (HIP events on its stream),
 * cumulative.  Round 6, n > 1 devices only -- the host clock at which every device's stream reached the start and the end of each pass
 * (hipLaunchHostFunc: one clock for all devices): device_busy_ms[k] = end - start, device_idle_ms[k] = the slowest device's end - device
 * k's end (what an uneven tile deal costs), device_start_skew_ms[k] = device k's start - the first device's start (what enqueueing
 * device after device costs), all summed over the passes; passes_overlapped counts the passes in which every device had started before
 * any had finished -- Render_Pass releases all its workers before it waits for one, ray_tracer.adb:271-277. */
typedef struct ArtReduceInfo {
  int32_t devices;               /* contexts of this process (art_init_devices n) */
  int32_t rccl_ranks;            /* ranks of the RCCL communicator, 0 = none */
  int32_t path;                  /* 0 nothing to reduce, 1 grouped ncclReduce, 2 local adds (contexts on one GPU) */
  int32_t reduces;               /* reduces enqueued since art_resize */
  double  reduce_ms;             /* cumulative */
  double  device_pass_ms[8];     /* per device, cumulative */
  double  device_busy_ms[8], device_idle_ms[8], device_start_skew_ms[8];
  int32_t passes, passes_overlapped;
} ArtReduceInfo;
int  art_get_reduce_info(ArtReduceInfo* out);
int  art_set_stream(void* hip_stream);                   /* hipStream_t; NULL = default stream */
int  art_upload_scene(const ArtSceneDesc* scene);        /* Scene.Init: flatten + BVH build + copy to HBM */
int  art_resize(int32_t width, int32_t height);          /* Resize_Viewport (ray_tracer.adb:297-320): zero accum, spp := 0 */
int  art_set_shard(int32_t rank, int32_t nranks, int32_t tile);   /* pixel tiles (tile x tile) dealt along diagonals over the ranks */

/* One Render_Pass.  accum_host (float3 per pixel) and screen_host (u32 per pixel) may be NULL; when given
 * they receive the cumulative accum buffer and the resolved LDR image in p->layout.  *spp_inout is g_spp. */
int  art_render_pass(const ArtPassParams* p, float* accum_host, uint32_t* screen_host, int32_t* spp_inout);

/* Debug_Ray_Tracing + resolve (ray_tracer.adb:208-261).  All pointers optional. */
int  art_debug_hit_pass(const ArtPassParams* p, float* accum_host, uint32_t* screen_host,
                        int32_t* prim_index, int32_t* mat_id, int32_t* prim_type);

/* device-resident variants for the multi-GPU harness: accumulate into caller-owned HBM (row-major float3),
 * e.g. a buffer that is then reduced over xGMI with RCCL. */
int  art_bind_accum(void* device_accum_rowmajor);        /* NULL: back to the internal buffer */
void* art_accum_device(void);
int  art_download(float* accum_host, uint32_t* screen_host, int32_t layout, int32_t spp);
int  art_synchronize(void);

/* Scene.Find_Closest_Hit for a list of rays (host arrays of 3*n floats; tfar may be NULL = unbounded).
 * kernel: 0 = cooperative kernel (bvh_width lanes per ray: 4 by default, 8 as an option), 1 = one-ray-per-lane kernel.  stats may be NULL. */
int  art_trace_rays(const float* origins, const float* dirs, const float* tfar, int64_t n,
                    ArtHit* out, int32_t kernel, ArtStats* stats);

int  art_export_bvh(float* nodes, int64_t node_floats_cap, float* tris, int64_t tri_floats_cap, ArtBvhInfo* info);
int  art_get_stats(ArtStats* out);
/* The wavefront stages around the trace kernel (device 0, cumulative since art_resize; cooperative schedule): GPU time per kind of
 * kernel (HIP events on the launch stream, like ArtStats::trace_ms) and the work items every bounce read and kept -- what bench.py's
 * `stages` object and its whole-job roofline are made of.  items_in[b] / items_out[b]: input items of bounce b (b = 0: the camera paths)
 * and the items it wrote for bounce b + 1. */
typedef struct ArtStageStats {
  double   shade_ms, raygen_ms, fold_ms;    /* k_shade_compact (all instantiations) | k_raygen | k_resolve_last + k_fold_level + k_accumulate */
  uint64_t shade_launches, batches;
  uint64_t items_in[16], items_out[16];
} ArtStageStats;
int  art_get_stage_stats(ArtStageStats* out);
/* Tuning / test options (defaults in brackets):  "trace_kernel" [0] 0 cooperative, 1 one ray per lane;  "batch_paths" [128M];
 * "blocks_per_cu" [occupancy];  "count_tests" [0];  "node_min" [0 = 4, instanced scenes 2];  "refill_min" [2];  "ray_chunk" [48];  "queue_segments" [8];  "shadow_anyhit" [1];
 * "lds_stack_cap" [0 = automatic];  BVH build (take effect at the next art_upload_scene): "bvh_width" [4] lanes per ray = children
 * per node, 4 or 8;  "bvh_builder" [3] 3 binned SAH on the GPU (the tree of 0, built in milliseconds), 0 binned SAH on the host, 1 LBVH on the GPU, 2 PLOC on the GPU;  "bvh_ploc_radius" [8];  "bvh_spatial_splits" [0];  "bvh_max_leaf" [width];
 * "bvh_leaf_base_milli", "bvh_node_cost_milli", "bvh_tri_cost_milli".  The wavefront stages: "shade_per" [0 = measured; 2 | 4 items per thread],
 * "shade_split" [0], "skip_null_shadow" [0] (1: a shadow ray whose explicit colour is exactly zero under either verdict -- the light sample behind the
 * surface, a BxDF that is zero there -- is not traced: the same picture 7-10 % sooner, but fewer rays than the reference issues, integrators.adb:270).  Instanced scenes: "inst_coop" [1] the cooperative kernel crosses the instance boundary (0: one ray per lane, the cross-check);
 * "inst_open" [0] entry points per instance the instance tree ends at (1 whole instances, n > 1 about n subtrees of the mesh's tree per
 * instance, 0 chosen from how much the instances' boxes overlap; takes effect at the next art_upload_scene).  How the path state is mapped
 * (round 6, profiles/r6_bimodal: the shade stage's rate depends on the size of the pieces its 35-74 GB are mapped in; the picture never does):
 * "paths_spread" [-1] chunk size in MB -- the path state as one address range over separately created physical chunks (HIP virtual memory
 * management; falls back to hipMalloc); -1: 64 MB chunks for a path state of 1 GB or more, 0: plain hipMalloc (13-17 % slower stages in
 * about half of the processes); "paths_spread_holes" [0] 1: spacer chunks between the chunks, released after mapping; "paths_contiguous"
 * [0] 1: physically contiguous memory (the slowest and the one deterministic placement: for A/B work on the stage); "hot_pad" [0] items
 * between the fields of a bank's block (a multiple of 64; moves nothing).
 * Test options: "inject_lost" (the next pass counts one lost path: art_synchronize must fail), "spread_fail_at" (creating that chunk of the
 * path state fails: everything created so far is undone and the path state comes from hipMalloc), "lds_stack_cap". */
int  art_set_option(const char* name, int64_t value);
const char* art_last_error(void);
void art_shutdown(void);

/* ---- legacy geometry-core seam: embree_connect.cpp:51-244 / scene_hydra_embree.adb:37-66 -------- */

typedef struct HitCpp {          /* embree_connect.cpp:186-194, 36 bytes */
  int32_t primIndex;
  int32_t geomIndex;
  int32_t instIndex;
  float   t;
  float   normal[3];             /* unnormalised geometric normal Ng = cross(B-A, C-A) */
  float   texCoord[2];           /* barycentrics u, v */
} HitCpp;

void gcore_init_and_clear(void);                                                       /* :60-67  */
void gcore_destroy(void);                                                              /* :51-58  */
int  gcore_add_mesh_3f(const float* a_vertices3f, int a_vertexNum, const int* a_indices, int a_indicesNum);  /* :69-144: returns mesh id, copies the data */
void gcore_instance_meshes(int a_geomId, const float* a_matrices16f, int a_matrixNum); /* :147-184: 3x4 row-major taken from each 16-float block */
void gcore_commit_scene(void);                                                         /* :241-244: BVH build + upload */
#ifdef __cplusplus
bool gcore_closest_hit(const float a_rayPos[3], const float a_rayDir[3], float t_near, float t_far, HitCpp* pHit);  /* :196-238 */
#else
_Bool gcore_closest_hit(const float a_rayPos[3], const float a_rayDir[3], float t_near, float t_far, HitCpp* pHit);
#endif

/* Scene representation chosen by gcore_commit_scene: -1 automatic (one tree per mesh + a tree over the instances from 16 instances on,
 * like Embree's instance geometries, embree_connect.cpp:147-184; below that every instance is flattened into one world-space mesh),
 * 0 always flatten, 1 always two-level. */
void gcore_set_two_level(int mode);
/* Where a single-ray gcore_closest_hit is answered: 0 (default, round 4) on the calling thread -- a host walk of the committed tree, the
 * same tree, walk and triangle arithmetic as the GPU kernels, so the same hit bit for bit: what the reference's call pattern needs (28
 * tasks, one ray each: scene_hydra_embree.adb:426-446; Embree's rtcIntersect1 also runs on the caller's core, embree_connect.cpp:218);
 * 1: concurrent callers are combined into shared GPU launches (rounds 2-3; kept for A/B and for the host == GPU parity test). */
void gcore_set_single_ray_on_gpu(int on);
/* Batch form (extension; no counterpart in embree_connect.cpp): t_near / t_far may be NULL for 0 / 1e5; returns the number of hits. */
int  gcore_closest_hit_n(int a_rayNum, const float* a_rayPos3f, const float* a_rayDir3f, const float* t_near, const float* t_far,
                         HitCpp* pHits, unsigned char* pFound);

#ifdef __cplusplus
}
#endif
#endif

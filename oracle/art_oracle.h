/*
 * art_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference path tracer's per-pixel render loop
 * (FROL256/ada-ray-tracer: Ray_Tracer.Render_Pass -> Integrator.DoPass ->
 * PathTrace -> Scene.Find_Closest_Hit and everything below it).  Every
 * function in art_oracle.c cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (ada-ray-tracer_amd/) never includes,
 * links or calls anything in this directory.
 *
 * PARITY PIN STATUS (see DESIGN.md "Oracle"):
 *   - The reference is Ada 2012; no Ada compiler exists in the build image and
 *     the reference has no tests / golden vectors, so bit-level parity with the
 *     Ada binary is UNPINNED for the two GNAT-runtime dependencies that are not
 *     in the reference tree: Ada.Numerics.Float_Random (MT19937 stream) and
 *     Ada.Numerics.Generic_Elementary_Functions (sin/cos/tan/"**").
 *     This restatement replaces them with (a) a counter-based Philox4x32-10
 *     generator keyed by (seed, pixel, sample, bounce) and (b) the "ART-M1"
 *     transcendental functions defined below (double-precision kernels,
 *     rounded once to float).
 *   - What IS pinned: data/pyramid2.vsgf decode (tests/golden), the hard-coded
 *     scene constants of scene.adb; the reference's own output picture image.png
 *     AS A WHOLE -- it is the HEAD scene seen from (0, 2.55, 11): the oracle's
 *     render from there matches it block by block (0.33 LDR levels mean on
 *     16x16-pixel block means) and in 18 named regions (glass, Phong, caustic,
 *     pyramid, light, walls; tests/picture_pin.py, tests/test_oracle_image_pin.py);
 *     and the sampling code (lights.adb, materials.adb, vector_math.adb helpers)
 *     bit for bit against an independent numpy-float32 transcription of the Ada
 *     text (tests/ada_transcription.py, tests/test_sampling_kat.py).
 */
#ifndef ART_ORACLE_H
#define ART_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- flattened scene (mirrors scene.ads:61-80, geometry.ads:15-111) ---- */

enum { ORC_MAT_NULL = 0, ORC_MAT_LIGHT = 1, ORC_MAT_LAMBERT = 2,
       ORC_MAT_MIRROR = 3, ORC_MAT_GLASS = 4, ORC_MAT_PHONG = 5 };

/* p[] meaning by type (materials.ads:58-130):
 *  LIGHT   : light = index of the light (lref)
 *  LAMBERT : p[0..2] = kd
 *  MIRROR  : p[0..2] = reflection
 *  GLASS   : p[0..2] = reflection, p[3..5] = transparency, p[6] = ior
 *  PHONG   : p[0..2] = reflection, p[3] = cosPower                         */
typedef struct { int32_t type; int32_t light; float p[8]; } orc_material;

enum { ORC_LIGHT_RECT = 0, ORC_LIGHT_SPHERE = 1 };

/* lights.ads:36-55.  mat = material-table index of the MaterialLight that
 * refers back to this light (used for the rect light's geometry hit).       */
typedef struct {
  int32_t shape;
  int32_t mat;
  float boxMin[3], boxMax[3], normal[3];   /* rect   */
  float center[3], radius;                 /* sphere */
  float intensity[3];
  float surfaceArea;
} orc_light;

typedef struct { float pos[3]; float r; int32_t mat; } orc_sphere;  /* geometry.ads:21-25 */

enum { ORC_MESH_REFERENCE_BF = 0,   /* geometry.adb:266-323 verbatim (first-hit window quirk, matId := 2) */
       ORC_MESH_CLOSEST      = 1 }; /* true closest hit (Embree semantics, embree_connect.cpp:196-238) with
                                       the reference's Moeller-Trumbore arithmetic; per-triangle material ids */

typedef struct {
  int32_t mode;
  int32_t nverts, ntris;
  const float*   pos;     /* 3*nverts, already transformed (geometry.adb:593-607) */
  const float*   nrm;     /* 3*nverts, NOT transformed (geometry.adb:605)          */
  const float*   uv;      /* 2*nverts (zeros from the VSGF loader, geometry.adb:565-566) */
  const int32_t* idx;     /* 3*ntris  */
  const int32_t* matid;   /* ntris    */
  float bbmin[3], bbmax[3];
  /* optional (ORC_MESH_CLOSEST only): the product's exported BVH8 (orc_bvh_walk layout).  When set, the closest-hit
   * search walks it instead of scanning all triangles -- same result (tests), used for the timed CPU baseline. */
  const float* bvh_nodes;
  const float* bvh_tris;
  int32_t bvh_width;      /* children per node of that tree: 8 or 4 (0 = 8) */
} orc_mesh;

typedef struct {
  int32_t n_spheres;  const orc_sphere*   spheres;
  int32_t has_cornell;                        /* scene.ads:75-80 */
  float   cb_min[3], cb_max[3];
  int32_t cb_mat[6];
  float   cb_nrm[6][3];
  int32_t n_lights;   const orc_light*    lights;
  int32_t n_materials; const orc_material* materials;
  int32_t n_meshes;   const orc_mesh*     meshes;   /* at most one per mode */
  float   cam_pos[3];
  float   cam_matrix[16];                      /* row-major float4x4 (generic_vector_math.ads:64) */
} orc_scene;

enum { ORC_RT_DEBUG = 0, ORC_RT_WHITTED = 1, ORC_PT_STUPID = 2, ORC_PT_SHADOW = 3, ORC_PT_MIS = 4 }; /* ray_tracer.ads:40 */

typedef struct {
  int32_t width, height;
  int32_t render_type;
  int32_t aa_on;          /* Anti_Aliasing_On  ray_tracer.ads:24 */
  int32_t max_depth;      /* Max_Trace_Depth   ray_tracer.ads:25 */
  int32_t vthreads;       /* Threads_Num       ray_tracer.ads:23: samples per pass = vthreads * (aa?4:1) */
  float   background[3];  /* ray_tracer.ads:27 */
  uint64_t seed;
  int32_t  nthreads;      /* host worker threads (OpenMP); 0 = all */
} orc_params;

typedef struct {
  uint64_t rays;          /* calls to Find_Closest_Hit (camera + bounce + shadow) */
  uint64_t samples;       /* camera samples */
  uint64_t tri_tests;     /* IntersectTriangle calls */
} orc_counters;

/* ---- ART-M1 math + RNG (exported for KATs) ---- */
float    orc_sinf(float x);
float    orc_cosf(float x);
float    orc_tanf(float x);
float    orc_powf(float x, float y);          /* restates Ada "**" + vector_math.adb:24-47 */
void     orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
float    orc_rng_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t slot);

/* ---- per-function KAT entry points: lights.adb Sample / EvalPDF, materials.adb SampleAndEvalBxDF / EvalBxDF / EvalPDF ---- */
void  orc_kat_light_sample(const orc_light* l, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, const float p[3], float out10[10]);
float orc_kat_light_eval_pdf(const orc_light* l, const float p[3], const float ray_dir[3], float hit_dist);
void  orc_kat_mat_sample(const orc_material* m, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, const float ray_dir[3],
                         const float normal[3], float out8[8]);
void  orc_kat_mat_eval(const orc_material* m, const float l[3], const float v[3], const float n[3], float out4[4]);

/* ---- scene construction helpers ---- */
/* scene.adb:89-217.  Fills caller-provided storage; mesh arrays must come from orc_load_vsgf. */
typedef struct {
  orc_scene    scene;
  orc_sphere   spheres[3];
  orc_light    lights[1];
  orc_material materials[11];
  orc_mesh     meshes[1];
} orc_cornell_storage;

/* geometry.adb:499-609.  Returns 0 on success.  Arrays are malloc'ed; free with orc_free_mesh. */
int  orc_load_vsgf(const char* path, const float transform16[16], orc_mesh* out);
int  orc_load_vsgf_mem(const void* data, int64_t nbytes, const float transform16[16], orc_mesh* out);
void orc_free_mesh(orc_mesh* m);
void orc_cornell_mesh_transform(float out16[16]);                  /* scene.adb:194-206 */
void orc_build_cornell(orc_cornell_storage* st, const orc_mesh* pyramid, int use_rect_light);

/* ---- the hot path ---- */
/* One Render_Pass (ray_tracer.adb:240-293) minus the LDR resolve.  accum is float3[height][width]
 * ROW-MAJOR (y*width+x), cumulative across passes; *spp is advanced like g_spp.
 * sample index of (vthread t, aa i) = *spp_before + t*(aa?4:1) + i.                              */
void orc_render_pass(const orc_scene* scn, const orc_params* prm, float* accum, int32_t* spp, orc_counters* cnt);
/* the same pass for a list of pixels: accum[3k..3k+2] = colBuff(xs[k], ys[k]) (cumulative), samples spp0 .. spp0 + vthreads*per - 1 */
void orc_render_pixels(const orc_scene* scn, const orc_params* prm, const int32_t* xs, const int32_t* ys, int64_t n, float* accum, int32_t spp0,
                       orc_counters* cnt);

/* Radiance of one camera sample (for debugging mismatches). */
/* same result, organised as the reference's task pool: vthreads tasks, each a whole-frame DoPass into a private frame (returns 0 on success) */
int  orc_render_pass_tasks(const orc_scene* scn, const orc_params* prm, float* accum, int32_t* spp, orc_counters* cnt);
void orc_sample_radiance(const orc_scene* scn, const orc_params* prm, int32_t x, int32_t y,
                         uint32_t sample_index, float out_rgb[3]);

/* Debug_Ray_Tracing (ray_tracer.adb:208-238): writes palette colour into accum, plus the raw ids. */
void orc_debug_pass(const orc_scene* scn, const orc_params* prm, float* accum,
                    int32_t* prim_index, int32_t* mat_id, int32_t* prim_type);

/* Resolve (ray_tracer.adb:281-291 + 19-57): screen = pack(tonemap(gamma(accum/spp))). row-major. */
void orc_resolve(const float* accum, int32_t width, int32_t height, int32_t spp, uint32_t* screen);

/* Closest hit for a list of rays (a8).  out: t, prim_type(0 plane,1 sphere,2 triangle,3 quad,-1 miss),
 * prim_index, matId, mat (resolved table index), normal, tx/ty = (u,v) barycentrics for triangles. */
typedef struct { float t; int32_t is_hit; int32_t prim_type; int32_t prim_index; int32_t mat_id; int32_t mat;
                 float normal[3]; float tx, ty; } orc_hit;
void orc_closest_hits(const orc_scene* scn, const float* origins, const float* dirs, int64_t n, orc_hit* out);

/* Bitmap.SaveBMP (bitmap.adb:31-85).  image = u32[height*width] row-major (test.adb:63-67). */
int  orc_save_bmp(const char* path, const uint32_t* image, int32_t width, int32_t height);
int64_t orc_bmp_bytes(const uint32_t* image, int32_t width, int32_t height, uint8_t* out, int64_t cap);

/* ---- BVH walk with counters: mirrors the product's published traversal order on the
 * product's exported BVH8 (include/art_hip.h: art_export_bvh).  Counts valid-child slab tests (B) and
 * triangle tests (T) for a list of rays, and returns the closest triangle hit (t, prim).           */
typedef struct { uint64_t rays, box_tests, tri_tests, node_visits, leaf_visits; } orc_bvh_counters;
void orc_bvh_walk(const float* nodes /* 64 floats per node */, int32_t n_nodes,
                  const float* tris /* 12 floats per tri  */, int32_t n_tris,
                  const float* origins, const float* dirs, const float* tfar, int64_t n,
                  float* out_t, int32_t* out_prim, orc_bvh_counters* cnt);
/* same for a tree of `width` children per node (8 * width floats per node: width x {lo.xyz, ref}, then width x {hi.xyz, count}) */
void orc_bvh_walk_w(const float* nodes, int32_t n_nodes, const float* tris, int32_t n_tris, int32_t width,
                    const float* origins, const float* dirs, const float* tfar, int64_t n,
                    float* out_t, int32_t* out_prim, orc_bvh_counters* cnt);

#ifdef __cplusplus
}
#endif
#endif

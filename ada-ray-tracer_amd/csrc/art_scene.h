// art_scene.h -- the flattened scene as the kernels see it (SoA / packed arrays resident in HBM),
// plus the wavefront path-state arrays.  Mirrors scene.ads:61-80 / geometry.ads:15-111 /
// materials.ads:58-130 / lights.ads:36-55 after flattening the tagged-type hierarchies into tables.
#pragma once
#include "art_math.h"

namespace art {

enum MatType : int32_t { MAT_NULL = 0, MAT_LIGHT = 1, MAT_LAMBERT = 2, MAT_MIRROR = 3, MAT_GLASS = 4, MAT_PHONG = 5 };
enum LightShape : int32_t { LIGHT_RECT = 0, LIGHT_SPHERE = 1 };
enum RenderType : int32_t { RT_DEBUG = 0, RT_WHITTED = 1, PT_STUPID = 2, PT_SHADOW = 3, PT_MIS = 4 };  // ray_tracer.ads:40

// 40-byte material record: type tag + light back-reference + up to 8 parameters
//  LAMBERT p0-2 kd | MIRROR p0-2 reflection | GLASS p0-2 reflection p3-5 transparency p6 ior | PHONG p0-2 reflection p3 cosPower
struct DevMaterial { int32_t type; int32_t light; float p[8]; };

struct DevLight {
  int32_t shape; int32_t mat;
  float boxMin[3], boxMax[3], normal[3];
  float center[3], radius;
  float intensity[3];
  float surfaceArea;
};

struct DevSphere { float x, y, z, r; };

// closest hit of a ray, one 16-byte record: every producer writes it and every consumer reads it with ONE 16-byte access
struct alignas(16) DevHit { float t; uint32_t key; float u, v; };

// one 16-byte quarter of a 64-byte trace record (art_kernels.h): what the cooperative trace kernel starts a ray from
struct alignas(16) Rec4 { float x, y, z, w; };
// Which rays a work item of a bank owns a trace record for (round 3: the wavefront stages write the records themselves, at positions
// given by the item index -- no queue, no atomics, no separate pass over the rays):
//   REC_NONE    the plain layout: rays as SoA arrays, k_analytic prepares the queue (one-ray-per-lane schedule, host simulation, debug pass)
//   REC_EXT     one extension ray per item, record w            (after raygen; PT_STUPID)
//   REC_BOTH    k_shade_compact: per wave and round, the extension rays of its nk kept items as one block of nk records, then their nk
//               shadow rays (the two rays of a surface point within 64 records of each other in the queue); the unstaged path of
//               the host simulation: extension ray at record 2w, shadow ray at 2w + 1 (rec_slot)
//   REC_SHADOW  one shadow ray per item, record w               (after the last bounce: the path itself has ended)
enum RecMode : int32_t { REC_NONE = 0, REC_EXT = 1, REC_BOTH = 2, REC_SHADOW = 3 };

// Hit key: (class << 28) | index.  Class order == candidate order of Scene.Find_Closest_Hit
// (scene.adb:62-78: spheres, Cornell box, flat light, mesh) so that the reference's strict-'<' merge
// is the lexicographic minimum over (t, key).
constexpr uint32_t KEY_SPHERE = 0u << 28, KEY_CORNELL = 1u << 28, KEY_QUAD = 2u << 28, KEY_BFTRI = 3u << 28, KEY_TRI = 4u << 28;
constexpr uint32_t KEY_MISS = 0x7fffffffu;
constexpr uint32_t KEY_INDEX_MASK = (1u << 28) - 1u;

// BVH node of width W (8 or 4 children), 32*W bytes = 8*W floats, aligned to its size:
//   half A (16*W B): child j -> { lo.x, lo.y, lo.z, ref }      j = 0..W-1
//   half B (16*W B): child j -> { hi.x, hi.y, hi.z, count }
//   ref = -1: empty slot.  count == 0: inner child, ref = node index.  count 1..W: leaf, ref = first
//   triangle record.  Lane j of a W-lane ray group loads exactly its 2 x 16 B; a group reads 2 x 16*W contiguous bytes.
// Triangle record, 48 bytes = 12 floats: A.xyz B.xyz C.xyz prim(int) pad pad   (prim = index in the caller's mesh)
constexpr int kNodeFloats = 64;      // W = 8; a node of width W has node_floats(W) floats
constexpr int node_floats(int width) { return 8 * width; }
constexpr int kTriFloats = 12;
constexpr int kTriShadeFloats = 16;  // nA.xyz nB.xyz nC.xyz matId, padded to 64 bytes
constexpr int kMaxLeafTris = 8;
constexpr int kStackEntries = 160;  // private stack of the one-ray-per-lane traversal; art_upload_scene rejects deeper trees

// One instance of a mesh (round 5: art_upload_scene takes meshes + 3x4 instance transforms, embree_connect.cpp:147-184, and the render loop
// walks them as a two-level tree without flattening).  key index of a hit = instance << DevScene::inst_shift | triangle of the mesh.
//
// ENTRY POINTS.  The instance tree does not have to end at whole instances: an instance may be "opened" at build time (art_instanced_build.cpp),
// its mesh's tree entered at several subtrees instead of at the root, each with the tight world box of ITS triangles -- a torus or a
// rotated plate overlaps far fewer neighbours that way, and every overlap is a mesh tree entered for nothing.  The table therefore holds
// DevScene::n_entry >= n_inst records: record i < n_inst is instance i (and its first entry point), the records behind are further
// entry points (copies of their instance's record but for the two root words).  The searches index the table by ENTRY (what an instance
// tree leaf names), shading by INSTANCE (what a hit's key names: `inst`).
struct DevInstance {
  float m[12];                 // object -> world, 3x4 row-major (the first 12 floats of the reference's 16-float block, embree_connect.cpp:169)
  float minv[12];              // world -> object
  int32_t node_base;           // first node of the mesh's tree in DevScene::blas_nodes (in nodes)
  int32_t tri_base;            // first triangle record of the mesh in DevScene::blas_tris (object space, one winding, prim = triangle of the mesh)
  int32_t shade_base;          // first shading record of the mesh in DevScene::m_shade (object-space vertex normals + material id)
  int32_t root_entry;          // where this entry point enters the mesh's tree: (node << 4) or (first record << 4) | count, relative to node_base / tri_base (0: the root)
  uint32_t qroot;              // the same as an entry word of the cooperative kernel's quantised node array (absolute: node byte offset, or leaf flag | record byte offset | count)
  int32_t inst;                // the instance this entry point belongs to (= its own index for the first n_inst records)
  int32_t pad_[2];
};
static_assert(sizeof(DevInstance) == 128, "k_trace_coop reads the instance table by byte offsets");

struct DevScene {
  int32_t n_spheres; const DevSphere* spheres; const int32_t* sphere_mat;
  int32_t has_cornell;
  float cb_min[3], cb_max[3];
  int32_t cb_mat[6];
  float cb_nrm[6][3];
  int32_t n_lights; const DevLight* lights;
  int32_t n_materials; const DevMaterial* materials;
  // reference-semantics brute-force mesh (geometry.adb:266-323)
  int32_t bf_ntris;
  const float* bf_pos; const float* bf_nrm; const float* bf_uv; const int32_t* bf_idx;
  float bf_bbmin[3], bf_bbmax[3];
  // closest-hit mesh behind the BVH
  int32_t n_tris; int32_t n_nodes; int32_t node_width;   // node_width: 8 or 4 children per node
  const float* nodes; const float* tris;
  // shading data of the closest-hit mesh, one 64-byte record per triangle (by prim index): the three vertex normals and the material id --
  // ONE line fetch per shaded hit instead of five scattered ones (index triple, three normals, material id)
  const float* m_shade;
  // instanced closest-hit meshes (n_inst > 0; then nodes / tris above are unused and n_tris is the instances' total): a 4-wide tree over
  // the instances' world boxes (a leaf holds ONE proxy record whose prim is the instance), one tree per mesh in object space
  int32_t n_inst, inst_shift, n_entry;
  const DevInstance* inst;     // n_entry records (see DevInstance)
  const float* tlas_nodes; const float* tlas_tris; const float* blas_nodes; const float* blas_tris;
  // camera (scene.ads:27-32)
  float cam_pos[3];
  float cam_matrix[16];
};

struct DevFrame {
  int32_t width, height;
  int32_t render_type, aa_on, max_depth;
  uint32_t seed_lo, seed_hi;
  float background[3];
  float cam_z;          // -float(width)/safe_tan(fov/2)   ray_tracer.adb:67 (computed once on the host)
  // Option skip_null_shadow (default 0 = the reference's work: Compute_Shadow for every surface hit, integrators.adb:270).  1: a shadow ray
  // whose explicit colour is exactly zero whatever its verdict -- the light sample lies behind the surface (cosTheta1 = 0) or the BxDF is
  // zero there -- is not traced.  The picture is the same bits (a zero that is not added is a zero); the RAY COUNT is not the reference's.
  int32_t skip_null_shadow;
};

// Record schedule (round 5): a bank's per-item arrays are ONE block, field f of item w at hot[f * stride + w] (stride: P rounded up to 64
// items), and the fold arrays ONE block `cold`.  k_shade_compact addresses everything from these two bases with scalar arithmetic: held as
// ~50 separate pointers (the fields of DevPaths, which still name the same memory for the other kernels) they cost it 80 spilled SGPRs
// and a scalar load + s_waitcnt in front of almost every access.
enum HotField : int32_t { HF_OX = 0, HF_OY = 1, HF_OZ = 2, HF_DX = 3, HF_DY = 4, HF_DZ = 5, HF_HIT = 6 /* 4 words per item: fields 6..9 */,
                          HF_SHT = 10, HF_PDF = 11, HF_FLAGS = 12, HF_SHMIN = 13, HF_SLOT = 14, kHotFields = 15 };

// Wavefront state for a batch of P path slots (slot = local_sample * npix + local_pixel).  All SoA.
struct DevPaths {
  int32_t P;                    // slots in this batch
  float* hot; int32_t stride;   // record schedule: the bank's block (HotField); nullptr: only the pointer fields below exist
  float* cold; int32_t depth;   // record schedule: e_c[level] = cold + (c (depth + 1) + level) P, w_c[level] = cold + (3 (depth + 1) + c depth + level) P, child[level] = cold + (3 (depth + 1) + 3 depth + level) P
  int32_t npix;                 // pixels owned by this device (shard)
  const uint32_t* pixmap;       // local pixel -> global pixel index (y*width + x); nullptr = identity
  uint32_t sample_base;         // global index of local sample 0
  // rays: [0,P) extension rays, [P,2P) shadow rays.  tfar < 0 marks a dead / absent ray.
  float* ray_ox; float* ray_oy; float* ray_oz;
  float* ray_dx; float* ray_dy; float* ray_dz;
  float* ray_tfar;
  // hits, same indexing
  DevHit* hit;
  // Record schedule (round 5): a SHADOW ray's result is ONE word, sh_t[item] = t of the closest hit the search ended with, or -1 (none) --
  // Compute_Shadow only asks whether that t lies above 10 eps (ray_tracer.adb:122), so the 16-byte record (key, u, v) was 12 bytes
  // written by the stage, 12 read by the next one and a 16-byte store of the trace kernel for nothing.  The record names the word by
  // kShadowWord | item (art_kernels.h TraceArgs::sh_t).  nullptr: the plain layout, shadow hits at hit[P + item].
  float* sh_t;
  // per-path
  float* prev_pdf;              // MatSample.pdf of the previous bounce
  uint32_t* flags;              // bit0 alive, bit1 prev pureSpecular, bit2 shadow pending, bits 8.. levels recorded
  float* sh_min_t;              // 10*epsilon of Compute_Shadow (ray_tracer.adb:122)
  float* cand_r; float* cand_g; float* cand_b;   // explicit colour awaiting its shadow test
  // inside-out fold stack: level k -> e_k (explicit), w_k (|cos| * bxdf)   integrators.adb:299
  float* e_r; float* e_g; float* e_b;            // [max_depth][P]
  float* w_r; float* w_g; float* w_b;
  float* term_r; float* term_g; float* term_b;   // value returned by the deepest PathTrace call
  // per-sample radiance, consumed by the accumulate kernel in the reference's summation order
  float* rad_r; float* rad_g; float* rad_b;      // [samples_in_batch][npix]
  // Compacted work sets (round 2).  The arrays above the fold stack ("hot" state: rays, hits, prev_pdf, flags, sh_min_t, cand) are
  // indexed by WORK ITEM; the fold stack, term and rad ("cold") by SLOT.  slot_id maps item -> slot (nullptr: the identity, i.e. the
  // plain one-item-per-slot layout of raygen, of the one-ray-per-lane schedule and of the host simulation).  final_flags[slot] keeps the
  // flags word a path ended with (levels recorded, bits 8..) for the fold; with the identity layout it is the flags array itself.
  const uint32_t* slot_id;
  uint32_t* final_flags;
  // Dense fold records (round 4; the compacted schedule).  fold_dense = 0: level k's e_k / w_k live at [k][slot] (the arrays above), a path's
  // end at term / final_flags[slot]; a stage then scatters 4-byte words over every line of the level.  fold_dense = 1: a level's records are
  // indexed by the ITEM index of that bounce's input set, so every store is dense.  Item w of bounce k leaves w_*[k][w] = its weight w_k, or
  // the value its path ended with, and child[k][w] (art_shade.h fold_child_word: the item's index at bounce k + 1, or "ended here", or "a
  // surface whose successor was not kept", plus one bit for the shadow test the item resolved for its predecessor); it also stores its
  // explicit light e_k -- still subject to that shadow test -- where the fold will look for it: e_*[k + 1][w'], w' its successor's index.
  // The successor only records the verdict in its own child word (k_resolve_last, after the last bounce, zeroes a shadowed e instead).
  // k_fold_level walks the levels from the deepest up.  (The pending explicit colour thus needs no hot-state words: cand_* are unused.)
  int32_t* child;               // [max_depth][P]
  int32_t fold_dense;
  // 1 (the compacted schedule): what raygen would write for every camera ray alike, or what bounce 0 can recompute from the slot, is not
  // stored at all -- flags = alive | previous-specular, previous pdf 1 (StartSample, materials.ads:25), origin = the camera position,
  // direction = camera_dir(pixel, sample) again (60 instructions against 24 B written and 24 B read per path).  The identity-layout
  // bank (slot_id == nullptr) is then raygen's and is read by bounce 0 only.
  int32_t synth0;
  // Trace records of this bank's rays (round 3), 4 x Rec4 per record, or nullptr (REC_NONE: the SoA ray arrays above are used).
  // WRITE-ONLY for the stages: both banks point at ONE record array (a bank's records are dead once its rays are traced), so a stage
  // that read its input item's record would race with the records other workgroups are writing for the output bank -- a stage reads
  // the extension ray from the six SoA words of its input bank instead.  A stage writes the records of the rays it emits -- analytic
  // primitives already intersected (the starting bound), slab set-up done -- where the trace kernel picks them up in item order; a
  // record names its hit slot itself (r3.y), so its position in the array carries no meaning beyond the order.
  Rec4* rec;
  int32_t rec_mode;             // RecMode of `rec`
  int32_t shadow_rule;          // shadow rays carry Compute_Shadow's 10*eps so that the search may use the visibility rule (option shadow_anyhit)
  int32_t has_bvh;              // the scene has a BVH mesh: records are only worth writing if a trace kernel will read them
};

constexpr uint32_t FLAG_ALIVE = 1u, FLAG_PREV_SPEC = 2u, FLAG_SHADOW_PENDING = 4u;
constexpr uint32_t kShadowWord = 0x80000000u;      // hit-slot word of a trace record: bit 31 set = the result goes to sh_t[word & ~bit 31] as one float

}  // namespace art

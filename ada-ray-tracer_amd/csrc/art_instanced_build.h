// art_instanced_build.h -- host-side construction of the two-level scene of art_instanced.h (no HIP here: the geometry-core seam
// uploads the result, the host simulation of the tests walks it on the CPU).
#pragma once
#include <string>
#include <vector>
#include "art_bvh.h"
#include "art_instanced.h"
#include "art_qnode.h"

namespace art {

struct InstMeshIn { const float* verts; size_t n_verts; const int32_t* idx; size_t n_tris; };   // object space, 3 floats / 3 indices
struct InstIn { int32_t mesh; float m[12]; };                                                    // object -> world, 3x4 row-major

struct TwoLevelHost {
  std::vector<InstRec> inst;            // valid instances (a singular matrix drops its instance)
  std::vector<int32_t> inst_src;        // inst[k] is the caller's instance inst_src[k]
  Bvh8 tlas;                            // 4-wide tree over one proxy triangle per instance (its corners span the padded world box)
  std::vector<float> blas_nodes, blas_tris;
  // one-sided (render) builds: the whole two-level tree as ONE array of 64-byte quantised nodes for the cooperative trace kernel
  // (art_qnode.h): the instance tree first (its leaves rewritten to instance markers, kQEntryInstance), then every mesh's tree with its
  // entry words made absolute (node offsets + qnode_base[mesh], triangle offsets + the mesh's tri_base).  qnode_base: first node per mesh.
  std::vector<uint32_t> qnodes; std::vector<int32_t> qnode_base;
  // what the instance tree's leaves name (art_scene.h DevInstance): entry[i], i < inst.size(), is instance i's first (or only) entry point,
  // further ones follow.  root_entry: (node << 4) | 0 or (first record << 4) | count inside the mesh's tree; qroot: the cooperative kernel's word.
  struct EntryPoint { int32_t inst; int32_t root_entry; uint32_t qroot; };
  std::vector<EntryPoint> entry;
  int32_t open_factor = 1;              // entry points per instance asked of the build (the automatic choice, if that was left to it)
  int32_t blas_max_stack = 0;           // the largest worst-case traversal stack of the meshes' trees
  std::vector<float> mesh_pad_abs;      // per mesh: the absolute pad its tree's boxes were built with (>= the caller's; follows the mesh's instances)
  InstScene view() const {
    InstScene S;
    S.tlas_nodes = tlas.nodes.data(); S.tlas_tris = tlas.tris.data(); S.blas_nodes = blas_nodes.data(); S.blas_tris = blas_tris.data();
    S.inst = inst.data(); S.n_inst = (int32_t)inst.size(); S.width = 4;
    return S;
  }
};

// one tree per mesh (object space, both windings: record 2t front, 2t+1 back), one tree over the instances' world boxes
// two_sided = false (the render loop's instanced scenes, art_host_scene.cpp): one record per triangle in the caller's winding, prim = its
// index in the mesh -- the reference's one-sided test; the meshes' boxes then carry the wider padding `pad_rel` / `pad_abs` (the walk
// tests them with the ray taken to object space in binary32, while the triangles are tested in world space: no box may cull a
// triangle that the world-space arithmetic would accept), and a singular instance matrix is an error instead of a dropped instance.
// open_factor > 1 (one-sided builds): the instance tree ends at about open_factor entry points per instance -- subtrees of the meshes'
// trees under the tight world boxes of their own triangles -- instead of at whole instances.  0: chosen from how much the instances' boxes overlap.
// scene_extent: the largest |coordinate| of anything a ray can start at besides the instances themselves (walls, spheres, the camera); the
// absolute pad of a mesh's boxes is derived from it and from the mesh's instances (their inverse matrices): TwoLevelHost::mesh_pad_abs.
bool build_two_level_host(const std::vector<InstMeshIn>& meshes, const std::vector<InstIn>& insts, TwoLevelHost& out, std::string& err,
                          bool two_sided = true, float pad_rel = -1.0f, float pad_abs = -1.0f, int open_factor = 1, float scene_extent = 0.0f);

}  // namespace art

// art_host_scene.h -- validation + flattening of an ArtSceneDesc into the packed host arrays that are
// copied to HBM (and, in the host-simulation test build, consumed directly).  Scene.Init's job.
#pragma once
#include <string>
#include <vector>
#include "../../include/art_hip.h"
#include "art_bvh.h"
#include "art_scene.h"
#include "art_instanced_build.h"

namespace art {

struct HostScene {
  std::vector<DevSphere> spheres; std::vector<int32_t> sphere_mat;
  std::vector<DevLight> lights;
  std::vector<DevMaterial> materials;
  // brute-force (reference) mesh
  std::vector<float> bf_pos, bf_nrm, bf_uv; std::vector<int32_t> bf_idx;
  int32_t bf_ntris = 0;
  // closest-hit mesh
  std::vector<float> m_shade;   // kTriShadeFloats per triangle: vertex normals of A, B, C and the material id (art_scene.h)
  std::vector<float> m_pos;   // kept for gcore-style geometric normals / export
  Bvh8 bvh;
  std::vector<float> deferred_tri9;   // builder == 1: triangle corners for the GPU build (bvh stays empty until then)
  bool gpu_built = false;             // nodes / tris live only in HBM (art_export_bvh copies them back on demand)
  double bvh_build_ms = 0.0;
  // instanced scene (ArtSceneDesc::n_instances > 0): m_shade then holds the MESHES' records (object-space normals), one block per mesh
  TwoLevelHost two; std::vector<DevInstance> inst;
  DevScene hdr;               // scalar part; pointer members are filled by the owner (host or device addresses)
};

// the explicit flattening of an instanced scene's meshes -- what an instanced render must equal bit for bit: positions, normals (3 floats per
// vertex), indices, material ids of ONE world-space mesh, triangles in the order (instance, triangle of the mesh)
bool flatten_instances(const ArtSceneDesc& d, std::vector<float>& pos, std::vector<float>& nrm, std::vector<int32_t>& idx, std::vector<int32_t>& matid, std::string& err);


bool flatten_scene(const ArtSceneDesc& d, const BvhBuildParams& bp, HostScene& out, std::string& err);

// pixel ownership for multi-GPU sharding: tile x tile pixel tiles dealt along diagonals, (bx + s by) % nranks == rank (s = 3, or 5 / 7 when 3 divides nranks).
// Returns the global pixel indices (y*W + x) owned by `rank`, in tile order.
std::vector<uint32_t> build_pixmap(int W, int H, int rank, int nranks, int tile);

// points hdr's pointer members at the host vectors (host simulation / CPU-side checks)
void bind_host_pointers(HostScene& hs);

}  // namespace art

// art_lbvh.h -- GPU BVH8 construction (see art_lbvh.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <string>
#include "art_bvh.h"

namespace art {

struct GpuBvh {
  float* nodes = nullptr;       // device, kNodeFloats per node (capacity n nodes), owned by the caller after success
  void* qnodes = nullptr;       // device, width 4: 64-byte quantised nodes (art_qnode.h), owned by the caller after success
  float* tris = nullptr;        // device, kTriFloats per triangle, Morton order
  int32_t n_nodes = 0, n_tris = 0, max_stack = 1, levels = 0;
  float build_ms = 0.0f;        // HIP events around the whole build
};

// d_tri9: device pointer, 9 floats per triangle; n >= 2.  Uses prm.max_leaf / inflate_rel / inflate_abs.
// prm.builder: 1 LBVH, 2 PLOC (art_lbvh.hip), 3 binned SAH (art_sah.hip: the host builder's tree, built breadth-first on the GPU)
bool build_bvh8_gpu(const float* d_tri9, int n, const BvhBuildParams& prm, hipStream_t stream, GpuBvh& out, std::string& err);
bool build_bvh_sah_gpu(const float* d_tri9, int n, const BvhBuildParams& prm, hipStream_t stream, GpuBvh& out, std::string& err);
// width 4: the 64-byte quantised form of every node; the binary32 packets become the dequantised tree (art_qnode.h).  qnodes: n_nodes * 64 bytes.
void launch_quantise_nodes(hipStream_t stream, float* nodes, int n_nodes, void* qnodes);

}  // namespace art

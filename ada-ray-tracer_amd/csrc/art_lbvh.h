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
bool build_bvh8_gpu(const float* d_tri9, int n, const BvhBuildParams& prm, hipStream_t stream, GpuBvh& out, std::string& err);

}  // namespace art

// calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE on THIS backend's access pattern (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern").  8-lane groups read random, 128-byte aligned 128-byte
// segments (16 B per lane, dwordx4) from a 2 GiB table (>> 256 MiB Infinity Cache, so every segment is a miss): the same
// shape as one half of a BVH8 node packet.  Known bytes = groups * iters * 128.  Run:
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_gather128(const float4* __restrict__ table, uint32_t n_segments, int iters, float* sink) {
  const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) >> 3, j = threadIdx.x & 7;
  uint32_t state = gid * 2654435761u + 12345u;
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    state = state * 1664525u + 1013904223u;
    const uint32_t seg = (state >> 4) % n_segments;
    const float4 v = table[(size_t)seg * 8 + j];
    acc += v.x + v.w;
    state ^= __float_as_uint(v.y) & 1u;     // dependent chain like a traversal
  }
  if (acc == 123.456f) sink[0] = acc;
}

int main() {
  const size_t bytes = 2ull << 30;
  float4* table; float* sink;
  hipMalloc(&table, bytes); hipMalloc(&sink, 4);
  hipMemset(table, 0, bytes);
  const uint32_t n_segments = (uint32_t)(bytes / 128);
  const int blocks = 256 * 8, threads = 256, iters = 4096;
  hipLaunchKernelGGL(k_gather128, dim3(blocks), dim3(threads), 0, 0, table, n_segments, iters, sink);
  hipDeviceSynchronize();
  const double groups = (double)blocks * threads / 8.0;
  std::printf("known_bytes %.0f  (groups %.0f x iters %d x 128 B)\n", groups * iters * 128.0, groups, iters);
  return 0;
}

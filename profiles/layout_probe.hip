// layout_probe.hip -- does the LAYOUT of a bank's per-item fields matter to a latency-bound stage?  A stand-in for k_shade_compact's memory
// shape at its occupancy (256 threads, 6 waves per SIMD, 26 KB of LDS per workgroup): per item 15 dword loads (one per field) whose values
// feed a dependent second wave of 4 loads (the item's neighbours), then 10 dword stores.  Three layouts of the same 15 x n words:
//   soa   field f of item w at base[f * stride + w], stride = n                      (the hot block of art_scene.h)
//   pad   the same with stride = n + 1088
//   blk   blocks of 64 items: field f of item w at base[((w >> 6) * 15 + f) * 64 + (w & 63)]   (a wave's 15 loads hit 3840 consecutive bytes)
// for n = 2^26 (C3's batch: the stride is 2^28 bytes) and n = 132,710,400 (C4's).  Prints ms per launch (HIP events, best and median of 7);
// run the PROGRAM several times: profiles/r5_shade/ab18 found k_shade_compact bimodal between processes on C3.
// build: hipcc -O3 --offload-arch=gfx950 -o layout_probe profiles/layout_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int LAYOUT>
__device__ __forceinline__ size_t at(size_t w, int f, size_t stride) {
  return LAYOUT == 2 ? ((w >> 6) * 15 + (size_t)f) * 64 + (w & 63) : (size_t)f * stride + w;
}

template <int LAYOUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_probe(const float* __restrict__ in, float* __restrict__ out, size_t n, size_t stride) {
  __shared__ float pad_lds[26 * 256];                    // 26 KB: six workgroups per CU, as the stage
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) pad_lds[blockIdx.x & 1023] = 0.0f;
  if (w >= n) return;
  float v[15];
#pragma unroll
  for (int f = 0; f < 15; ++f) v[f] = in[at<LAYOUT>(w, f, stride)];
  float acc = 0.0f;
#pragma unroll
  for (int f = 0; f < 15; ++f) acc += v[f];
  const size_t w2 = (w ^ 64) < n ? (w ^ 64) : w;          // a dependent second round trip (the neighbouring tile's first four fields)
  const int sel = (acc == 12345.0f) ? 1 : 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) acc += in[at<LAYOUT>(w2, f + sel, stride)];
#pragma unroll
  for (int f = 0; f < 10; ++f) __builtin_nontemporal_store(v[f] + acc, &out[at<LAYOUT>(w, f, stride)]);
}

template <int LAYOUT>
static void run(const char* name, const float* in, float* out, size_t n, size_t stride) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int it = 0; it < 8; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_probe<LAYOUT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, in, out, n, stride);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (it) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (19.0 + 10.0) * 4.0 * (double)n;
  std::printf("%-4s n %zu stride %zu: best %.3f ms  median %.3f ms  (%.0f GB/s at the median)\n", name, n, stride, ms[0], ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
}

int main() {
  const size_t nmax = 132710400ull, words = 15 * (nmax + 4096) + 4096;
  float *in, *out;
  CK(hipMalloc(&in, words * 4)); CK(hipMalloc(&out, words * 4));
  CK(hipMemset(in, 0, words * 4)); CK(hipMemset(out, 0, words * 4)); CK(hipDeviceSynchronize());
  for (size_t n : {(size_t)1 << 26, nmax}) {
    run<0>("soa", in, out, n, n);
    run<0>("pad", in, out, n, n + 1088);
    run<2>("blk", in, out, n, 0);
  }
  return 0;
}

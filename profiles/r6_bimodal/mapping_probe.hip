// mapping_probe.hip -- round 6, profiles/r6_bimodal: does the way a buffer is MAPPED change the rate of a many-stream kernel, outside the
// library?  The memory shape of k_shade_compact (layout_probe.hip of round 5): per item S dword loads, one per field, field f of item w at
// in[f * n + w]; a dependent second round trip of four loads; T non-temporal dword stores at out[f * n + w]; 256 threads, 6 waves per SIMD,
// 26 KB of LDS per workgroup.  n = 2^26 items (C3's batch), S = 15 / T = 10 (the stage's hot block) and S = 30 / T = 30 (about the stage's
// forty streams plus the trace records).  Backings of `in` and `out`:
//   malloc      hipMalloc
//   contiguous  hipExtMallocWithFlags(hipDeviceMallocContiguous)
//   vmm<MB>     one reserved address range over separately created chunks of <MB> MB (hipMemCreate / hipMemMap): 2, 64, 1024
// Prints GB/s at the median of 7 launches.  Run the PROGRAM several times (the driver's placement of a hipMalloc differs by process).
// build: hipcc -O3 --offload-arch=gfx950 -o mapping_probe profiles/r6_bimodal/mapping_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int S, int T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_probe(const float* __restrict__ in, float* __restrict__ out, size_t n) {
  __shared__ float pad_lds[26 * 256];                    // 26 KB: six workgroups per CU, as the stage
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) pad_lds[blockIdx.x & 1023] = 0.0f;
  if (w >= n) return;
  float v[S];
#pragma unroll
  for (int f = 0; f < S; ++f) v[f] = in[(size_t)f * n + w];
  float acc = 0.0f;
#pragma unroll
  for (int f = 0; f < S; ++f) acc += v[f];
  const size_t w2 = (w ^ 64) < n ? (w ^ 64) : w;          // a dependent second round trip (the neighbouring tile's first four fields)
  const int sel = (acc == 12345.0f) ? 1 : 0;
#pragma unroll
  for (int f = 0; f < 4; ++f) acc += in[(size_t)(f + sel) * n + w2];
#pragma unroll
  for (int f = 0; f < T; ++f) __builtin_nontemporal_store(v[f % S] + acc, &out[(size_t)f * n + w]);
}

// the same bytes in 16-byte fields: S4 float4 streams in (field f of item w at in4[f * n + w]), the dependent second round trip on the first
// field, T4 float4 non-temporal streams out -- what packing the stage's dword fields four to a field would look like to the memory system
template <int S4, int T4>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_probe4(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  __shared__ float pad_lds[26 * 256];
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (threadIdx.x == 0) pad_lds[blockIdx.x & 1023] = 0.0f;
  if (w >= n) return;
  float4 v[S4];
#pragma unroll
  for (int f = 0; f < S4; ++f) v[f] = in[(size_t)f * n + w];
  float acc = 0.0f;
#pragma unroll
  for (int f = 0; f < S4; ++f) acc += v[f].x + v[f].y + v[f].z + v[f].w;
  const size_t w2 = (w ^ 64) < n ? (w ^ 64) : w;
  const int sel = (acc == 12345.0f) ? 1 : 0;
  const float4 nb = in[(size_t)sel * n + w2];
  acc += nb.x + nb.y + nb.z + nb.w;
#pragma unroll
  for (int f = 0; f < T4; ++f) {
    const float4 o = make_float4(v[f % S4].x + acc, v[f % S4].y + acc, v[f % S4].z + acc, v[f % S4].w + acc);
    float* dst = reinterpret_cast<float*>(&out[(size_t)f * n + w]);
    typedef float v4f __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(v4f{o.x, o.y, o.z, o.w}, reinterpret_cast<v4f*>(dst));
  }
}

struct Buf { void* p = nullptr; size_t bytes = 0, chunk = 0, reserved = 0; std::vector<hipMemGenericAllocationHandle_t> h; };

static bool alloc(Buf& b, size_t bytes, int mode) {      // mode 0 malloc, -1 contiguous, > 0 chunk MB
  b = Buf(); b.bytes = bytes;
  if (mode == 0) return hipMalloc(&b.p, bytes) == hipSuccess;
  if (mode < 0) { const bool ok = hipExtMallocWithFlags(&b.p, bytes, hipDeviceMallocContiguous) == hipSuccess; if (!ok) (void)hipGetLastError(); return ok; }
  hipMemAllocationProp prop; std::memset(&prop, 0, sizeof prop);
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  b.chunk = (((size_t)mode << 20) + gran - 1) / gran * gran;
  const size_t nch = (bytes + b.chunk - 1) / b.chunk;
  b.reserved = nch * b.chunk;
  CK(hipMemAddressReserve(&b.p, b.reserved, 0, nullptr, 0));
  for (size_t i = 0; i < nch; ++i) {
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, b.chunk, &prop, 0)); b.h.push_back(h);
    CK(hipMemMap((char*)b.p + i * b.chunk, b.chunk, 0, h, 0));
  }
  hipMemAccessDesc acc; std::memset(&acc, 0, sizeof acc); acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  CK(hipMemSetAccess(b.p, b.reserved, &acc, 1));
  return true;
}
static void release(Buf& b) {
  if (b.reserved) { for (size_t i = 0; i < b.h.size(); ++i) { (void)hipMemUnmap((char*)b.p + i * b.chunk, b.chunk); (void)hipMemRelease(b.h[i]); } (void)hipMemAddressFree(b.p, b.reserved); }
  else if (b.p) (void)hipFree(b.p);
  b = Buf();
}

template <int S, int T>
static void run(const char* name, int mode, size_t n) {
  Buf in, out;
  if (!alloc(in, (size_t)(S + 1) * n * 4, mode) || !alloc(out, (size_t)T * n * 4, mode)) { std::printf("%-11s S %2d T %2d: allocation refused\n", name, S, T); release(in); release(out); return; }
  CK(hipMemset(in.p, 0, in.bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int it = 0; it < 8; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_probe<S, T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float*)in.p, (float*)out.p, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (it) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (double)(S + 4 + T) * 4.0 * (double)n;
  std::printf("%-11s S %2d T %2d: best %.3f ms  median %.3f ms  %.0f GB/s\n", name, S, T, ms[0], ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
  std::fflush(stdout);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  release(in); release(out);
}

template <int S4, int T4>
static void run4(const char* name, int mode, size_t n) {
  Buf in, out;
  if (!alloc(in, (size_t)(S4 + 1) * n * 16, mode) || !alloc(out, (size_t)T4 * n * 16, mode)) { std::printf("%-11s S4 %d T4 %d: allocation refused\n", name, S4, T4); release(in); release(out); return; }
  CK(hipMemset(in.p, 0, in.bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int it = 0; it < 8; ++it) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_probe4<S4, T4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float4*)in.p, (float4*)out.p, n);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float t; CK(hipEventElapsedTime(&t, a, b));
    if (it) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double bytes = (double)(S4 + 1 + T4) * 16.0 * (double)n;
  std::printf("%-11s %d + %d float4 streams (= %d + %d dwords per item): best %.3f ms  median %.3f ms  %.0f GB/s\n", name, S4, T4, 4 * S4, 4 * T4, ms[0], ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
  std::fflush(stdout);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  release(in); release(out);
}

int main() {
  const size_t n = (size_t)1 << 26;
  const struct { const char* name; int mode; } B[] = {{"malloc", 0}, {"contiguous", -1}, {"vmm2", 2}, {"vmm64", 64}, {"vmm1024", 1024}, {"malloc", 0}};
  for (auto& b : B) run<15, 10>(b.name, b.mode, n);
  for (auto& b : B) run<30, 30>(b.name, b.mode, n);
  for (auto& b : B) run4<8, 8>(b.name, b.mode, n);          // 32 + 32 dwords per item in 16 streams
  for (auto& b : B) run4<4, 3>(b.name, b.mode, n);          // 16 + 12: the hot block's 15 + 10
  return 0;
}

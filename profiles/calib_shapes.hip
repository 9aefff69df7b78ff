// calib_shapes.hip -- round 5: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on the access shapes this backend's kernels really have
// (MI355X_MICROARCH.md, HBM section: "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Round 1 calibrated ONE shape (profiles/calib_fetch.hip: random 128-byte segments read by 8 lanes x 16 B, the 8-wide node of that round)
// and every later summary doubled FETCH_SIZE on its strength.  The kernels have since changed shape:
//   k_node64      random 64-byte packets, 4 lanes x 16 B each            (k_trace_coop's node visit, csrc/art_qnode.h)
//   k_tri64       random 64-byte records, one lane reads 48 B of it      (k_trace_coop's triangle test: 3 x dwordx4 by the lanes that hold one)
//   k_soa4        coalesced 4-B-per-lane streams                         (the stages' SoA words: rays, flags, pdf ...)
//   k_hit16       coalesced 16-B-per-lane stream                         (hit records; the guide's own x2 case, the control)
//   k_shade64     random 64-byte records, one lane reads 40 B by dwords  (the stages' triangle shading record)
//   k_store16     coalesced 16-B-per-lane stores                         (trace records after LDS staging, hit records)
//   k_store4      coalesced 4-B-per-lane stores                          (the stages' SoA words)
//   k_stage_mix   the stage's traffic mix per item: 18 dword streams + one random 64-byte record in, 13 dword streams + 2 x 64 B staged out
// Every table is >= 2 GiB (>> 256 MiB Infinity Cache) and every kernel touches each byte once, so known bytes = what the kernel asks for.
// Run (two passes, program directly after --):
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out/fetch -- ./calib_shapes
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out/write -- ./calib_shapes
// The program prints the known bytes per kernel; profiles/calib_shapes.py divides.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t s) { return s * 1664525u + 1013904223u; }
__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

// random 64-byte packets: lane j of a quad reads bytes [16 j, 16 j + 16) of packet p(quad, i); iters packets per quad, dependent chain
__global__ void k_node64(const float4* __restrict__ t, uint32_t n_packets, int iters, float* sink) {
  const uint32_t quad = (blockIdx.x * blockDim.x + threadIdx.x) >> 2, j = threadIdx.x & 3;
  uint32_t s = mix(quad + 1u);
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    s = lcg(s);
    const uint32_t p = mix(s) % n_packets;
    const float4 v = t[(size_t)p * 4 + j];
    acc += v.x + v.w;
    s ^= (__float_as_uint(v.y) & 1u);
  }
  if (acc == 123.456f) sink[0] = acc;
}

// random 64-byte records, the reading lane takes 48 B (3 x 16 B); 16 of a wave's 64 lanes read (a leaf step serves about that many)
__global__ void k_tri64(const float4* __restrict__ t, uint32_t n_records, int iters, float* sink) {
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  if ((threadIdx.x & 3) != 0) return;
  uint32_t s = mix(gid + 7u);
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    s = lcg(s);
    const uint32_t p = mix(s) % n_records;
    const float4 a = t[(size_t)p * 4], b = t[(size_t)p * 4 + 1], c = t[(size_t)p * 4 + 2];
    acc += a.x + b.y + c.z;
    s ^= (__float_as_uint(a.y) & 1u);
  }
  if (acc == 123.456f) sink[0] = acc;
}

// n_streams coalesced dword streams of n items each (stream k at t + k * n): thread i reads item i of every stream
template <int NS>
__global__ void k_soa4(const float* __restrict__ t, size_t n, float* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < NS; ++k) acc += t[(size_t)k * n + i];
  if (acc == 123.456f) sink[0] = acc;
}

__global__ void k_hit16(const float4* __restrict__ t, size_t n, float* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = t[i];
  if (v.x + v.y + v.z + v.w == 123.456f) sink[0] = v.x;
}

// one random 64-byte record per lane, 10 dword loads of it (the shading record: three normals + the material id)
template <int TABLE_MB>
__global__ void k_shade64(const float* __restrict__ t, uint32_t n_records, size_t n, float* sink) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = mix((uint32_t)i * 2654435761u + 99u) % n_records;
  const float* r = t + (size_t)p * 16;
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 10; ++k) acc += r[k];
  if (acc == 123.456f) sink[0] = acc;
}

__global__ void k_store16(float4* __restrict__ t, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) t[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}
template <int NS>
__global__ void k_store4(float* __restrict__ t, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < NS; ++k) t[(size_t)k * n + i] = (float)(i + k);
}

// The stage's mix per item: reads 14 dword streams + 2 hit records (16 B) + one random 64-byte record; writes 13 dword streams, 2 hit
// records and two 64-byte trace records through an LDS stage (4 x 16 B per lane to consecutive addresses)
template <int TABLE_MB>
__global__ __launch_bounds__(256) void k_stage_mix(const float* __restrict__ in, const float4* __restrict__ hit_in, const float* __restrict__ shade, uint32_t n_records,
                                                   float* __restrict__ out, float4* __restrict__ hit_out, float4* __restrict__ rec, size_t n) {
  __shared__ float4 s_stage[4][4 * 65];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t first = i - lane;
  float v[14] = {};
  const bool on = i < n;
  float4 h0 = make_float4(0, 0, 0, 0), h1 = h0;
  float acc = 0.0f;
  if (on) {
#pragma unroll
    for (int k = 0; k < 14; ++k) v[k] = in[(size_t)k * n + i];
    h0 = hit_in[i]; h1 = hit_in[n + i];
    const uint32_t p = mix(__float_as_uint(h0.y) + (uint32_t)i * 2654435761u) % n_records;
    const float* r = shade + (size_t)p * 16;
#pragma unroll
    for (int k = 0; k < 10; ++k) acc += r[k];
#pragma unroll
    for (int k = 0; k < 13; ++k) out[(size_t)k * n + i] = v[k] + acc;
    hit_out[i] = make_float4(h0.x + acc, h0.y, h1.z, h1.w); hit_out[n + i] = make_float4(h1.x, h1.y + v[13], h0.z, h0.w);
  }
  for (int kind = 0; kind < 2; ++kind) {
    float4* st = s_stage[wave];
    st[lane] = make_float4(acc, h0.x, h1.x, (float)kind); st[65 + lane] = h0; st[130 + lane] = h1; st[195 + lane] = make_float4(v[0], v[1], v[2], v[3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4* o = rec + 4 * (2 * first + (size_t)kind * 64);
    for (int it = 0; it < 4; ++it) { const int g = it * 64 + lane; if (first + (g >> 2) < n) o[g] = st[(g & 3) * 65 + (g >> 2)]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

int main() {
  const size_t GiB = 1ull << 30;
  float *a, *b, *sink;
  CK(hipMalloc(&a, 6 * GiB)); CK(hipMalloc(&b, 10 * GiB)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 0, 6 * GiB)); CK(hipMemset(b, 0, 10 * GiB));
  CK(hipDeviceSynchronize());
  const int T = 256;
  {   // node packets: 2 GiB table, 256 CUs x 8 blocks x 256 threads = 131072 quads x 512 packets
    const uint32_t n_packets = (uint32_t)(2 * GiB / 64); const int blocks = 2048, iters = 512;
    hipLaunchKernelGGL(k_node64, dim3(blocks), dim3(T), 0, 0, (const float4*)a, n_packets, iters, sink);
    std::printf("known k_node64 fetch %.0f write 0\n", (double)blocks * T / 4 * iters * 64.0);
    hipLaunchKernelGGL(k_tri64, dim3(blocks), dim3(T), 0, 0, (const float4*)a, n_packets, iters, sink);
    std::printf("known k_tri64 fetch %.0f write 0 (48 B asked of every 64-byte record: %.0f)\n", (double)blocks * T / 4 * iters * 64.0, (double)blocks * T / 4 * iters * 48.0);
  }
  {   // 16 dword streams of 64 Mi items = 4 GiB
    const size_t n = 64ull << 20;
    hipLaunchKernelGGL((k_soa4<16>), dim3((unsigned)(n / T)), dim3(T), 0, 0, a, n, sink);
    std::printf("known k_soa4<16> fetch %.0f write 0\n", 16.0 * n * 4);
    hipLaunchKernelGGL(k_hit16, dim3((unsigned)((4 * GiB / 16) / T)), dim3(T), 0, 0, (const float4*)a, (size_t)(4 * GiB / 16), sink);
    std::printf("known k_hit16 fetch %.0f write 0\n", (double)(4 * GiB));
    hipLaunchKernelGGL((k_shade64<4096>), dim3((unsigned)(n / T)), dim3(T), 0, 0, a, (uint32_t)(4 * GiB / 64), n, sink);
    std::printf("known k_shade64<4096> fetch %.0f write 0 (40 B asked of every 64-byte record: %.0f)\n", 64.0 * n, 40.0 * n);
    // the same gather out of tables the size of the workloads' own shading tables (C4: 1 M triangles x 64 B; C3: 100 k): they sit in the
    // Infinity Cache / L2, so the memory-side counter and the rate are what the stages really see
    hipLaunchKernelGGL((k_shade64<64>), dim3((unsigned)(n / T)), dim3(T), 0, 0, a, (uint32_t)((64ull << 20) / 64), n, sink);
    std::printf("known k_shade64<64> fetch %.0f write 0 (64 MiB table: cache-resident, bytes asked)\n", 64.0 * n);
    hipLaunchKernelGGL((k_shade64<6>), dim3((unsigned)(n / T)), dim3(T), 0, 0, a, (uint32_t)((6ull << 20) / 64), n, sink);
    std::printf("known k_shade64<6> fetch %.0f write 0 (6 MiB table: cache-resident, bytes asked)\n", 64.0 * n);
  }
  {
    const size_t n16 = 4 * GiB / 16;
    hipLaunchKernelGGL(k_store16, dim3((unsigned)(n16 / T)), dim3(T), 0, 0, (float4*)b, n16);
    std::printf("known k_store16 fetch 0 write %.0f\n", (double)(4 * GiB));
    const size_t n = 64ull << 20;
    hipLaunchKernelGGL((k_store4<16>), dim3((unsigned)(n / T)), dim3(T), 0, 0, b, n);
    std::printf("known k_store4<16> fetch 0 write %.0f\n", 16.0 * n * 4);
  }
  {   // stage mix: n = 16 Mi items.  in: 14 n floats + 2 n hits (a); shade: 2 GiB of records (a + offset); out: 13 n floats, 2 n hits, 2 n records (b)
    const size_t n = 16ull << 20;
    const float* in = a; const float4* hit_in = (const float4*)(a + 14 * n); const float* shade = a + (4 * GiB / 4);
    float* out = b; float4* hit_out = (float4*)(b + 13 * n); float4* rec = (float4*)(b + 13 * n + 8 * n);
    hipLaunchKernelGGL((k_stage_mix<2048>), dim3((unsigned)(n / T)), dim3(T), 0, 0, in, hit_in, shade, (uint32_t)(2 * GiB / 64), out, hit_out, rec, n);
    std::printf("known k_stage_mix<2048> fetch %.0f write %.0f\n", (14.0 * 4 + 32 + 64) * n, (13.0 * 4 + 32 + 128) * n);
    hipLaunchKernelGGL((k_stage_mix<64>), dim3((unsigned)(n / T)), dim3(T), 0, 0, in, hit_in, shade, (uint32_t)((64ull << 20) / 64), out, hit_out, rec, n);
    std::printf("known k_stage_mix<64> fetch %.0f write %.0f (64 MiB shading table)\n", (14.0 * 4 + 32 + 64) * n, (13.0 * 4 + 32 + 128) * n);
    hipLaunchKernelGGL((k_stage_mix<6>), dim3((unsigned)(n / T)), dim3(T), 0, 0, in, hit_in, shade, (uint32_t)((6ull << 20) / 64), out, hit_out, rec, n);
    std::printf("known k_stage_mix<6> fetch %.0f write %.0f (6 MiB shading table)\n", (14.0 * 4 + 32 + 64) * n, (13.0 * 4 + 32 + 128) * n);
    // the same without the gather at all, and without the two hit records per item (what a stage that needs neither could reach)
    hipLaunchKernelGGL((k_stage_mix<0>), dim3((unsigned)(n / T)), dim3(T), 0, 0, in, hit_in, shade, 1u, out, hit_out, rec, n);
    std::printf("known k_stage_mix<0> fetch %.0f write %.0f (one shading record: no gather traffic)\n", (14.0 * 4 + 32) * n, (13.0 * 4 + 32 + 128) * n);
  }
  CK(hipDeviceSynchronize());
  return 0;
}

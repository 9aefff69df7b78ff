// ta_rate.hip -- sustained cost of the vector-memory (TA / L1) path per wave64 load instruction and CU, for the access shapes the
// trace kernel is made of.  Every load hits in the CU's vector L1 (an 8 KB region shared by all waves) unless the mode says
// otherwise, so what is timed is the address / tag / return path alone, not L2 or the fabric.  8 waves per SIMD (2048
// workgroups of 256 threads), 8 loads in flight per wave.  Build: hipcc -O3 --offload-arch=gfx950 profiles/ta_rate.hip
// Output: one JSON object, ns and clocks (at the measured rate of a dependent VALU chain) per wave instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <string>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

enum Mode {
  M_DWORD_QUAD12 = 0,   // dword, 4 lanes of a quad at stride 12 B inside one 64-B record, 16 quads in 16 different lines (the diagnostic load)
  M_X4_QUAD_SAME,       // dwordx4, the 4 lanes of a quad read the SAME 16 B, 16 different lines (node header)
  M_X3_QUAD12,          // dwordx3, lanes of a quad at stride 12 B, 16 different lines (child records)
  M_X4_QUAD64,          // dwordx4, the quad reads 64 contiguous bytes, 16 different lines (one-load node layout)
  M_DWORD_QUAD16,       // dword, the quad reads 16 contiguous bytes, 16 different lines
  M_DWORD_COALESCED,    // dword, 64 lanes read 256 contiguous bytes
  M_X4_COALESCED,       // dwordx4, 64 lanes read 1024 contiguous bytes
  M_X4_QUAD64_HALF,     // M_X4_QUAD64 with only quads 0..7 enabled (exec mask)
  M_X4_QUAD64_QUARTER,  // ... only quads 0..3
  M_X4_QUAD64_LANE0,    // ... only lane 0 of every quad
  M_X4_LANE_LINES,      // dwordx4, every lane in a different line (lane-per-ray node fetch), L1-resident 8 KB
  M_X4_QUAD_SAME_HALF,  // M_X4_QUAD_SAME with quads 0..7 enabled
  M_X4_TRI3,            // three dwordx4 per lane from a 64-B record per lane-of-quad (triangle fetch: quad reads 4 records = 256 B)
  M_X4_QUAD64_L2,       // M_X4_QUAD64 over a 2 MB region (misses L1, hits L2)
  M_X2_QUAD32,          // dwordx2, the quad reads 32 contiguous bytes
  M_COUNT
};
static const char* kNames[M_COUNT] = {"dword_quad_stride12", "x4_quad_same16", "x3_quad_stride12", "x4_quad_64B", "dword_quad_16B", "dword_coalesced", "x4_coalesced",
                                      "x4_quad_64B_half_exec", "x4_quad_64B_quarter_exec", "x4_quad_64B_lane0_exec", "x4_lane_per_line", "x4_quad_same16_half_exec",
                                      "x4_tri_3loads", "x4_quad_64B_L2", "x2_quad_32B"};

template <int MODE>
__global__ __launch_bounds__(256) void k_ta(const char* __restrict__ base, uint32_t* out, int iters) {
  const uint32_t lane = threadIdx.x & 63, quad = lane >> 2, j = lane & 3;
  uint32_t acc = 0;
  uint32_t line = quad * 5u + (threadIdx.x >> 6) * 3u;
  const bool on = (MODE == M_X4_QUAD64_HALF || MODE == M_X4_QUAD_SAME_HALF) ? (quad < 8) : (MODE == M_X4_QUAD64_QUARTER) ? (quad < 4) : (MODE == M_X4_QUAD64_LANE0) ? (j == 0) : true;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      line += 7u;
      uint32_t off;
      if (MODE == M_X4_QUAD64_L2) off = ((line * 2654435761u) >> 11) & 0x1fffc0u & ~127u;      // 128-B aligned, spread over 2 MB
      else if (MODE == M_X4_LANE_LINES) off = ((lane * 5u + line) & 63u) * 128u;
      else if (MODE == M_DWORD_COALESCED) off = (line & 31u) * 256u + 4u * lane;
      else if (MODE == M_X4_COALESCED) off = (line & 7u) * 1024u + 16u * lane;
      else off = (line & 63u) * 128u;
      asm volatile("" : "+v"(off));
      if (on) {
        if (MODE == M_DWORD_QUAD12) acc += *reinterpret_cast<const uint32_t*>(base + off + 24u + 12u * j);
        else if (MODE == M_X4_QUAD_SAME || MODE == M_X4_QUAD_SAME_HALF) { const u32x4 v = *reinterpret_cast<const u32x4*>(base + off); acc += v.x ^ v.w; }
        else if (MODE == M_X3_QUAD12) { const uint32_t* p = reinterpret_cast<const uint32_t*>(base + off + 16u + 12u * j); acc += p[0] ^ p[1] ^ p[2]; }
        else if (MODE == M_X4_QUAD64 || MODE == M_X4_QUAD64_HALF || MODE == M_X4_QUAD64_QUARTER || MODE == M_X4_QUAD64_LANE0 || MODE == M_X4_QUAD64_L2) { const u32x4 v = *reinterpret_cast<const u32x4*>(base + off + 16u * j); acc += v.x ^ v.w; }
        else if (MODE == M_DWORD_QUAD16) acc += *reinterpret_cast<const uint32_t*>(base + off + 4u * j);
        else if (MODE == M_DWORD_COALESCED) acc += *reinterpret_cast<const uint32_t*>(base + off);
        else if (MODE == M_X4_COALESCED || MODE == M_X4_LANE_LINES) { const u32x4 v = *reinterpret_cast<const u32x4*>(base + off); acc += v.x ^ v.w; }
        else if (MODE == M_X2_QUAD32) { const u32x2 v = *reinterpret_cast<const u32x2*>(base + off + 8u * j); acc += v.x ^ v.y; }
        else if (MODE == M_X4_TRI3) {
          const uint32_t o2 = ((line & 15u) * 256u + 64u * j) & 8191u;     // 4 records of 64 B per quad
          const u32x4 a = *reinterpret_cast<const u32x4*>(base + o2), b = *reinterpret_cast<const u32x4*>(base + o2 + 16u), c = *reinterpret_cast<const u32x4*>(base + o2 + 32u);
          acc += a.x ^ b.y ^ c.z;
        }
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// dependent VALU chain: 1 instruction issues per 4 clocks per wave -> the clock the chip holds in this kind of loop
__global__ __launch_bounds__(256) void k_clock(uint32_t* out, int iters) {
  const uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint32_t a = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[1] = (uint32_t)(t1 - t0); out[2] = (uint32_t)(r1 - r0); }
  out[4 + threadIdx.x] = a;
}

template <int MODE>
static double run(const char* base, uint32_t* out, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_ta<MODE>, dim3(blocks), dim3(256), 0, 0, base, out, iters / 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_ta<MODE>, dim3(blocks), dim3(256), 0, 0, base, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 2000;
  char* base; uint32_t* out;
  hipMalloc(&base, 4 << 20); hipMemset(base, 1, 4 << 20);
  hipMalloc(&out, (size_t)blocks * 256 * 4 + 4096);
  hipLaunchKernelGGL(k_clock, dim3(1), dim3(256), 0, 0, out, 20000);
  hipDeviceSynchronize();
  uint32_t clk[4]; hipMemcpy(clk, out, 16, hipMemcpyDeviceToHost);
  const double ghz = (double)clk[1] / ((double)clk[2] * 10.0);       // s_memrealtime ticks at 100 MHz
  double ms[M_COUNT];
#define RUN(M) ms[M] = run<M>(base, out, iters, blocks);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14)
  printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": 8, \"clock_GHz_idle_chain\": %.3f, \"modes\": {", p.name, cus, ghz);
  for (int m = 0; m < M_COUNT; ++m) {
    const double loads_per_lane = (m == M_X4_TRI3) ? 3.0 : 1.0;
    const double insts_per_cu = (double)iters * 8.0 * 32.0 * loads_per_lane;          // 32 waves per CU
    const double ns = ms[m] * 1e6 / insts_per_cu;
    printf("%s\"%s\": {\"ms\": %.3f, \"ns_per_wave_inst_per_cu\": %.3f, \"clk_at_2.2GHz\": %.2f}", m ? ", " : "", kNames[m], ms[m], ns, ns * 2.2);
  }
  printf("}}\n");
  return 0;
}

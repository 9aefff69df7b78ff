// valu_rate.hip -- measures the sustained wave64 VALU issue rate per SIMD on MI355X for the instruction kinds the trace kernel
// uses (fp32 mul/add, min/max, compare+select, DPP mov).  8 waves per SIMD, independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float c = 1.0001f, d = 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) { a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d); a4 = __builtin_fmaf(a4, c, d); a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d); }
      if (KIND == 1) { a0 = fminf(a0, a1 + 0.0f); a1 = fmaxf(a1, a2); a2 = fminf(a2, a3); a3 = fmaxf(a3, a4); a4 = fminf(a4, a5); a5 = fmaxf(a5, a6); a6 = fminf(a6, a7); a7 = fmaxf(a7, a0); }
      if (KIND == 2) { a0 = (a0 < a1) ? a2 : a0; a1 = (a1 < a2) ? a3 : a1; a2 = (a2 < a3) ? a4 : a2; a3 = (a3 < a4) ? a5 : a3; a4 = (a4 < a5) ? a6 : a4; a5 = (a5 < a6) ? a7 : a5; a6 = (a6 < a7) ? a0 : a6; a7 = (a7 < a0) ? a1 : a7; }
      if (KIND == 3) { a0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a1), 0xB1, 0xf, 0xf, false)); a1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a2), 0x4E, 0xf, 0xf, false)); a2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a3), 0x1B, 0xf, 0xf, false)); a3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a0), 0x141, 0xf, 0xf, false)); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(const char* name, int instr_per_iter, float* out) {
  const int iters = 20000, blocks = 256 * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100);
  hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_instr_per_simd = (double)iters * instr_per_iter * 8.0;   // 8 waves on each SIMD
  std::printf("%-10s %.3f ms  -> %.3f ns per wave-instruction per SIMD (at 2.4 GHz: %.2f cycles)\n", name, ms, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
}
int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0>("fma", 64, out); run<1>("minmax", 64 + 8, out); run<2>("cmp+sel", 128, out); run<3>("dpp+add", 64, out);
  return 0;
}

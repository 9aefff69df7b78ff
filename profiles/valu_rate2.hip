// valu_rate2.hip -- sustained issue cost (cycles per wave64 instruction per SIMD) of every instruction kind the trace kernel's
// node / leaf steps are made of, measured with inline asm so that the instruction is exactly the one named.  8 waves per SIMD
// (2048 workgroups of 256 threads), 8 independent dependency chains per wave.  Build: hipcc -O3 --offload-arch=gfx950.
// Output feeds bench.py's VALU-issue bound (profiles/<round>/valu_rate.json).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEF_KERNEL(NAME, ASMSTR, ...)                                                                   \
  __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters) {                                          \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = 1.0001f, c = 0.5f;                                                                                    \
    asm volatile("" : "+v"(b), "+v"(c));                                                                            \
    for (int i = 0; i < iters; ++i) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                               \
        asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c) __VA_ARGS__);                                          \
        asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c) __VA_ARGS__);                                          \
        asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c) __VA_ARGS__);                                          \
        asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c) __VA_ARGS__);                                          \
        asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c) __VA_ARGS__);                                          \
        asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c) __VA_ARGS__);                                          \
        asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c) __VA_ARGS__);                                          \
        asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c) __VA_ARGS__);                                          \
      }                                                                                                             \
    }                                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                             \
  }

DEF_KERNEL(fma, "v_fma_f32 %0, %0, %1, %2", )
DEF_KERNEL(mul, "v_mul_f32 %0, %0, %1", )
DEF_KERNEL(add, "v_add_f32 %0, %0, %1", )
DEF_KERNEL(sub, "v_sub_f32 %0, %0, %1", )
DEF_KERNEL(min, "v_min_f32 %0, %0, %1", )
DEF_KERNEL(max, "v_max_f32 %0, %0, %1", )
DEF_KERNEL(min3, "v_min3_f32 %0, %0, %1, %2", )
DEF_KERNEL(max3, "v_max3_f32 %0, %0, %1, %2", )
DEF_KERNEL(med3, "v_med3_f32 %0, %0, %1, %2", )
DEF_KERNEL(cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1", : "vcc")
DEF_KERNEL(cmp_sgpr, "v_cmp_lt_f32_e64 s[20:21], %0, %1", : "s20", "s21")
DEF_KERNEL(cmp_u32, "v_cmp_lt_u32 vcc, %0, %1", : "vcc")
DEF_KERNEL(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc", : "vcc")
DEF_KERNEL(cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", )
DEF_KERNEL(add_u32, "v_add_u32 %0, %0, %1", )
DEF_KERNEL(sub_u32, "v_sub_u32 %0, %0, %1", )
DEF_KERNEL(and_b32, "v_and_b32 %0, %0, %1", )
DEF_KERNEL(or_b32, "v_or_b32 %0, %0, %1", )
DEF_KERNEL(lshlrev, "v_lshlrev_b32 %0, 3, %0", )
DEF_KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1", )
DEF_KERNEL(add3, "v_add3_u32 %0, %0, %1, %2", )
DEF_KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2", )
DEF_KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 4, %1", )
DEF_KERNEL(mov, "v_mov_b32 %0, %1", )
DEF_KERNEL(mov_dpp, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", )
DEF_KERNEL(add_u32_dpp, "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", )
DEF_KERNEL(min_u32, "v_min_u32 %0, %0, %1", )
DEF_KERNEL(min_u32_dpp, "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", )
DEF_KERNEL(add_f32_dpp, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1", )
DEF_KERNEL(subbrev, "v_subbrev_co_u32 %0, vcc, 0, %0, vcc", : "vcc")
DEF_KERNEL(rcp, "v_rcp_f32 %0, %0", )
DEF_KERNEL(sqrt, "v_sqrt_f32 %0, %0", )
DEF_KERNEL(mul_lo, "v_mul_lo_u32 %0, %0, %1", )
DEF_KERNEL(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2", )
DEF_KERNEL(cvt_f32_u32, "v_cvt_f32_u32 %0, %0", )
DEF_KERNEL(bfe, "v_bfe_u32 %0, %0, 4, 8", )
DEF_KERNEL(xor3, "v_xor_b32 %0, %0, %1", )
DEF_KERNEL(perm, "v_perm_b32 %0, %0, %1, %2", )
DEF_KERNEL(div_fixup, "v_div_fixup_f32 %0, %0, %1, %2", )
DEF_KERNEL(div_fmas, "v_div_fmas_f32 %0, %0, %1, %2", : "vcc")
DEF_KERNEL(div_scale, "v_div_scale_f32 %0, vcc, %0, %1, %2", : "vcc")

// 128-bit pair kernels (64-bit operands)
#define DEF_KERNEL64(NAME, ASMSTR)                                                                                   \
  __global__ __launch_bounds__(256) void k_##NAME(float* out, int iters) {                                          \
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double b = 1.0001, c = 0.5;                                                                                     \
    asm volatile("" : "+v"(b), "+v"(c));                                                                            \
    for (int i = 0; i < iters; ++i) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                               \
        asm volatile(ASMSTR : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASMSTR : "+v"(a1) : "v"(b), "v"(c));         \
        asm volatile(ASMSTR : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASMSTR : "+v"(a3) : "v"(b), "v"(c));         \
        asm volatile(ASMSTR : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASMSTR : "+v"(a5) : "v"(b), "v"(c));         \
        asm volatile(ASMSTR : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASMSTR : "+v"(a7) : "v"(b), "v"(c));         \
      }                                                                                                             \
    }                                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                    \
  }
DEF_KERNEL64(fma_f64, "v_fma_f64 %0, %0, %1, %2")
DEF_KERNEL64(pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_KERNEL64(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEF_KERNEL64(pk_add_f32, "v_pk_add_f32 %0, %0, %1")

// LDS: one ds_read_b64 / ds_write_b64 per lane per instruction, stack-like addresses (8-byte entries, per-group stride)
__global__ __launch_bounds__(256) void k_lds_rw(float* out, int iters) {
  __shared__ uint2 s[256 * 8];
  uint2 v = make_uint2(threadIdx.x, 1u);
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      s[threadIdx.x * 8 + u] = v;
      __builtin_amdgcn_wave_barrier();
      const uint2 r = s[threadIdx.x * 8 + ((u + 3) & 7)];
      acc += r.x; v.y = acc;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}

struct Entry { const char* name; void (*fn)(float*, int); int per_iter; };

int main(int argc, char** argv) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 8, iters = 4000;
#define E(N) {#N, k_##N, 64}
  Entry es[] = {E(fma), E(mul), E(add), E(sub), E(min), E(max), E(min3), E(max3), E(med3), E(cmp_vcc), E(cmp_sgpr), E(cmp_u32), E(cndmask_vcc), E(cndmask_sgpr),
                E(add_u32), E(sub_u32), E(and_b32), E(or_b32), E(lshlrev), E(lshl_add), E(add3), E(and_or), E(lshl_or), E(mov), E(mov_dpp), E(add_u32_dpp),
                E(min_u32), E(min_u32_dpp), E(add_f32_dpp), E(subbrev), E(pk_fma), E(rcp), E(sqrt), E(mul_lo), E(mad_u24), E(cvt_f32_u32), E(bfe), E(xor3), E(perm),
                E(div_fixup), E(div_fmas), E(div_scale), E(fma_f64), E(pk_mul_f32), E(pk_add_f32), {"lds_write+read_b64", k_lds_rw, 16}};
  // clock: measure with the fma kernel's known work?  No: report ns and cycles at the clock GRBM reports elsewhere (2.35-2.4 GHz); the
  // ratios between kinds are what the bound uses.
  std::printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_simd\": 8, \"ns_per_wave_instruction_per_simd\": {", prop.gcnArchName, cus);
  bool first = true;
  for (const Entry& e : es) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)iters * e.per_iter * 8.0;
    std::printf("%s\"%s\": %.4f", first ? "" : ", ", e.name, ms * 1e6 / wave_instr_per_simd);
    first = false;
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
  std::printf("}}\n");
  return 0;
}

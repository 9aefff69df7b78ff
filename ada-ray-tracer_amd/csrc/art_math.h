// art_math.h -- float3 / float4x4 arithmetic, ART-M1 transcendentals and the counter-based RNG
// used by the MI355X render backend.  Every function here is host+device so that the C++ host
// mirror (scene setup, camera constants) and the HIP kernels evaluate bit-identical values.
//
// Arithmetic contract (DESIGN.md "Numerics"): IEEE binary32, the reference's operation order
// (generic_vector_math.adb:81-156, vector_math.adb:14-153), no FMA contraction (-ffp-contract=off),
// correctly rounded / and sqrt, denormals preserved.  min/max keep the reference's NaN-asymmetric
// compare-select form (generic_vector_math.adb:19-35), not fminf/fmaxf.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ART_HD __host__ __device__ __forceinline__
#else
#define ART_HD inline
#endif

namespace art {

struct f3 { float x, y, z; };

// A pointer that is KNOWN to point into LDS (device pass; an ordinary pointer elsewhere).  Reads through it are ds_read instructions, which
// wait on lgkmcnt alone; the same table behind a generic pointer is read with flat_load, whose s_waitcnt vmcnt(0) also waits for every
// store the wave still has in flight -- in k_shade_compact that put a store-queue drain in front of each ray's sphere loop (round 5).
#if defined(__HIP_DEVICE_COMPILE__)
#define ART_LDS __attribute__((address_space(3)))
#else
#define ART_LDS
#endif

// a generic pointer to a __shared__ object as an LDS-qualified one (the low 32 bits of a generic LDS address are the LDS address)
template <class T> ART_HD const ART_LDS T* as_lds(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (const ART_LDS T*)(uint32_t)(uintptr_t)p;
#else
  return p;
#endif
}

// scalar base + 32-bit byte offset: on the device the address costs no vector instruction and no VGPR pair (a 64-bit pointer per array costs a
// v_lshl_add_u64 and two registers each).  The caller guarantees that the offset fits 32 bits.  at(base, i) = base[i] through it.
template <class T> ART_HD T ld_off(const T* base, uint32_t byte_off) { return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off); }
template <class T> ART_HD void st_off(T* base, uint32_t byte_off, T v) { *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v; }
// at(base, i) / put(base, i, v): base[i] that way -- every per-item array of a path bank has fewer than 2^27 items of at most 16 bytes
template <class T> ART_HD T at(const T* base, int i) { return ld_off(base, (uint32_t)i * (uint32_t)sizeof(T)); }
template <class T> ART_HD void put(T* base, int i, T v) { st_off(base, (uint32_t)i * (uint32_t)sizeof(T), v); }
// Streaming variants (round 5): what a wavefront stage reads was written by another kernel and is read once; what it writes is read once by
// a later kernel.  Non-temporal accesses keep that traffic from displacing the lines that ARE re-used (the triangles' shading records, the
// material table, the next rounds' windows).  ART_NT bits: 1 stores, 2 loads.
#ifndef ART_NT
#define ART_NT 1
#endif
template <class T> ART_HD void put_s(T* base, int i, T v) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((ART_NT & 1) != 0 && sizeof(T) == 4) { __builtin_nontemporal_store(v, reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (uint32_t)i * 4u)); return; }
  else if constexpr ((ART_NT & 1) != 0 && sizeof(T) == 16) {
    typedef float f4nt __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(__builtin_bit_cast(f4nt, v), reinterpret_cast<f4nt*>(reinterpret_cast<char*>(base) + (uint32_t)i * 16u)); return;
  }
#endif
  put(base, i, v);
}
template <class T> ART_HD T at_s(const T* base, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr ((ART_NT & 2) != 0 && sizeof(T) == 4) return __builtin_nontemporal_load(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (uint32_t)i * 4u));
  else if constexpr ((ART_NT & 2) != 0 && sizeof(T) == 16) {
    typedef float f4nt __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(T, __builtin_nontemporal_load(reinterpret_cast<const f4nt*>(reinterpret_cast<const char*>(base) + (uint32_t)i * 16u)));
  }
#endif
  return at(base, i);
}

constexpr float kInfinity = 3.4028234663852886e38f;  // vector_math.ads:17 (Float'Last, finite)
constexpr float kPi       = 0x1.921fb6p+1f;          // vector_math.ads:19
constexpr float kInvPi    = 0x1.45f306p-2f;          // vector_math.ads:20
constexpr float kHalfPi   = 0x1.921fb6p+0f;

ART_HD float amin(float a, float b) { return (a < b) ? a : b; }
ART_HD float amax(float a, float b) { return (a >= b) ? a : b; }
ART_HD float amax3(float a, float b, float c) {   // generic_vector_math.adb:48-57
  if (a >= b && a >= c) return a;
  if (b >= c && b >= a) return b;
  return c;
}
ART_HD float aclamp(float x, float lo, float hi) { return amin(amax(x, lo), hi); }
ART_HD float asign(float x) { return (x >= 0.0f) ? 1.0f : -1.0f; }            // vector_math.adb:49-56
ART_HD float alerp(float t, float a, float b) { return (1.0f - t) * a + t * b; } // vector_math.adb:59-62

ART_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
ART_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
ART_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
ART_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
ART_HD f3 operator*(f3 a, float k) { return mk3(k * a.x, k * a.y, k * a.z); }
ART_HD f3 operator*(float k, f3 a) { return mk3(k * a.x, k * a.y, k * a.z); }
ART_HD float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ART_HD f3 cross(f3 a, f3 b) {
  return mk3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
ART_HD f3 neg(f3 a) { return a * -1.0f; }   // the reference always writes (-1.0)*v
ART_HD f3 normalize(f3 a) {                  // vector_math.adb:64-72
  float li = 1.0f / sqrtf(dot(a, a));
  return mk3(li * a.x, li * a.y, li * a.z);
}
ART_HD float length(f3 a) { return sqrtf(dot(a, a)); }
ART_HD f3 reflect(f3 dir, f3 n) { return normalize(((n * dot(dir, n)) * -2.0f) + dir); }  // vector_math.adb:79-82

// float4x4 (row-major, generic_vector_math.ads:64); m*v adds the translation column (vector_math.adb:137-144)
// world normal of an object-space vertex normal under an instance whose inverse 3x4 is minv: inverse transpose of the 3x3, normalised
// (instanced scenes, art_scene.h DevInstance; the flattening of art_host_scene.cpp and the shade stage evaluate this very expression)
ART_HD f3 instance_normal(const float* minv, f3 n) { return normalize(mk3(minv[0] * n.x + minv[4] * n.y + minv[8] * n.z, minv[1] * n.x + minv[5] * n.y + minv[9] * n.z, minv[2] * n.x + minv[6] * n.y + minv[10] * n.z)); }
ART_HD f3 xform_point(const float* m, f3 v) {
  return mk3(m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3],
             m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7],
             m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11]);
}

// ------------------------------------------------------------------------------------------------
// ART-M1: sin / cos / tan / "**" evaluated in binary64 with + - * / only, rounded once to binary32.
// They stand in for Ada.Numerics.Generic_Elementary_Functions (GNAT runtime, not in the reference).
// ------------------------------------------------------------------------------------------------
namespace m1 {

// A binary64 coefficient on the device is two s_mov_b32 literals in front of the instruction that uses it.  Left to itself the compiler
// hoists the pairs out of k_shade_compact's round loop, runs out of SGPRs and keeps them in VGPR lanes (v_writelane / v_readlane: the 70
// "spilled SGPRs" of the round-5 stage were exactly these constants).  ART_KD pins a coefficient to its use: an empty volatile asm on
// its two halves, which is neither hoisted nor merged -- the value is the literal's, bit for bit (round 6, review item 6).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ART_F64_CONST_FREE)
__device__ __forceinline__ double kd_at_use(double c) {
  const uint64_t b = __builtin_bit_cast(uint64_t, c);
  uint32_t lo = (uint32_t)b, hi = (uint32_t)(b >> 32);
  asm volatile("" : "+s"(lo), "+s"(hi));
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | (uint64_t)lo);
}
#define ART_KD(x) ::art::m1::kd_at_use(x)
#else
#define ART_KD(x) (x)
#endif

ART_HD double poly_sin(double r) {   // |r| <= pi/4, odd Taylor series through r^15
  const double z = r * r;
  double q = ART_KD(0x1.ae7f3e733b81fp-41);
  q = q * z - ART_KD(0x1.6124613a86d09p-33);
  q = q * z + ART_KD(0x1.ae64567f544e4p-26);
  q = q * z - ART_KD(0x1.71de3a556c734p-19);
  q = q * z + ART_KD(0x1.a01a01a01a01ap-13);
  q = q * z - ART_KD(0x1.1111111111111p-7);
  q = q * z + ART_KD(0x1.5555555555555p-3);
  return r - (r * z) * q;
}

ART_HD double poly_cos(double r) {   // |r| <= pi/4, even Taylor series through r^16
  const double z = r * r;
  double q = ART_KD(0x1.ae7f3e733b81fp-45);
  q = q * z - ART_KD(0x1.93974a8c07c9dp-37);
  q = q * z + ART_KD(0x1.1eed8eff8d898p-29);
  q = q * z - ART_KD(0x1.27e4fb7789f5cp-22);
  q = q * z + ART_KD(0x1.a01a01a01a01ap-16);
  q = q * z - ART_KD(0x1.6c16c16c16c17p-10);
  q = q * z + ART_KD(0x1.5555555555555p-5);
  q = q * z - 0.5;
  return 1.0 + z * q;
}

// x = k*(pi/2) + r, two-constant Cody-Waite; exact for |x| < 2^20
ART_HD void sincos(double x, double& s, double& c) {
  const double v = x * ART_KD(0x1.45f306dc9c883p-1);
  const int k = (int)(v + (v >= 0.0 ? 0.5 : -0.5));
  const double kd = (double)k;
  const double r = (x - kd * ART_KD(0x1.921fb54400000p+0)) - kd * ART_KD(0x1.0b4611a626331p-34);
  const double sr = poly_sin(r), cr = poly_cos(r);
  switch (k & 3) {
    case 0:  s = sr;  c = cr;  break;
    case 1:  s = cr;  c = -sr; break;
    case 2:  s = -sr; c = -cr; break;
    default: s = -cr; c = sr;  break;
  }
}

ART_HD double log_pos(double x) {   // x: positive, normal binary64
  uint64_t b = __builtin_bit_cast(uint64_t, x);
  int e = (int)((b >> 52) & 0x7ff) - 1023;
  double m = __builtin_bit_cast(double, (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);
  if (m > ART_KD(0x1.6a09e667f3bcdp+0)) { m = m * 0.5; e = e + 1; }
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double q = ART_KD(0x1.8618618618618p-5);
  q = q * z + ART_KD(0x1.af286bca1af28p-5);
  q = q * z + ART_KD(0x1.e1e1e1e1e1e1ep-5);
  q = q * z + ART_KD(0x1.1111111111111p-4);
  q = q * z + ART_KD(0x1.3b13b13b13b14p-4);
  q = q * z + ART_KD(0x1.745d1745d1746p-4);
  q = q * z + ART_KD(0x1.c71c71c71c71cp-4);
  q = q * z + ART_KD(0x1.2492492492492p-3);
  q = q * z + ART_KD(0x1.999999999999ap-3);
  q = q * z + ART_KD(0x1.5555555555555p-2);
  const double lm = 2.0 * s + (2.0 * s) * (z * q);
  return (double)e * ART_KD(0x1.62e42fefa39efp-1) + lm;
}

ART_HD double exp_small(double t) {   // |t| <= 200
  const double v = t * ART_KD(0x1.71547652b82fep+0);
  const int k = (int)(v + (v >= 0.0 ? 0.5 : -0.5));
  const double kd = (double)k;
  const double r = (t - kd * ART_KD(0x1.62e42fee00000p-1)) - kd * ART_KD(0x1.a39ef35793c76p-33);
  double q = ART_KD(0x1.6124613a86d09p-33);
  q = q * r + ART_KD(0x1.1eed8eff8d898p-29);
  q = q * r + ART_KD(0x1.ae64567f544e4p-26);
  q = q * r + ART_KD(0x1.27e4fb7789f5cp-22);
  q = q * r + ART_KD(0x1.71de3a556c734p-19);
  q = q * r + ART_KD(0x1.a01a01a01a01ap-16);
  q = q * r + ART_KD(0x1.a01a01a01a01ap-13);
  q = q * r + ART_KD(0x1.6c16c16c16c17p-10);
  q = q * r + ART_KD(0x1.1111111111111p-7);
  q = q * r + ART_KD(0x1.5555555555555p-5);
  q = q * r + ART_KD(0x1.5555555555555p-3);
  q = q * r + 0.5;
  q = q * r + 1.0;
  q = q * r + 1.0;
  const double scale = __builtin_bit_cast(double, (uint64_t)(int64_t)(k + 1023) << 52);
  return q * scale;
}

}  // namespace m1

ART_HD float asin_m1(float x) { double s, c; m1::sincos((double)x, s, c); return (float)s; }
ART_HD float acos_m1(float x) { double s, c; m1::sincos((double)x, s, c); return (float)c; }
ART_HD float atan_m1(float x) { double s, c; m1::sincos((double)x, s, c); return (float)(s / c); }
ART_HD void asincos_m1(float x, float& s, float& c) { double sd, cd; m1::sincos((double)x, sd, cd); s = (float)sd; c = (float)cd; }

// Ada "**" for Float (RM A.5.1; GNAT special-cases 2.0 and 0.5) == vector_math.adb:24-47 `pow`.
ART_HD float apow(float x, float y) {
  const float qnan = __builtin_bit_cast(float, 0x7fc00000u);
  const float pinf = __builtin_bit_cast(float, 0x7f800000u);
  if (x != x || y != y) return qnan;
  if (x == 0.0f && y == 0.0f) return qnan;   // Argument_Error
  if (x < 0.0f) return qnan;                 // Argument_Error
  if (y == 0.0f) return 1.0f;
  if (x == 0.0f) return (y < 0.0f) ? pinf : 0.0f;
  if (x == 1.0f) return 1.0f;
  if (y == 1.0f) return x;
  if (y == 2.0f) return x * x;
  if (y == 0.5f) return sqrtf(x);
  if (x > kInfinity) return (y > 0.0f) ? pinf : 0.0f;
  const double t = (double)y * m1::log_pos((double)x);
  if (t > 200.0) return pinf;
  if (t < -200.0) return 0.0f;
  return (float)m1::exp_small(t);
}

ART_HD float safe_tan(float x) {   // vector_math.adb:14-22
  return (fabsf(x) == kHalfPi) ? kInfinity : atan_m1(x);
}

// ------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10, counter = (pixel, sample, bounce, stream), key = seed.  One call yields the
// four uniforms of a bounce: .x,.y light sample, .z,.w BSDF sample (SURVEY Appendix B draw order).
// ------------------------------------------------------------------------------------------------
struct u4 { uint32_t x, y, z, w; };

ART_HD u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  u4 r; r.x = c0; r.y = c1; r.z = c2; r.w = c3;
  return r;
}

ART_HD float u01(uint32_t u) {   // [0,1) on a 2^-24 grid; rnd_uniform(0,1) = 0 + (1-0)*t  (vector_math.adb:170)
  const float t = (float)(u >> 8) * 0x1.0p-24f;
  return 0.0f + (1.0f - 0.0f) * t;
}

}  // namespace art

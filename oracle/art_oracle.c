/*
 * art_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See art_oracle.h.
 *
 * Literal plain-C restatement of the reference (paths relative to /root/reference).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  All arithmetic is
 * IEEE binary32 in the reference's operation order; Ada "Constraint_Error" handlers on float
 * overflow are dead code on GNAT/x86-64 (Float'Machine_Overflows = False) and are restated as
 * plain IEEE inf/NaN propagation.
 *
 * Parity: UNPINNED at bit level vs the Ada binary (GNAT RNG + elementary functions are not in
 * the reference tree, and the reference holds no golden vectors).  Pinned as far as the reference allows:
 * statistically against the reference's own output picture image.png AS A WHOLE (64 x 64 block means + 18 named regions, rendered
 * from the picture's camera: tests/picture_pin.py, tests/test_oracle_image_pin.py), and bit for bit against an independent
 * numpy-float32 transcription of lights.adb / materials.adb / the sampling helpers (tests/ada_transcription.py, tests/test_sampling_kat.py).
 */
#include "art_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================================== */
/* ART-M1 transcendental functions (replace Ada.Numerics.Generic_Elementary_Functions).     */
/* Double-precision kernels using only + - * / and bit moves, rounded once to float.        */
/* ======================================================================================== */

static const double M1_TWO_OVER_PI = 0x1.45f306dc9c883p-1;
static const double M1_PIO2_HI     = 0x1.921fb54400000p+0;   /* 33 bits of pi/2 */
static const double M1_PIO2_LO     = 0x1.0b4611a626331p-34;
static const double M1_LN2         = 0x1.62e42fefa39efp-1;
static const double M1_LN2_HI      = 0x1.62e42fee00000p-1;
static const double M1_LN2_LO      = 0x1.a39ef35793c76p-33;
static const double M1_INV_LN2     = 0x1.71547652b82fep+0;
static const double M1_SQRT2       = 0x1.6a09e667f3bcdp+0;

/* sin(r), |r| <= pi/4 : odd Taylor polynomial through r^15 */
static double m1_ksin(double r) {
  double z = r * r;
  double p = 0x1.ae7f3e733b81fp-41;          /* 1/15! */
  p = p * z - 0x1.6124613a86d09p-33;         /* 1/13! */
  p = p * z + 0x1.ae64567f544e4p-26;         /* 1/11! */
  p = p * z - 0x1.71de3a556c734p-19;         /* 1/9!  */
  p = p * z + 0x1.a01a01a01a01ap-13;         /* 1/7!  */
  p = p * z - 0x1.1111111111111p-7;          /* 1/5!  */
  p = p * z + 0x1.5555555555555p-3;          /* 1/3!  */
  return r - (r * z) * p;
}

/* cos(r), |r| <= pi/4 : even Taylor polynomial through r^16 */
static double m1_kcos(double r) {
  double z = r * r;
  double p = 0x1.ae7f3e733b81fp-45;          /* 1/16! */
  p = p * z - 0x1.93974a8c07c9dp-37;         /* 1/14! */
  p = p * z + 0x1.1eed8eff8d898p-29;         /* 1/12! */
  p = p * z - 0x1.27e4fb7789f5cp-22;         /* 1/10! */
  p = p * z + 0x1.a01a01a01a01ap-16;         /* 1/8!  */
  p = p * z - 0x1.6c16c16c16c17p-10;         /* 1/6!  */
  p = p * z + 0x1.5555555555555p-5;          /* 1/4!  */
  p = p * z - 0.5;                           /* 1/2!  */
  return 1.0 + z * p;
}

/* Cody-Waite reduction: x = k*(pi/2) + r.  Valid for |x| < 2^20 (all uses are |x| <= 2*pi). */
static double m1_reduce(double x, int* quadrant) {
  double v = x * M1_TWO_OVER_PI;
  long long k = (long long)(v + (v >= 0.0 ? 0.5 : -0.5));
  double kd = (double)k;
  double r = (x - kd * M1_PIO2_HI) - kd * M1_PIO2_LO;
  *quadrant = (int)(k & 3);
  return r;
}

static void m1_sincos(double x, double* s, double* c) {
  int q;
  double r = m1_reduce(x, &q);
  double sr = m1_ksin(r), cr = m1_kcos(r);
  switch (q) {
    case 0:  *s =  sr; *c =  cr; break;
    case 1:  *s =  cr; *c = -sr; break;
    case 2:  *s = -sr; *c = -cr; break;
    default: *s = -cr; *c =  sr; break;
  }
}

float orc_sinf(float x) { double s, c; m1_sincos((double)x, &s, &c); return (float)s; }
float orc_cosf(float x) { double s, c; m1_sincos((double)x, &s, &c); return (float)c; }
float orc_tanf(float x) { double s, c; m1_sincos((double)x, &s, &c); return (float)(s / c); }

/* natural log of a positive finite double that came from a float (always a normal double) */
static double m1_log(double x) {
  uint64_t b; memcpy(&b, &x, 8);
  int e = (int)((b >> 52) & 0x7ff) - 1023;
  b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
  double m; memcpy(&m, &b, 8);                     /* m in [1,2) */
  if (m > M1_SQRT2) { m = m * 0.5; e = e + 1; }   /* m in (sqrt(1/2), sqrt(2)] */
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  double p = 0x1.8618618618618p-5;                 /* 1/21 */
  p = p * z + 0x1.af286bca1af28p-5;                /* 1/19 */
  p = p * z + 0x1.e1e1e1e1e1e1ep-5;                /* 1/17 */
  p = p * z + 0x1.1111111111111p-4;                /* 1/15 */
  p = p * z + 0x1.3b13b13b13b14p-4;                /* 1/13 */
  p = p * z + 0x1.745d1745d1746p-4;                /* 1/11 */
  p = p * z + 0x1.c71c71c71c71cp-4;                /* 1/9  */
  p = p * z + 0x1.2492492492492p-3;                /* 1/7  */
  p = p * z + 0x1.999999999999ap-3;                /* 1/5  */
  p = p * z + 0x1.5555555555555p-2;                /* 1/3  */
  double lm = 2.0 * s + (2.0 * s) * (z * p);
  return (double)e * M1_LN2 + lm;
}

/* exp(t) for |t| <= 200 */
static double m1_exp(double t) {
  double v = t * M1_INV_LN2;
  long long k = (long long)(v + (v >= 0.0 ? 0.5 : -0.5));
  double kd = (double)k;
  double r = (t - kd * M1_LN2_HI) - kd * M1_LN2_LO;
  double p = 0x1.6124613a86d09p-33;                /* 1/13! */
  p = p * r + 0x1.1eed8eff8d898p-29;               /* 1/12! */
  p = p * r + 0x1.ae64567f544e4p-26;               /* 1/11! */
  p = p * r + 0x1.27e4fb7789f5cp-22;               /* 1/10! */
  p = p * r + 0x1.71de3a556c734p-19;               /* 1/9!  */
  p = p * r + 0x1.a01a01a01a01ap-16;               /* 1/8!  */
  p = p * r + 0x1.a01a01a01a01ap-13;               /* 1/7!  */
  p = p * r + 0x1.6c16c16c16c17p-10;               /* 1/6!  */
  p = p * r + 0x1.1111111111111p-7;                /* 1/5!  */
  p = p * r + 0x1.5555555555555p-5;                /* 1/4!  */
  p = p * r + 0x1.5555555555555p-3;                /* 1/3!  */
  p = p * r + 0.5;
  p = p * r + 1.0;
  p = p * r + 1.0;
  uint64_t sb = (uint64_t)(k + 1023) << 52;
  double sc; memcpy(&sc, &sb, 8);
  return p * sc;
}

/* Ada "**" on Float (RM A.5.1(13-20); GNAT additionally special-cases 2.0 and 0.5), together
 * with the wrapper vector_math.adb:24-47 which tests the same special cases in the same order.
 * Argument_Error / Constraint_Error cases return NaN / +inf (the Ada task would die). */
float orc_powf(float x, float y) {
  if (x != x || y != y) return NAN;
  if (x == 0.0f && y == 0.0f) return NAN;
  if (x < 0.0f) return NAN;
  if (y == 0.0f) return 1.0f;
  if (x == 0.0f) return (y < 0.0f) ? INFINITY : 0.0f;
  if (x == 1.0f) return 1.0f;
  if (y == 1.0f) return x;
  if (y == 2.0f) return x * x;
  if (y == 0.5f) return sqrtf(x);
  if (x > 3.4028234663852886e38f) return (y > 0.0f) ? INFINITY : 0.0f;
  {
    double t = (double)y * m1_log((double)x);
    if (t > 200.0) return INFINITY;
    if (t < -200.0) return 0.0f;
    return (float)m1_exp(t);
  }
}

/* ======================================================================================== */
/* RNG: replaces Ada.Numerics.Float_Random (vector_math.adb:166-171).                       */
/* Philox4x32-10 (Salmon et al., SC'11), counter = (pixel, sample, bounce, stream),         */
/* key = (seed_lo, seed_hi).  Slots 0,1 = light sample, 2,3 = BSDF sample;  */
/* stream 1 slot 0 (exposed as slot 4) = light selection when a scene has several lights.   */
/* ======================================================================================== */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int i = 0; i < 10; ++i) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n1 = lo1;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    uint32_t n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

float orc_rng_uniform(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t slot) {
  uint32_t ctr[4] = { pixel, sample, bounce, slot >> 2 };
  uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
  uint32_t o[4];
  orc_philox4x32_10(ctr, key, o);
  return (float)(o[slot & 3] >> 8) * 0x1.0p-24f;     /* [0,1), 24 bits; rnd_uniform(0,1): l + (h-l)*t */
}

typedef struct { uint64_t seed; uint32_t pixel, sample, bounce; } rng_ctx;

static float rnd(const rng_ctx* g, uint32_t slot) {
  float t = orc_rng_uniform(g->seed, g->pixel, g->sample, g->bounce, slot);
  return 0.0f + (1.0f - 0.0f) * t;                    /* vector_math.adb:170 */
}

/* ======================================================================================== */
/* generic_vector_math.adb / vector_math.adb                                                */
/* ======================================================================================== */

typedef struct { float x, y, z; } f3;

static const float ORC_INFINITY = 3.4028234663852886e38f;   /* vector_math.ads:17  float'Last */
static const float M_PI_F   = 0x1.921fb6p+1f;               /* vector_math.ads:19  (0x40490fdb) */
static const float INV_PI_F = 0x1.45f306p-2f;               /* vector_math.ads:20  (0x3ea2f983) */

static inline float min2(float a, float b) { return (a < b) ? a : b; }            /* generic_vector_math.adb:19-26 */
static inline float max2(float a, float b) { return (a >= b) ? a : b; }           /* :28-35 */
static inline float max3(float a, float b, float c) {                              /* :48-57 */
  if (a >= b && a >= c) return a; else if (b >= c && b >= a) return b; else return c;
}
static inline float clampf(float x, float a, float b) { return min2(max2(x, a), b); } /* :60-63 */

static inline f3 v3(float x, float y, float z) { f3 r = { x, y, z }; return r; }
static inline f3 add(f3 a, f3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :81-88 */
static inline f3 sub(f3 a, f3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :92-99 */
static inline f3 mulv(f3 a, f3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }  /* :101-108 */
static inline f3 muls(f3 a, float k) { return v3(k * a.x, k * a.y, k * a.z); }      /* :148-156 */
static inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* :110-113 */
static inline f3 cross(f3 a, f3 b) {                                                /* :115-122 */
  return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline f3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

static inline float signf_ada(float x) { return (x >= 0.0f) ? 1.0f : -1.0f; }       /* vector_math.adb:49-56 */
static inline float lerpf(float t, float a, float b) { return (1.0f - t) * a + t * b; } /* :59-62 */
static inline f3 normalize(f3 a) {                                                   /* :64-72 */
  float l_inv = 1.0f / sqrtf(dot(a, a));
  return v3(l_inv * a.x, l_inv * a.y, l_inv * a.z);
}
static inline float length3(f3 a) { return sqrtf(dot(a, a)); }                      /* :74-77 */
static inline f3 reflect(f3 dir, f3 normal) {                                        /* :79-82 */
  return normalize(add(muls(muls(normal, dot(dir, normal)), -2.0f), dir));
}
static inline f3 mat_mul_v(const float* m, f3 v) {                                   /* :137-144 (adds translation) */
  f3 r;
  r.x = m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3];
  r.y = m[4] * v.x + m[5] * v.y + m[6] * v.z + m[7];
  r.z = m[8] * v.x + m[9] * v.y + m[10] * v.z + m[11];
  return r;
}
static void mat_mul_m(const float* a, const float* b, float* o) {                    /* generic_vector_math.adb:233-256 */
  float t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      t[i * 4 + j] = a[i * 4 + 0] * b[0 * 4 + j] + a[i * 4 + 1] * b[1 * 4 + j] + a[i * 4 + 2] * b[2 * 4 + j] + a[i * 4 + 3] * b[3 * 4 + j];
  memcpy(o, t, sizeof t);
}
static void mat_identity(float* m) { memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }

static void rotation_matrix(float angle, f3 a_v, float* M) {                         /* vector_math.adb:85-111 */
  mat_identity(M);
  f3 v = normalize(a_v);
  float cos_t = orc_cosf(angle), sin_t = orc_sinf(angle);
  M[0]  = (1.0f - cos_t) * v.x * v.x + cos_t;
  M[1]  = (1.0f - cos_t) * v.x * v.y - sin_t * v.z;
  M[2]  = (1.0f - cos_t) * v.x * v.z + sin_t * v.y;
  M[4]  = (1.0f - cos_t) * v.y * v.x + sin_t * v.z;
  M[5]  = (1.0f - cos_t) * v.y * v.y + cos_t;
  M[6]  = (1.0f - cos_t) * v.y * v.z - sin_t * v.x;
  M[8]  = (1.0f - cos_t) * v.x * v.z - sin_t * v.y;
  M[9]  = (1.0f - cos_t) * v.z * v.y + sin_t * v.x;
  M[10] = (1.0f - cos_t) * v.z * v.z + cos_t;
}

static float safe_tan(float x) {                                                     /* vector_math.adb:14-22 */
  const float Half_Pi = 0x1.921fb6p+0f;   /* static Ada.Numerics.Pi*0.5 rounded to Float */
  if (fabsf(x) == Half_Pi) return ORC_INFINITY; else return orc_tanf(x);
}

/* ---- sampling helpers, vector_math.adb:175-326 ---- */

static f3 get_perpendicular(f3 a) {                                                  /* :175-200 */
  f3 least;
  float xp = fabsf(a.x), yp = fabsf(a.y), zp = fabsf(a.z);
  if ((xp <= yp + 1.0e-5f) && (xp <= zp + 1.0e-5f)) least = v3(1.0f, 0.0f, 0.0f);
  else if ((yp < xp + 1.0e-5f) && (yp <= zp + 1.0e-5f)) least = v3(0.0f, 1.0f, 0.0f);
  else least = v3(0.0f, 0.0f, 1.0f);
  return normalize(cross(a, least));
}

static f3 cosine_frame_tail(f3 deviation, f3 direction, f3 normal) {                 /* :218-250 and :278-310 (identical) */
  f3 ny = direction;
  f3 nx = get_perpendicular(ny);
  f3 nz = normalize(cross(nx, ny));
  f3 tmp = ny; ny = nz; nz = tmp;
  f3 res = add(add(muls(nx, deviation.x), muls(ny, deviation.y)), muls(nz, deviation.z));
  float invSign = (dot(direction, normal) >= 0.0f) ? 1.0f : -1.0f;
  if (invSign * dot(res, normal) < 0.0f) {
    nx = normalize(cross(normal, direction));
    nz = normalize(cross(nx, ny));
    if (dot(nz, res) < 0.0f) nz = muls(nz, -1.0f);
    res = reflect(muls(res, -1.0f), nz);
    if (dot(res, normal) < 0.0f) res = direction;
  }
  return res;
}

static f3 map_sample_to_cosine_dist(float r1, float r2, f3 direction, f3 normal, float power) { /* :202-252 */
  float e = power;
  float sin_phi = orc_sinf(2.0f * r1 * M_PI_F);
  float cos_phi = orc_cosf(2.0f * r1 * M_PI_F);
  float cos_theta = orc_powf(1.0f - r2, 1.0f / (e + 1.0f));
  float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
  f3 deviation = v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta);
  return cosine_frame_tail(deviation, direction, normal);
}

static f3 map_sample_to_cosine_dist_fixed(float r1, float r2, f3 direction, f3 normal, float power) { /* :255-312 */
  const float TWO_PI = 2.0f * M_PI_F;       /* static 2.0*M_PI, exact doubling */
  float h = sqrtf(1.0f - orc_powf(r1, 2.0f / (power + 1.0f)));
  f3 deviation;
  deviation.x = h * orc_cosf(TWO_PI * r2);
  deviation.y = h * orc_sinf(TWO_PI * r2);
  deviation.z = orc_powf(r1, 1.0f / (power + 1.0f));
  return cosine_frame_tail(deviation, direction, normal);
}

/* ======================================================================================== */
/* lights.adb                                                                               */
/* ======================================================================================== */

typedef struct { f3 pos, dir, intensity; float pdf; } shadow_sample;                 /* lights.ads:15-20 */

static const float L_EPS_DIV = 1.0e-20f;                                             /* lights.adb:39 */

static float pdf_a_to_w(float aPdfA, float aDist, float aCosThere) {                 /* :42-45 */
  return aPdfA * aDist * aDist / max2(aCosThere, L_EPS_DIV);
}

static shadow_sample area_light_sample(const orc_light* l, const rng_ctx* g, f3 p) { /* :56-79 */
  float r1 = rnd(g, 0), r2 = rnd(g, 1);
  shadow_sample res; res.pos = v3(0, 0, 0); res.dir = v3(0, 0, 0); res.intensity = v3(0, 0, 0); res.pdf = 1.0f;
  res.pos.x = l->boxMin[0] + r1 * (l->boxMax[0] - l->boxMin[0]);
  res.pos.y = l->boxMin[1];
  res.pos.z = l->boxMin[2] + r2 * (l->boxMax[2] - l->boxMin[2]);
  res.dir = ld3(l->normal);
  f3 rayDir = sub(res.pos, p);
  float d = length3(rayDir);
  rayDir = muls(rayDir, 1.0f / d);
  float cosTheta = max2(dot(rayDir, muls(ld3(l->normal), -1.0f)), 0.0f);
  res.pdf = pdf_a_to_w(1.0f / l->surfaceArea, d, cosTheta);
  res.intensity = ld3(l->intensity);
  return res;
}

static float area_light_eval_pdf(const orc_light* l, f3 p, f3 rayDir, float hitDist) { /* :81-85 */
  (void)p;
  float cosTheta = max2(dot(rayDir, muls(ld3(l->normal), -1.0f)), 0.0f);
  return pdf_a_to_w(1.0f / l->surfaceArea, hitDist, cosTheta);
}

static void coordinate_system(f3 v1, f3* v2, f3* v3o) {                              /* :107-121 */
  if (fabsf(v1.x) > fabsf(v1.y)) {
    float invLen = 1.0f / sqrtf(v1.x * v1.x + v1.z * v1.z);
    *v2 = v3(-v1.z * invLen, 0.0f, v1.x * invLen);
  } else {
    float invLen = 1.0f / sqrtf(v1.y * v1.y + v1.z * v1.z);
    *v2 = v3(0.0f, v1.z * invLen, -v1.y * invLen);
  }
  *v3o = cross(v1, *v2);
}

static float distance_squared(f3 a, f3 b) { f3 d = sub(b, a); return dot(d, d); }    /* :123-130 */

static f3 uniform_sample_sphere(float u1, float u2) {                                /* :133-142 */
  float z = 1.0f - 2.0f * u1;
  float r = sqrtf(max2(0.0f, 1.0f - z * z));
  float phi = 2.0f * M_PI_F * u2;
  return v3(r * orc_cosf(phi), r * orc_sinf(phi), z);
}

static f3 uniform_sample_cone(float u1, float u2, float costhetamax, f3 x, f3 y, f3 z) { /* :144-151 */
  float costheta = lerpf(u1, costhetamax, 1.0f);
  float sintheta = sqrtf(1.0f - costheta * costheta);
  float phi = u2 * 2.0f * M_PI_F;
  return add(add(muls(x, orc_cosf(phi) * sintheta), muls(y, orc_sinf(phi) * sintheta)), muls(z, costheta));
}

static float uniform_cone_pdf(float cosThetaMax) {                                   /* :153-159 */
  return 1.0f / (2.0f * M_PI_F * (1.0f - cosThetaMax));
}

static void ray_sphere_intersect(f3 rayPos, f3 rayDir, f3 sphPos, float radius, float* rx, float* ry) { /* :161-194 */
  f3 k = sub(rayPos, sphPos);
  float b = dot(k, rayDir);
  float c = dot(k, k) - radius * radius;
  float d = b * b - c;
  if (d >= 0.0f) {
    float sqrtd = sqrtf(d);
    float t1 = -b - sqrtd, t2 = -b + sqrtd;
    *rx = min2(t1, t2); *ry = max2(t1, t2);
  } else { *rx = -ORC_INFINITY; *ry = -ORC_INFINITY; }
}

static float sphere_light_eval_pdf(const orc_light* l, f3 p, f3 rayDir, float hitDist) { /* :242-255 */
  (void)rayDir; (void)hitDist;
  f3 c = ld3(l->center);
  if (distance_squared(p, c) - l->radius * l->radius < 1.0e-4f) return 1.0f / l->surfaceArea;
  float sinThetaMax2 = l->radius * l->radius / distance_squared(p, c);
  float cosThetaMax = sqrtf(max2(0.0f, 1.0f - sinThetaMax2));
  return uniform_cone_pdf(cosThetaMax);
}

static shadow_sample sphere_light_sample(const orc_light* l, const rng_ctx* g, f3 p) { /* :197-240 */
  float u1 = rnd(g, 0), u2 = rnd(g, 1);
  shadow_sample res; res.pos = v3(0, 0, 0); res.dir = v3(0, 0, 0); res.pdf = 1.0f;
  f3 c = ld3(l->center);
  res.intensity = ld3(l->intensity);
  if (distance_squared(p, c) - l->radius * l->radius < 1.0e-4f) {
    res.pos = add(c, muls(uniform_sample_sphere(u1, u2), l->radius));
    res.dir = normalize(sub(res.pos, c));
    return res;
  }
  f3 wc = normalize(sub(c, p)), wcX, wcY;
  coordinate_system(wc, &wcX, &wcY);
  float sinThetaMax2 = l->radius * l->radius / distance_squared(p, c);
  float cosThetaMax = sqrtf(max2(0.0f, 1.0f - sinThetaMax2));
  f3 rdir = uniform_sample_cone(u1, u2, cosThetaMax, wcX, wcY, wc);
  f3 rpos = add(p, muls(rdir, 1.0e-3f));
  float hx, hy, thit;
  ray_sphere_intersect(rpos, rdir, c, l->radius, &hx, &hy);
  if (hx < 0.0f) thit = dot(sub(c, p), normalize(rdir)); else thit = hx;
  res.pos = add(rpos, muls(rdir, thit));
  res.dir = normalize(sub(res.pos, c));
  res.pdf = sphere_light_eval_pdf(l, p, rdir, thit);
  return res;
}

static shadow_sample light_sample(const orc_light* l, const rng_ctx* g, f3 p) {      /* :21-24 dispatch */
  return (l->shape == ORC_LIGHT_RECT) ? area_light_sample(l, g, p) : sphere_light_sample(l, g, p);
}
static float light_eval_pdf(const orc_light* l, f3 p, f3 rayDir, float hitDist) {    /* :26-29 */
  return (l->shape == ORC_LIGHT_RECT) ? area_light_eval_pdf(l, p, rayDir, hitDist) : sphere_light_eval_pdf(l, p, rayDir, hitDist);
}

/* ======================================================================================== */
/* materials.adb                                                                            */
/* ======================================================================================== */

typedef struct { f3 color, direction; float pdf; int pureSpecular; } mat_sample;     /* materials.ads:18-23 */

static const float EPS_DIV = 1.0e-20f;   /* materials.adb:15 */
static const float EPS_COS = 1.0e-6f;    /* materials.adb:16 */

static int total_internal_reflection(float ior, f3 rayDir, f3 normal) {              /* :18-32 */
  float cos_thetai = dot(muls(rayDir, -1.0f), normal);
  float eta = ior;
  if (cos_thetai < 0.0f) eta = 1.0f / eta;
  return (1.0f - (1.0f - cos_thetai * cos_thetai) / (eta * eta)) < 0.0f;
}

static f3 refract_dir(float ior, f3 rayDir, f3 normal) {                             /* :34-52 */
  f3 n = normal;
  f3 wo = muls(rayDir, -1.0f);
  float cos_thetai = dot(muls(rayDir, -1.0f), normal);
  float eta = ior;
  if (cos_thetai < 0.0f) { eta = 1.0f / eta; cos_thetai = -cos_thetai; n = muls(n, -1.0f); }
  float cos_theta2 = sqrtf(1.0f - (1.0f - cos_thetai * cos_thetai) / (eta * eta));
  return normalize(sub(muls(muls(wo, -1.0f), 1.0f / eta), muls(n, cos_theta2 - cos_thetai / eta)));
}

static float fresnel_dielectric(float cosTheta1, float cosTheta2, float etaExt, float etaInt) { /* :70-76 */
  float Rs = (etaExt * cosTheta1 - etaInt * cosTheta2) / (etaExt * cosTheta1 + etaInt * cosTheta2);
  float Rp = (etaInt * cosTheta1 - etaExt * cosTheta2) / (etaInt * cosTheta1 + etaExt * cosTheta2);
  return (Rs * Rs + Rp * Rp) / 2.0f;
}

static float fresnel(float cosTheta1, float a_etaExt, float a_etaInt) {              /* :79-99 */
  float etaExt = a_etaExt, etaInt = a_etaInt;
  if (cosTheta1 < 0.0f) { float tmp = etaExt; etaExt = etaInt; etaInt = tmp; }
  float sinTheta2 = (etaExt / etaInt) * sqrtf(max2(0.0f, 1.0f - cosTheta1 * cosTheta1));
  if (sinTheta2 > 1.0f) return 1.0f;
  float cosTheta2 = sqrtf(max2(0.0f, 1.0f - sinTheta2 * sinTheta2));
  return fresnel_dielectric(fabsf(cosTheta1), cosTheta2, etaInt, etaExt);
}

static int mat_is_light(const orc_material* m) { return m->type == ORC_MAT_LIGHT; }  /* :142,182,232,270,348 */

static f3 mat_emittance(const orc_scene* s, const orc_material* m) {                 /* :147-156 */
  if (m->type != ORC_MAT_LIGHT) return v3(0, 0, 0);
  if (m->light < 0 || m->light >= s->n_lights) return v3(0, 0, 0);
  return ld3(s->lights[m->light].intensity);
}

static mat_sample mat_sample_and_eval(const orc_material* m, const rng_ctx* g, f3 ray_dir, f3 normal) {
  mat_sample r;
  switch (m->type) {
    case ORC_MAT_LAMBERT: {                                                          /* :197-215 */
      float r1 = rnd(g, 2), r2 = rnd(g, 3);                                          /* vector_math.adb:314-319 */
      f3 newDir = map_sample_to_cosine_dist(r1, r2, normal, normal, 1.0f);
      float cosTheta = dot(newDir, normal);
      float pdf = fabsf(cosTheta) * INV_PI_F;
      f3 color = muls(ld3(m->p), INV_PI_F);
      if (cosTheta < EPS_COS) color = v3(0, 0, 0);
      r.color = color; r.direction = newDir; r.pdf = pdf; r.pureSpecular = 0;
      return r;
    }
    case ORC_MAT_MIRROR: {                                                           /* :247-254 */
      f3 nextDir = reflect(ray_dir, normal);
      float cosThetaDiv = 1.0f / max2(dot(nextDir, normal), EPS_DIV);
      r.color = muls(ld3(m->p), cosThetaDiv); r.direction = nextDir; r.pdf = 1.0f; r.pureSpecular = 1;
      return r;
    }
    case ORC_MAT_GLASS: {                                                            /* :295-331 */
      f3 refl = ld3(m->p), trans = ld3(m->p + 3);
      float ior = m->p[6];
      {                                                                              /* ApplyFresnel :285-293 */
        float f = fresnel(dot(ray_dir, normal), ior, 1.0f);
        refl = muls(refl, f);
        trans = muls(trans, 1.0f - f);
      }
      float ksitrans = length3(trans) / (length3(refl) + length3(trans));
      float ksirefl  = length3(refl) / (length3(refl) + length3(trans));
      float ksi = rnd(g, 2);
      f3 nextDirection, bxdf;
      if (ksi > ksitrans) {
        nextDirection = reflect(ray_dir, normal);
        bxdf = muls(refl, 1.0f / ksirefl);
      } else {
        bxdf = muls(trans, 1.0f / ksitrans);
        if (!total_internal_reflection(ior, ray_dir, normal)) nextDirection = refract_dir(ior, ray_dir, normal);
        else nextDirection = reflect(ray_dir, normal);
      }
      float cosThetaDiv = 1.0f / max2(fabsf(dot(nextDirection, normal)), EPS_DIV);
      r.color = muls(bxdf, cosThetaDiv); r.direction = nextDirection; r.pdf = 1.0f; r.pureSpecular = 1;
      return r;
    }
    case ORC_MAT_PHONG: {                                                            /* :363-387 */
      float cosPower = m->p[3];
      f3 rr = reflect(ray_dir, normal);
      float r1 = rnd(g, 2), r2 = rnd(g, 3);                                          /* vector_math.adb:321-326 */
      f3 nextDir = map_sample_to_cosine_dist_fixed(r1, r2, rr, normal, cosPower);
      float cosTheta = clampf(dot(nextDir, rr), 0.0f, 0x1.921eaep+0f /* M_PI*0.499995 */);
      f3 color = muls(muls(muls(muls(ld3(m->p), cosPower + 2.0f), 0.5f), INV_PI_F), orc_powf(cosTheta, cosPower));
      float pdf = orc_powf(cosTheta, cosPower) * (cosPower + 1.0f) * (0.5f * INV_PI_F);
      float cosThetaGeo = dot(nextDir, normal);
      float cosThetaDiv = 1.0f / max2(fabsf(cosThetaGeo), EPS_DIV);
      if (cosThetaGeo < EPS_COS) color = v3(0, 0, 0);
      r.color = muls(color, cosThetaDiv); r.direction = nextDir; r.pdf = pdf; r.pureSpecular = 0;
      return r;
    }
    default: {                                                                       /* MaterialLight :163-166 */
      r.color = v3(0, 0, 0); r.direction = v3(0, 0, 0); r.pdf = 1.0f; r.pureSpecular = 0;
      return r;
    }
  }
}

static f3 mat_eval_bxdf(const orc_material* m, f3 l, f3 v, f3 n) {
  switch (m->type) {
    case ORC_MAT_LAMBERT: return muls(ld3(m->p), INV_PI_F);                          /* :217-220 */
    case ORC_MAT_PHONG: {                                                            /* :389-398 */
      float cosPower = m->p[3];
      f3 r = reflect(muls(v, -1.0f), n);
      float cosTheta = clampf(dot(l, r), 0.0f, 0x1.921eaep+0f);
      float cosThetaDiv = 1.0f / max2(dot(l, n), EPS_DIV);
      return muls(muls(muls(muls(muls(ld3(m->p), cosPower + 2.0f), 0.5f), INV_PI_F), orc_powf(cosTheta, cosPower)), cosThetaDiv);
    }
    default: return v3(0, 0, 0);                                                     /* :168-171, 256-259, 333-336 */
  }
}

static float mat_eval_pdf(const orc_material* m, f3 l, f3 v, f3 n) {
  switch (m->type) {
    case ORC_MAT_LAMBERT: { float cosTheta = max2(dot(n, l), 0.0f); return cosTheta * INV_PI_F; } /* :222-226 */
    case ORC_MAT_PHONG: {                                                            /* :401-410 */
      float cosPower = m->p[3];
      f3 r = reflect(muls(v, -1.0f), n);
      float cosTheta = clampf(dot(l, r), 0.0f, 0x1.921eaep+0f);
      return orc_powf(cosTheta, cosPower) * (cosPower + 1.0f) * (0.5f * INV_PI_F);
    }
    default: return 1.0f;                                                            /* :173-176, 261-264, 338-341 */
  }
}

/* ---- per-function known-answer entry points (tests/test_oracle_kat.py compares them with an independent numpy-float32
   transcription of lights.adb / materials.adb; the uniforms are the ones the render draws for (seed, pixel, sample, bounce)) ---- */
void orc_kat_light_sample(const orc_light* l, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, const float p[3], float out10[10]) {
  rng_ctx g = { seed, pixel, sample, bounce };
  shadow_sample r = light_sample(l, &g, ld3(p));
  out10[0] = r.pos.x; out10[1] = r.pos.y; out10[2] = r.pos.z; out10[3] = r.dir.x; out10[4] = r.dir.y; out10[5] = r.dir.z;
  out10[6] = r.intensity.x; out10[7] = r.intensity.y; out10[8] = r.intensity.z; out10[9] = r.pdf;
}
float orc_kat_light_eval_pdf(const orc_light* l, const float p[3], const float ray_dir[3], float hit_dist) {
  return light_eval_pdf(l, ld3(p), ld3(ray_dir), hit_dist);
}
void orc_kat_mat_sample(const orc_material* m, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, const float ray_dir[3],
                        const float normal[3], float out8[8]) {
  rng_ctx g = { seed, pixel, sample, bounce };
  mat_sample r = mat_sample_and_eval(m, &g, ld3(ray_dir), ld3(normal));
  out8[0] = r.color.x; out8[1] = r.color.y; out8[2] = r.color.z; out8[3] = r.direction.x; out8[4] = r.direction.y; out8[5] = r.direction.z;
  out8[6] = r.pdf; out8[7] = (float)r.pureSpecular;
}
void orc_kat_mat_eval(const orc_material* m, const float l[3], const float v[3], const float n[3], float out4[4]) {
  f3 b = mat_eval_bxdf(m, ld3(l), ld3(v), ld3(n));
  out4[0] = b.x; out4[1] = b.y; out4[2] = b.z; out4[3] = mat_eval_pdf(m, ld3(l), ld3(v), ld3(n));
}

/* ======================================================================================== */
/* geometry.adb                                                                             */
/* ======================================================================================== */

static inline int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }

typedef struct { f3 origin, direction; int x, y; } ray_t;                            /* geometry.ads:15-19 */

enum { PRIM_PLANE = 0, PRIM_SPHERE = 1, PRIM_TRIANGLE = 2, PRIM_QUAD = 3 };          /* geometry.ads:55 */

typedef struct {                                                                      /* geometry.ads:57-67 */
  int prim_type; int is_hit; float t; f3 normal; int mat /* -1 = null */; int matId; float tx, ty; int prim_index;
} hit_t;

typedef struct { int is_hit; float tmin, tmax, u, v; } lite_hit;                      /* geometry.ads:106-111 */

static hit_t null_hit(void) {                                                         /* geometry.ads:126-135 */
  hit_t h; h.prim_type = PRIM_PLANE; h.prim_index = -1; h.is_hit = 0; h.t = ORC_INFINITY; h.mat = -1; h.matId = 0;
  h.tx = 0.0f; h.ty = 0.0f; h.normal = v3(0.0f, 1.0f, 0.0f);
  return h;
}

static hit_t intersect_all_spheres(const ray_t* r, const orc_sphere* sph, int n) {    /* geometry.adb:48-115 */
  float min_t = ORC_INFINITY;
  int min_i = 0;
  f3 finalNormal = v3(0.0f, 1.0f, 0.0f);
  for (int i = 0; i < n; ++i) {
    f3 k = sub(r->origin, ld3(sph[i].pos));
    float b = dot(k, r->direction);
    float c = dot(k, k) - sph[i].r * sph[i].r;
    float d = b * b - c;
    if (d >= 0.0f) {
      float sqrtd = sqrtf(d);
      float t1 = -b - sqrtd, t2 = -b + sqrtd;
      if (t1 > 0.0f && t1 < min_t) { min_t = t1; min_i = i; }
      else if (t2 > 0.0f && t2 < min_t) { min_t = t2; min_i = i; }
    }
  }
  int is_hit = (min_t > 0.0f && min_t < ORC_INFINITY);
  if (!is_hit) min_t = 1.0f;
  else finalNormal = normalize(sub(add(r->origin, muls(r->direction, min_t)), ld3(sph[min_i].pos)));
  hit_t h; h.prim_type = PRIM_SPHERE; h.prim_index = min_i; h.is_hit = is_hit; h.t = min_t;
  h.mat = (n > 0) ? sph[min_i].mat : -1; h.matId = 0; h.normal = finalNormal; h.tx = 0.0f; h.ty = 0.0f;
  return h;
}

static hit_t intersect_flat_light(const ray_t* r, const orc_light* lg, int lightIndex) { /* :118-143 */
  float inv_dir_y = 1.0f / r->direction.y;
  float tmin = (lg->boxMax[1] - r->origin.y) * inv_dir_y;
  f3 hp = add(r->origin, muls(r->direction, tmin));
  int is_hit = (hp.x > lg->boxMin[0]) && (hp.x < lg->boxMax[0]) && (hp.z > lg->boxMin[2]) && (hp.z < lg->boxMax[2]) && (tmin >= 0.0f);
  hit_t h; h.prim_type = PRIM_QUAD; h.prim_index = lightIndex; h.is_hit = is_hit; h.t = tmin; h.mat = lg->mat; h.matId = 0;
  h.normal = v3(0.0f, -1.0f, 0.0f); h.tx = 0.0f; h.ty = 0.0f;
  return h;
}

static lite_hit intersect_box(const ray_t* r, const float* bmin, const float* bmax) { /* :146-191 */
  float inv_dir_x = 1.0f / r->direction.x, inv_dir_y = 1.0f / r->direction.y, inv_dir_z = 1.0f / r->direction.z;
  float lo  = (bmax[0] - r->origin.x) * inv_dir_x, hi  = (bmin[0] - r->origin.x) * inv_dir_x;
  float lo1 = (bmax[1] - r->origin.y) * inv_dir_y, hi1 = (bmin[1] - r->origin.y) * inv_dir_y;
  float lo2 = (bmax[2] - r->origin.z) * inv_dir_z, hi2 = (bmin[2] - r->origin.z) * inv_dir_z;
  float tmin = min2(lo, hi), tmax = max2(lo, hi);
  tmin = max2(tmin, min2(lo1, hi1)); tmax = min2(tmax, max2(lo1, hi1));
  tmin = max2(tmin, min2(lo2, hi2)); tmax = min2(tmax, max2(lo2, hi2));
  lite_hit res; res.u = 0.0f; res.v = 0.0f;
  res.tmin = tmin; res.tmax = tmax; res.is_hit = (tmax > 0.0f) && (tmin <= tmax);
  return res;
}

static hit_t intersect_cornell_box(const ray_t* r, const orc_scene* s) {              /* :193-229 */
  const float eps = 1.0e-5f;
  int planeId = 0;
  lite_hit tmpHit = intersect_box(r, s->cb_min, s->cb_max);
  if (tmpHit.is_hit) {
    f3 p = add(r->origin, muls(r->direction, tmpHit.tmax));
    if (fabsf(p.x - s->cb_min[0]) < eps) planeId = 0;
    if (fabsf(p.x - s->cb_max[0]) < eps) planeId = 1;
    if (fabsf(p.y - s->cb_min[1]) < eps) planeId = 2;
    if (fabsf(p.y - s->cb_max[1]) < eps) planeId = 3;
    if (fabsf(p.z - s->cb_min[2]) < eps) planeId = 4;
    if (fabsf(p.z - s->cb_max[2]) < eps) planeId = 5;
    hit_t h; h.prim_type = PRIM_PLANE; h.prim_index = planeId; h.is_hit = !(planeId == 5); h.t = tmpHit.tmax;
    h.mat = -1; h.matId = s->cb_mat[planeId]; h.normal = ld3(s->cb_nrm[planeId]); h.tx = 0.0f; h.ty = 0.0f;
    return h;
  }
  return null_hit();
}

static lite_hit intersect_triangle(const ray_t* r, f3 A, f3 B, f3 C, float t_min, float t_max) { /* :231-263 */
  const float epsilonDiv = 1.0e-25f;
  lite_hit res; res.is_hit = 0; res.tmin = 0.0f; res.tmax = 0.0f; res.u = 0.0f; res.v = 0.0f;
  f3 edge1 = sub(B, A), edge2 = sub(C, A);
  f3 pvec = cross(r->direction, edge2);
  f3 tvec = sub(r->origin, A);
  f3 qvec = cross(tvec, edge1);
  float invDet = 1.0f / max2(dot(edge1, pvec), epsilonDiv);
  float v = dot(tvec, pvec) * invDet;
  float u = dot(qvec, r->direction) * invDet;
  float t = dot(edge2, qvec) * invDet;
  if (v > 0.0f && u > 0.0f && u + v < 1.0f && t > t_min && t < t_max) {
    res.u = u; res.v = v; res.tmin = t; res.tmax = t + 1.0e-6f; res.is_hit = 1;
  }
  return res;
}

static hit_t mesh_hit_record(const orc_mesh* m, lite_hit nh, int triId, int matId) {   /* :298-315 */
  const int* tri = m->idx + 3 * triId;
  float w = 1.0f - nh.u - nh.v;
  f3 inorm = add(add(muls(ld3(m->nrm + 3 * tri[0]), w), muls(ld3(m->nrm + 3 * tri[1]), nh.v)), muls(ld3(m->nrm + 3 * tri[2]), nh.u));
  float itx = w * m->uv[2 * tri[0] + 0] + nh.v * m->uv[2 * tri[1] + 0] + nh.u * m->uv[2 * tri[2] + 0];
  float ity = w * m->uv[2 * tri[0] + 1] + nh.v * m->uv[2 * tri[1] + 1] + nh.u * m->uv[2 * tri[2] + 1];
  hit_t h; h.prim_type = PRIM_TRIANGLE; h.prim_index = triId; h.is_hit = nh.is_hit; h.t = nh.tmin; h.mat = -1;
  h.matId = matId; h.normal = inorm; h.tx = itx; h.ty = ity;
  return h;
}

static hit_t intersect_mesh_bf(const ray_t* r, const orc_mesh* m, uint64_t* tri_tests) { /* :266-323 */
  lite_hit tmpHit = intersect_box(r, m->bbmin, m->bbmax);
  if (tmpHit.is_hit) {
    lite_hit nearestHit; nearestHit.tmin = 0.0f; nearestHit.tmax = 1000000.0f; nearestHit.is_hit = 0; nearestHit.u = 0.0f; nearestHit.v = 0.0f;
    int nearestTriId = 0;
    for (int i = 0; i < m->ntris; ++i) {
      const int* tri = m->idx + 3 * i;
      tmpHit = intersect_triangle(r, ld3(m->pos + 3 * tri[0]), ld3(m->pos + 3 * tri[1]), ld3(m->pos + 3 * tri[2]), nearestHit.tmin, nearestHit.tmax);
      if (tmpHit.is_hit) { nearestHit = tmpHit; nearestTriId = i; }
    }
    *tri_tests += (uint64_t)m->ntris;
    return mesh_hit_record(m, nearestHit, nearestTriId, 2);
  }
  return null_hit();
}

/* Extension: true closest hit over all triangles -- Embree semantics at the gcore seam
 * (embree_connect.cpp:196-238) with the reference's Moeller-Trumbore arithmetic (geometry.adb:231-263)
 * and window (0, 1e6) (geometry.adb:277-278).  Ties: lowest triangle index.  matId from material_ids.
 * Hit record for the shader uses the interpolated vertex normal exactly like geometry.adb:301.       */
static void bvh_walk_one(const float* nodes, const float* tris, int width, const ray_t* r, float tfar,
                         lite_hit* best, int32_t* best_prim, uint64_t* counters /* box, tri, node, leaf */);

static hit_t intersect_mesh_closest(const ray_t* r, const orc_mesh* m, uint64_t* tri_tests) {
  if (m->bvh_nodes && m->bvh_tris) {
    lite_hit bh; int32_t prim = -1; uint64_t c[4] = { 0, 0, 0, 0 };
    bvh_walk_one(m->bvh_nodes, m->bvh_tris, m->bvh_width ? m->bvh_width : 8, r, ORC_INFINITY, &bh, &prim, c);
    *tri_tests += c[1];
    if (prim < 0) return null_hit();
    return mesh_hit_record(m, bh, prim, m->matid[prim]);
  }
  lite_hit best; best.is_hit = 0; best.tmin = 0.0f; best.tmax = 0.0f; best.u = 0.0f; best.v = 0.0f;
  int bestId = 0;
  float best_t = 1000000.0f;
  for (int i = 0; i < m->ntris; ++i) {
    const int* tri = m->idx + 3 * i;
    lite_hit h = intersect_triangle(r, ld3(m->pos + 3 * tri[0]), ld3(m->pos + 3 * tri[1]), ld3(m->pos + 3 * tri[2]), 0.0f, 1000000.0f);
    if (h.is_hit && h.tmin < best_t) { best = h; best_t = h.tmin; bestId = i; }
  }
  *tri_tests += (uint64_t)m->ntris;
  if (!best.is_hit) return null_hit();
  return mesh_hit_record(m, best, bestId, m->matid[bestId]);
}

/* ======================================================================================== */
/* scene.adb:56-86  Find_Closest_Hit                                                        */
/* ======================================================================================== */

typedef struct { uint64_t rays, tri_tests; } local_counters;

static hit_t find_closest_hit(const orc_scene* s, const ray_t* r, local_counters* lc) {
  hit_t hits[8];
  int n = 0;
  lc->rays += 1;
  hits[n++] = intersect_all_spheres(r, s->spheres, s->n_spheres);                   /* hits(1) */
  if (s->has_cornell) hits[n++] = intersect_cornell_box(r, s);                      /* hits(2) */
  for (int i = 0; i < s->n_lights && n < 6; ++i)                                     /* hits(3): rect light(s) only */
    if (s->lights[i].shape == ORC_LIGHT_RECT) hits[n++] = intersect_flat_light(r, &s->lights[i], i);
  for (int pass = 0; pass < 2; ++pass)                                               /* hits(4): BF mesh, then (extension) closest mesh */
    for (int i = 0; i < s->n_meshes; ++i)
      if (s->meshes[i].mode == (pass == 0 ? ORC_MESH_REFERENCE_BF : ORC_MESH_CLOSEST))
        hits[n++] = (pass == 0) ? intersect_mesh_bf(r, &s->meshes[i], &lc->tri_tests) : intersect_mesh_closest(r, &s->meshes[i], &lc->tri_tests);
  int nearest = 0;
  float nearestDist = ORC_INFINITY;
  for (int i = 0; i < n; ++i)
    if (hits[i].is_hit && hits[i].t < nearestDist) { nearest = i; nearestDist = hits[i].t; }
  if (hits[nearest].mat < 0) hits[nearest].mat = hits[nearest].matId;              /* scene.adb:80-82 */
  return hits[nearest];
}

/* ======================================================================================== */
/* ray_tracer.adb                                                                           */
/* ======================================================================================== */

static const float G_EPSILON     = 1.0e-5f;    /* ray_tracer.ads:29 */
static const float G_EPSILON_DIV = 1.0e-20f;   /* ray_tracer.ads:30 */

static f3 eye_ray_direction(const orc_params* p, int x, int y, float ox, float oy) { /* ray_tracer.adb:61-69, 72-97 */
  const float fov = 0x1.921fb6p+0f;             /* static Pi/2.0 */
  f3 res;
  res.x = (float)x + ox - ((float)p->width / 2.0f);
  res.y = (float)y + oy - ((float)p->height / 2.0f);
  res.z = -(float)p->width / safe_tan(fov / 2.0f);
  return normalize(res);
}

static int compute_shadow(const orc_scene* s, f3 hit_pos, f3 lpos, local_counters* lc) { /* ray_tracer.adb:100-132 */
  float epsilon = max3(fabsf(hit_pos.x), fabsf(hit_pos.y), fabsf(hit_pos.z)) * 0.000000001f;
  epsilon = max2(epsilon, 1.0e-30f);
  ray_t sr; sr.x = 0; sr.y = 0;
  sr.direction = normalize(sub(lpos, hit_pos));
  sr.origin = add(hit_pos, muls(sr.direction, epsilon));
  hit_t h = find_closest_hit(s, &sr, lc);
  float maxDist = length3(sub(hit_pos, lpos));
  float epsilon2 = max2(maxDist * 0.000001f, 1.0e-30f);
  return h.is_hit && (h.t < maxDist - epsilon2 && h.t > 10.0f * epsilon);
}

/* light selection: the reference has exactly one light (scene.adb:45-48).  Extension for the
 * synthetic multi-light configs: uniform choice with one extra draw, pdf scaled by 1/n.      */
static int pick_light(const orc_scene* s, const rng_ctx* g, float* selPdf) {
  int n = s->n_lights;
  *selPdf = 1.0f / (float)n;
  if (n <= 1) return 0;
  int i = (int)(rnd(g, 4) * (float)n);
  return (i > n - 1) ? n - 1 : i;
}

typedef struct { const orc_scene* s; const orc_params* p; rng_ctx g; local_counters* lc; } trace_ctx;

static const mat_sample START_SAMPLE = { { 0, 0, 0 }, { 0, 0, 0 }, 1.0f, 1 };        /* materials.ads:25 */

static f3 path_trace_stupid(trace_ctx* c, ray_t r, mat_sample prev, int level) {     /* integrators.adb:82-126 */
  (void)prev;
  if (level == 0) return v3(0, 0, 0);
  hit_t h = find_closest_hit(c->s, &r, c->lc);
  if (!h.is_hit) return v3(0, 0, 0);
  const orc_material* m = &c->s->materials[h.mat];
  if (mat_is_light(m)) {
    if (dot(muls(r.direction, -1.0f), h.normal) < 0.0f) return v3(0, 0, 0);
    return mat_emittance(c->s, m);
  }
  c->g.bounce = (uint32_t)(c->p->max_depth - level);
  mat_sample ms = mat_sample_and_eval(m, &c->g, r.direction, h.normal);
  f3 bxdfVal = muls(ms.color, 1.0f / max2(ms.pdf, G_EPSILON_DIV));
  float cosTheta = dot(ms.direction, h.normal);
  ray_t next = r;
  next.origin = add(r.origin, muls(r.direction, h.t));
  next.direction = ms.direction;
  next.origin = add(next.origin, muls(muls(h.normal, signf_ada(cosTheta)), G_EPSILON));
  return mulv(muls(bxdfVal, fabsf(cosTheta)), path_trace_stupid(c, next, ms, level - 1));
}

static f3 path_trace_shadow(trace_ctx* c, ray_t r, mat_sample prev, int level) {     /* integrators.adb:136-193 */
  (void)prev;
  f3 explicitColor = v3(0, 0, 0);
  if (level == 0) return v3(0, 0, 0);
  hit_t h = find_closest_hit(c->s, &r, c->lc);
  if (!h.is_hit) return v3(0, 0, 0);
  const orc_material* m = &c->s->materials[h.mat];
  if (mat_is_light(m)) return v3(0, 0, 0);
  c->g.bounce = (uint32_t)(c->p->max_depth - level);
  rng_ctx g = c->g;
  {
    f3 hpos = add(r.origin, muls(r.direction, h.t));
    float selPdf; int li = pick_light(c->s, &g, &selPdf);
    shadow_sample lsam = light_sample(&c->s->lights[li], &g, hpos);
    lsam.pdf = lsam.pdf * selPdf;
    f3 sdir = normalize(sub(lsam.pos, hpos));
    if (!compute_shadow(c->s, hpos, lsam.pos, c->lc)) {
      f3 bxdfVal = mat_eval_bxdf(m, sdir, muls(r.direction, -1.0f), h.normal);
      float cosTheta1 = max2(dot(sdir, h.normal), 0.0f);
      explicitColor = muls(mulv(lsam.intensity, muls(bxdfVal, cosTheta1)), 1.0f / max2(lsam.pdf, G_EPSILON_DIV));
    }
  }
  mat_sample ms = mat_sample_and_eval(m, &g, r.direction, h.normal);
  f3 bxdfVal = muls(ms.color, 1.0f / max2(ms.pdf, G_EPSILON_DIV));
  float cosTheta = dot(ms.direction, h.normal);
  ray_t next = r;
  next.origin = add(r.origin, muls(r.direction, h.t));
  next.direction = ms.direction;
  next.origin = add(next.origin, muls(muls(h.normal, signf_ada(cosTheta)), G_EPSILON));
  return add(explicitColor, mulv(muls(bxdfVal, fabsf(cosTheta)), path_trace_shadow(c, next, ms, level - 1)));
}

static f3 path_trace_mis(trace_ctx* c, ray_t r, mat_sample prev, int level) {        /* integrators.adb:203-301 */
  f3 explicitColor = v3(0, 0, 0);
  if (level == 0) return v3(0, 0, 0);
  hit_t h = find_closest_hit(c->s, &r, c->lc);
  if (!h.is_hit) return v3(0, 0, 0);
  const orc_material* m = &c->s->materials[h.mat];
  if (mat_is_light(m)) {
    if (dot(muls(r.direction, -1.0f), h.normal) < 0.0f) return v3(0, 0, 0);
    {
      float selPdf = 1.0f / (float)c->s->n_lights;
      float lgtPdf = light_eval_pdf(&c->s->lights[m->light], r.origin, r.direction, h.t) * selPdf;
      float bsdfPdf = prev.pdf;
      float misWeight;
      if (prev.pureSpecular) misWeight = 1.0f;
      else misWeight = bsdfPdf * bsdfPdf / (lgtPdf * lgtPdf + bsdfPdf * bsdfPdf);
      return muls(mat_emittance(c->s, m), misWeight);
    }
  }
  c->g.bounce = (uint32_t)(c->p->max_depth - level);
  rng_ctx g = c->g;
  {
    f3 hpos = add(r.origin, muls(r.direction, h.t));
    float selPdf; int li = pick_light(c->s, &g, &selPdf);
    shadow_sample lsam = light_sample(&c->s->lights[li], &g, hpos);
    f3 sdir = normalize(sub(lsam.pos, hpos));
    float lgtPdf = lsam.pdf * selPdf;
    if (!compute_shadow(c->s, hpos, lsam.pos, c->lc)) {
      float bsdfPdf = mat_eval_pdf(m, sdir, muls(r.direction, -1.0f), h.normal);
      f3 bxdfVal = mat_eval_bxdf(m, sdir, muls(r.direction, -1.0f), h.normal);
      float cosTheta1 = max2(dot(sdir, h.normal), 0.0f);
      float misWeight = lgtPdf * lgtPdf / (lgtPdf * lgtPdf + bsdfPdf * bsdfPdf);
      explicitColor = muls(mulv(muls(lsam.intensity, 1.0f / max2(lgtPdf, G_EPSILON_DIV)), muls(bxdfVal, cosTheta1)), misWeight);
    } else explicitColor = v3(0, 0, 0);
  }
  mat_sample ms = mat_sample_and_eval(m, &g, r.direction, h.normal);
  f3 bxdfVal = muls(ms.color, 1.0f / max2(ms.pdf, G_EPSILON_DIV));
  float cosTheta = dot(ms.direction, h.normal);
  ray_t next = r;
  next.origin = add(r.origin, muls(r.direction, h.t));
  next.direction = ms.direction;
  next.origin = add(next.origin, muls(muls(h.normal, signf_ada(cosTheta)), G_EPSILON));
  return add(explicitColor, mulv(muls(bxdfVal, fabsf(cosTheta)), path_trace_mis(c, next, ms, level - 1)));
}

static f3 path_trace(trace_ctx* c, ray_t r) {                                        /* ray_tracer.adb:151-157 dispatch */
  switch (c->p->render_type) {
    case ORC_PT_STUPID: return path_trace_stupid(c, r, START_SAMPLE, c->p->max_depth);
    case ORC_PT_SHADOW: return path_trace_shadow(c, r, START_SAMPLE, c->p->max_depth);
    default:            return path_trace_mis(c, r, START_SAMPLE, c->p->max_depth);
  }
}

/* sub-pixel offsets: ray_tracer.adb:76-90 (AA) and :65-66 (no AA) */
static const float AA_OFF[4][2] = { { 0x1.555556p-2f, 0x1.555556p-2f }, { 0x1.555556p-2f, 0x1.555556p-1f },
                                    { 0x1.555556p-1f, 0x1.555556p-2f }, { 0x1.555556p-1f, 0x1.555556p-1f } };

static f3 camera_sample(const orc_scene* s, const orc_params* p, int x, int y, uint32_t sample_index, local_counters* lc) {
  ray_t r; r.x = x; r.y = y;
  r.origin = ld3(s->cam_pos);
  float ox = 0.5f, oy = 0.5f;
  if (p->aa_on) { ox = AA_OFF[sample_index & 3][0]; oy = AA_OFF[sample_index & 3][1]; }
  f3 d = eye_ray_direction(p, x, y, ox, oy);
  r.direction = normalize(mat_mul_v(s->cam_matrix, d));                               /* integrators.adb:46, 57-58 */
  trace_ctx c; c.s = s; c.p = p; c.lc = lc;
  c.g.seed = p->seed; c.g.pixel = (uint32_t)(y * p->width + x); c.g.sample = sample_index; c.g.bounce = 0;
  return path_trace(&c, r);
}

void orc_sample_radiance(const orc_scene* scn, const orc_params* prm, int32_t x, int32_t y, uint32_t sample_index, float out_rgb[3]) {
  local_counters lc = { 0, 0 };
  f3 c = camera_sample(scn, prm, x, y, sample_index, &lc);
  out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
}

/* One pixel of Render_Pass: Threads_Num tasks x DoPass (integrators.adb:25-71), accumulated in task order
 * t = 0..vthreads-1 (the reference's order under GNAT.Task_Lock is scheduling dependent).  `a` = colBuff(x,y). */
static void pass_pixel(const orc_scene* scn, const orc_params* prm, int x, int y, uint32_t base, float* a, local_counters* lc) {
  for (int t = 0; t < prm->vthreads; ++t) {
    f3 color;
    if (prm->aa_on) {
      color = ld3(prm->background);                                               /* integrators.adb:42 */
      for (int i = 0; i < 4; ++i)
        color = add(color, camera_sample(scn, prm, x, y, base + (uint32_t)(t * 4 + i), lc)); /* :47 */
    } else {
      color = camera_sample(scn, prm, x, y, base + (uint32_t)t, lc);              /* :60 */
    }
    f3 cb = add(color, ld3(a));                                                   /* :51 / :63 */
    a[0] = cb.x; a[1] = cb.y; a[2] = cb.z;
  }
}

void orc_render_pass(const orc_scene* scn, const orc_params* prm, float* accum, int32_t* spp, orc_counters* cnt) {
  const int W = prm->width, H = prm->height;
  const int per = prm->aa_on ? 4 : 1;
  const uint32_t base = (uint32_t)*spp;
  uint64_t rays = 0, tris = 0;
#ifdef _OPENMP
  if (prm->nthreads > 0) omp_set_num_threads(prm->nthreads);
#endif
  const int64_t npx = (int64_t)W * H, nblk = (npx + 63) / 64;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, tris)
  for (int64_t blk = 0; blk < nblk; ++blk) {
    local_counters lc = { 0, 0 };
    const int64_t p1 = (blk + 1) * 64 < npx ? (blk + 1) * 64 : npx;
    for (int64_t pi = blk * 64; pi < p1; ++pi) {
      const int x = (int)(pi % W), y = (int)(pi / W);
      pass_pixel(scn, prm, x, y, base, accum + 3 * ((size_t)y * W + x), &lc);
    }
    rays += lc.rays; tris += lc.tri_tests;
  }
  *spp += prm->vthreads * per;                                                        /* ray_tracer.adb:168-175 */
  if (cnt) { cnt->rays += rays; cnt->tri_tests += tris; cnt->samples += (uint64_t)W * H * prm->vthreads * per; }
}

/* The same pass for a LIST of pixels of the frame (tests: a BASELINE configuration at its full frame size and sample count is out of
 * reach as a whole frame, a few hundred of its pixels are not): accum[3k..] = colBuff(xs[k], ys[k]), cumulative like the frame's.    */
void orc_render_pixels(const orc_scene* scn, const orc_params* prm, const int32_t* xs, const int32_t* ys, int64_t n, float* accum, int32_t spp0,
                       orc_counters* cnt) {
  const int per = prm->aa_on ? 4 : 1;
  uint64_t rays = 0, tris = 0;
#ifdef _OPENMP
  if (prm->nthreads > 0) omp_set_num_threads(prm->nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, tris)
  for (int64_t k = 0; k < n; ++k) {
    local_counters lc = { 0, 0 };
    pass_pixel(scn, prm, xs[k], ys[k], (uint32_t)spp0, accum + 3 * k, &lc);
    rays += lc.rays; tris += lc.tri_tests;
  }
  if (cnt) { cnt->rays += rays; cnt->tri_tests += tris; cnt->samples += (uint64_t)n * prm->vthreads * per; }
}

/* The same pass organised the way the reference runs it (ray_tracer.adb:142-194, 264-277): Threads_Num tasks, each renders the WHOLE
 * frame once (DoPass) with its own samples; here every task adds into a private frame instead of g_accBuff under GNAT.Task_Lock
 * (integrators.adb:50-52), and the private frames are added in task order afterwards -- bit-identical to orc_render_pass.
 * `nthreads` OS threads share the vthreads tasks.  Used as the reference-faithful CPU baseline (bench.py, cpu_baseline.mode_a).   */
int orc_render_pass_tasks(const orc_scene* scn, const orc_params* prm, float* accum, int32_t* spp, orc_counters* cnt) {
  const int W = prm->width, H = prm->height, T = prm->vthreads;
  const int per = prm->aa_on ? 4 : 1;
  const uint32_t base = (uint32_t)*spp;
  const size_t npx = (size_t)W * H;
  float* priv = (float*)malloc((size_t)T * npx * 3 * sizeof(float));
  if (!priv) return 1;
  uint64_t rays = 0, tris = 0;
#ifdef _OPENMP
  if (prm->nthreads > 0) omp_set_num_threads(prm->nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, tris)
  for (int t = 0; t < T; ++t) {                                                       /* one Path_Trace_Thread */
    local_counters lc = { 0, 0 };
    float* f = priv + (size_t)t * npx * 3;
    for (int y = 0; y < H; ++y)                                                        /* integrators.adb:32-33: for y, for x */
      for (int x = 0; x < W; ++x) {
        f3 color;
        if (prm->aa_on) {
          color = ld3(prm->background);
          for (int i = 0; i < 4; ++i) color = add(color, camera_sample(scn, prm, x, y, base + (uint32_t)(t * 4 + i), &lc));
        } else {
          color = camera_sample(scn, prm, x, y, base + (uint32_t)t, &lc);
        }
        float* o = f + 3 * ((size_t)y * W + x);
        o[0] = color.x; o[1] = color.y; o[2] = color.z;
      }
    rays += lc.rays; tris += lc.tri_tests;
  }
  for (int t = 0; t < T; ++t) {
    const float* f = priv + (size_t)t * npx * 3;
    for (size_t i = 0; i < npx; ++i) {
      f3 cb = add(ld3(f + 3 * i), ld3(accum + 3 * i));                                 /* colBuff(x,y) := color + colBuff(x,y) */
      accum[3 * i] = cb.x; accum[3 * i + 1] = cb.y; accum[3 * i + 2] = cb.z;
    }
  }
  free(priv);
  *spp += T * per;
  if (cnt) { cnt->rays += rays; cnt->tri_tests += tris; cnt->samples += (uint64_t)npx * T * per; }
  return 0;
}

void orc_debug_pass(const orc_scene* scn, const orc_params* prm, float* accum, int32_t* prim_index, int32_t* mat_id, int32_t* prim_type) { /* ray_tracer.adb:208-238 */
  static const float palette[8][3] = { { 0.5f, 0.0f, 0.0f }, { 0.0f, 0.5f, 0.0f }, { 0.0f, 0.0f, 0.5f }, { 0.5f, 0.5f, 0.5f },
                                       { 0.5f, 0.5f, 0.0f }, { 0.5f, 0.0f, 0.5f }, { 0.0f, 0.5f, 0.5f }, { 0.75f, 0.75f, 0.75f } };
  const int W = prm->width, H = prm->height;
#pragma omp parallel for schedule(dynamic, 4)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      local_counters lc = { 0, 0 };
      ray_t r; r.x = x; r.y = y; r.origin = ld3(scn->cam_pos);
      r.direction = normalize(mat_mul_v(scn->cam_matrix, eye_ray_direction(prm, x, y, 0.5f, 0.5f)));
      hit_t h = find_closest_hit(scn, &r, &lc);
      size_t i = (size_t)y * W + x;
      if (!h.is_hit) { if (accum) { accum[3 * i] = 0; accum[3 * i + 1] = 0; accum[3 * i + 2] = 0; } }
      else if (accum) { const float* c = palette[h.matId % 8]; accum[3 * i] = c[0]; accum[3 * i + 1] = c[1]; accum[3 * i + 2] = c[2]; }
      if (prim_index) prim_index[i] = h.is_hit ? h.prim_index : -1;
      if (mat_id) mat_id[i] = h.is_hit ? h.matId : -1;
      if (prim_type) prim_type[i] = h.is_hit ? h.prim_type : -1;
    }
}

void orc_closest_hits(const orc_scene* scn, const float* origins, const float* dirs, int64_t n, orc_hit* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    local_counters lc = { 0, 0 };
    ray_t r; r.x = 0; r.y = 0; r.origin = ld3(origins + 3 * i); r.direction = ld3(dirs + 3 * i);
    hit_t h = find_closest_hit(scn, &r, &lc);
    out[i].t = h.t; out[i].is_hit = h.is_hit; out[i].prim_type = h.is_hit ? h.prim_type : -1; out[i].prim_index = h.prim_index;
    out[i].mat_id = h.matId; out[i].mat = h.mat; out[i].normal[0] = h.normal.x; out[i].normal[1] = h.normal.y; out[i].normal[2] = h.normal.z;
    out[i].tx = h.tx; out[i].ty = h.ty;
  }
}

/* ---- resolve: ray_tracer.adb:281-291, ToneMapping :19-38, ColorToUnsigned_32 :41-57 ---- */

static uint32_t ada_round_u32(float v) {   /* Ada float->integer conversion: round to nearest, ties away from zero */
  if (!(v > 0.0f)) return 0u;              /* (negative / NaN would raise Constraint_Error on type Color) */
  uint32_t u = (uint32_t)v;
  float frac = v - (float)u;
  if (frac >= 0.5f) u += 1u;
  return u;
}

void orc_resolve(const float* accum, int32_t width, int32_t height, int32_t spp, uint32_t* screen) {
  const float g_gamma = 2.0f;                                                         /* ray_tracer.ads:32 */
  float normC = 1.0f / (float)spp;
  for (size_t i = 0; i < (size_t)width * height; ++i) {
    f3 rgb = muls(ld3(accum + 3 * i), normC);
    rgb.x = orc_powf(rgb.x, 1.0f / g_gamma);
    rgb.y = orc_powf(rgb.y, 1.0f / g_gamma);
    rgb.z = orc_powf(rgb.z, 1.0f / g_gamma);
    float cr = min2(rgb.x, 1.0f), cg = min2(rgb.y, 1.0f), cb = min2(rgb.z, 1.0f);
    uint32_t red = ada_round_u32(cr * 255.0f), green = ada_round_u32(cg * 255.0f), blue = ada_round_u32(cb * 255.0f);
    screen[i] = red | (green << 8) | (blue << 16);
  }
}

/* ---- Bitmap.SaveBMP: bitmap.adb:31-85 ---- */

int64_t orc_bmp_bytes(const uint32_t* image, int32_t width, int32_t height, uint8_t* out, int64_t cap) {
  int64_t need = 14 + 40 + (int64_t)width * height * 3;
  if (!out || cap < need) return need;
  uint8_t* p = out;
#define PUT16(v) do { uint16_t _v = (uint16_t)(v); memcpy(p, &_v, 2); p += 2; } while (0)
#define PUT32(v) do { uint32_t _v = (uint32_t)(v); memcpy(p, &_v, 4); p += 4; } while (0)
  PUT16(0x4d42); PUT32(14 + 40 + (uint32_t)(width * height * 3)); PUT16(0); PUT16(0); PUT32(14 + 40);
  PUT32(40); PUT32(width); PUT32(height); PUT16(1); PUT16(24); PUT32(0); PUT32(0); PUT32(0); PUT32(0); PUT32(0); PUT32(0);
  for (int64_t i = 0; i < (int64_t)width * height; ++i) {
    uint32_t pxU = image[i];
    uint8_t b = (uint8_t)((pxU >> 0) & 255), g = (uint8_t)((pxU >> 8) & 255), r = (uint8_t)((pxU >> 16) & 255);
    *p++ = r; *p++ = g; *p++ = b;        /* Pixel'Write writes components r,g,b in declaration order (bitmap.ads:32-34) */
  }
#undef PUT16
#undef PUT32
  return need;
}

int orc_save_bmp(const char* path, const uint32_t* image, int32_t width, int32_t height) {
  int64_t n = orc_bmp_bytes(image, width, height, NULL, 0);
  uint8_t* buf = (uint8_t*)malloc((size_t)n);
  if (!buf) return -1;
  orc_bmp_bytes(image, width, height, buf, n);
  FILE* f = fopen(path, "wb");
  if (!f) { free(buf); return -2; }
  size_t w = fwrite(buf, 1, (size_t)n, f);
  fclose(f); free(buf);
  return (w == (size_t)n) ? 0 : -3;
}

/* ---- LoadMeshFromVSGF: geometry.adb:499-609 ---- */

int orc_load_vsgf_mem(const void* data, int64_t nbytes, const float T[16], orc_mesh* out) {
  const uint8_t* p = (const uint8_t*)data;
  if (nbytes < 24) return -1;
  int64_t fileSize; int32_t nv, ni, nm, flags;
  memcpy(&fileSize, p, 8); memcpy(&nv, p + 8, 4); memcpy(&ni, p + 12, 4); memcpy(&nm, p + 16, 4); memcpy(&flags, p + 20, 4);
  (void)fileSize; (void)nm;
  int nt = ni / 3;
  int64_t need = 24 + (int64_t)nv * 16 * 2 + (int64_t)nv * 8 + (flags != 0 ? (int64_t)nv * 16 : 0) + (int64_t)nt * 12 + (int64_t)nt * 4;
  if (nv <= 0 || nt <= 0 || nbytes < need) return -2;
  float* pos = (float*)malloc(sizeof(float) * 3 * nv), * nrm = (float*)malloc(sizeof(float) * 3 * nv), * uv = (float*)calloc(2 * (size_t)nv, sizeof(float));
  int32_t* idx = (int32_t*)malloc(sizeof(int32_t) * 3 * nt), * mid = (int32_t*)malloc(sizeof(int32_t) * nt);
  const uint8_t* q = p + 24;
  for (int i = 0; i < nv; ++i) { memcpy(pos + 3 * i, q, 12); q += 16; }              /* :541-546 float4 -> xyz */
  for (int i = 0; i < nv; ++i) { memcpy(nrm + 3 * i, q, 12); q += 16; }              /* :550-558 */
  q += (int64_t)nv * 8;                                                               /* :562-567 texcoords read, forced to 0 */
  if (flags != 0) q += (int64_t)nv * 16;                                              /* :571-575 tangents skipped */
  memcpy(idx, q, sizeof(int32_t) * 3 * nt); q += (int64_t)nt * 12;                    /* :580-582 */
  memcpy(mid, q, sizeof(int32_t) * nt);                                               /* :586-589 */
  /* :593-607 transform positions only; bbox = true bounds (the reference leaves bbox uninitialised
     before the min/max loop -- a harmless defect that only enlarges a conservative box). */
  float bmin[3] = { ORC_INFINITY, ORC_INFINITY, ORC_INFINITY }, bmax[3] = { -ORC_INFINITY, -ORC_INFINITY, -ORC_INFINITY };
  for (int i = 0; i < nv; ++i) {
    f3 v = mat_mul_v(T, ld3(pos + 3 * i));
    pos[3 * i] = v.x; pos[3 * i + 1] = v.y; pos[3 * i + 2] = v.z;
    bmin[0] = min2(bmin[0], v.x); bmin[1] = min2(bmin[1], v.y); bmin[2] = min2(bmin[2], v.z);
    bmax[0] = max2(bmax[0], v.x); bmax[1] = max2(bmax[1], v.y); bmax[2] = max2(bmax[2], v.z);
  }
  out->mode = ORC_MESH_REFERENCE_BF; out->nverts = nv; out->ntris = nt;
  out->pos = pos; out->nrm = nrm; out->uv = uv; out->idx = idx; out->matid = mid;
  memcpy(out->bbmin, bmin, 12); memcpy(out->bbmax, bmax, 12);
  return 0;
}

int orc_load_vsgf(const char* path, const float T[16], orc_mesh* out) {
  FILE* f = fopen(path, "rb");
  if (!f) return -10;
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  void* buf = malloc((size_t)n);
  if (!buf) { fclose(f); return -11; }
  size_t rd = fread(buf, 1, (size_t)n, f); fclose(f);
  int rc = (rd == (size_t)n) ? orc_load_vsgf_mem(buf, n, T, out) : -12;
  free(buf);
  return rc;
}

void orc_free_mesh(orc_mesh* m) {
  free((void*)m->pos); free((void*)m->nrm); free((void*)m->uv); free((void*)m->idx); free((void*)m->matid);
  memset(m, 0, sizeof *m);
}

/* ---- Init_Cornell_Box: scene.adb:89-217 ---- */

void orc_cornell_mesh_transform(float out16[16]) {                                    /* scene.adb:194-206 */
  float mrot[16], mscale[16], mtans[16], tmp[16];
  rotation_matrix(-0x1.0c1524p-1f /* static -PI/6.0 = 0xbf060a92 */, v3(0.0f, 1.0f, 0.0f), mrot);
  mat_identity(mscale); mat_identity(mtans);
  mtans[3] = -0.75f; mtans[7] = 0.1f; mtans[11] = 3.1f; mtans[15] = 1.0f;             /* SetCol(mtans,3,...) */
  mscale[0] = 2.0f; mscale[5] = 2.0f; mscale[10] = 2.0f;
  mat_mul_m(mtans, mrot, tmp);
  mat_mul_m(tmp, mscale, out16);
}

void orc_build_cornell(orc_cornell_storage* st, const orc_mesh* pyramid, int use_rect_light) {
  memset(st, 0, sizeof *st);
  orc_scene* s = &st->scene;
  /* lights: scene.adb:104-122 */
  orc_light* L = &st->lights[0];
  const float intensity[3] = { 20.0f, 20.0f, 20.0f };
  if (use_rect_light) {
    L->shape = ORC_LIGHT_RECT;
    L->boxMin[0] = -0.75f; L->boxMin[1] = 4.98f; L->boxMin[2] = 1.25f;
    L->boxMax[0] = 0.75f;  L->boxMax[1] = 4.98f; L->boxMax[2] = 3.25f;
    L->normal[0] = 0.0f; L->normal[1] = -1.0f; L->normal[2] = 0.0f;
    L->intensity[0] = intensity[0]; L->intensity[1] = intensity[1]; L->intensity[2] = intensity[2];
    L->surfaceArea = (L->boxMax[0] - L->boxMin[0]) * (L->boxMax[2] - L->boxMin[2]);
  } else {                                                                            /* the shipped configuration (:128) */
    L->shape = ORC_LIGHT_SPHERE;
    L->center[0] = 0.0f; L->center[1] = 4.5f; L->center[2] = 1.0f; L->radius = 0.5f;
    L->intensity[0] = 0.5f * intensity[0]; L->intensity[1] = 0.5f * intensity[1]; L->intensity[2] = 0.5f * intensity[2];
    L->surfaceArea = 4.0f * M_PI_F * L->radius * L->radius;
  }
  L->mat = 4;
  /* materials: scene.adb:155-180 (6,7 stay null) */
  orc_material* M = st->materials;
  M[0].type = ORC_MAT_GLASS;   M[0].p[0] = M[0].p[1] = M[0].p[2] = 0.75f; M[0].p[3] = M[0].p[4] = M[0].p[5] = 0.85f; M[0].p[6] = 1.75f;
  M[1].type = ORC_MAT_LAMBERT; M[1].p[0] = M[1].p[1] = M[1].p[2] = 0.5f;
  M[2].type = ORC_MAT_LAMBERT; M[2].p[0] = 0.25f; M[2].p[1] = 0.5f; M[2].p[2] = 0.0f;
  M[3].type = ORC_MAT_LAMBERT; M[3].p[0] = 0.5f;  M[3].p[1] = 0.0f; M[3].p[2] = 0.0f;
  M[4].type = ORC_MAT_LIGHT;   M[4].light = 0;
  M[5].type = ORC_MAT_MIRROR;  M[5].p[0] = M[5].p[1] = M[5].p[2] = 0.75f;
  M[8].type = ORC_MAT_PHONG;   M[8].p[0] = M[8].p[1] = M[8].p[2] = 0.75f; M[8].p[3] = 80.0f;
  M[9] = M[1]; M[10] = M[1];
  /* spheres: scene.adb:98,139-144,182-192 */
  st->spheres[0].pos[0] = -1.5f; st->spheres[0].pos[1] = 1.0f; st->spheres[0].pos[2] = 1.5f; st->spheres[0].r = 1.0f; st->spheres[0].mat = 8;
  st->spheres[1].pos[0] = 1.4f;  st->spheres[1].pos[1] = 1.0f; st->spheres[1].pos[2] = 3.0f; st->spheres[1].r = 1.0f; st->spheres[1].mat = 0;
  s->n_spheres = 2;
  if (!use_rect_light) {
    st->spheres[2].pos[0] = 0.0f; st->spheres[2].pos[1] = 4.5f; st->spheres[2].pos[2] = 1.0f; st->spheres[2].r = 0.5f;
    st->spheres[2].mat = 4;   /* new MaterialLight'(lref => g_lightRef): same content as materials(4) */
    s->n_spheres = 3;
  }
  s->spheres = st->spheres;
  /* Cornell box: scene.ads:75-80 */
  s->has_cornell = 1;
  s->cb_min[0] = -2.5f; s->cb_min[1] = 0.0f; s->cb_min[2] = 0.0f; s->cb_max[0] = 2.5f; s->cb_max[1] = 5.0f; s->cb_max[2] = 5.0f;
  { const int mi[6] = { 2, 3, 1, 1, 8, 1 }; memcpy(s->cb_mat, mi, sizeof mi); }
  { const float nn[6][3] = { { 1, 0, 0 }, { -1, 0, 0 }, { 0, 1, 0 }, { 0, -1, 0 }, { 0, 0, 1 }, { 0, 0, -1 } }; memcpy(s->cb_nrm, nn, sizeof nn); }
  s->n_lights = 1; s->lights = st->lights;
  s->n_materials = 11; s->materials = st->materials;
  if (pyramid) { st->meshes[0] = *pyramid; st->meshes[0].mode = ORC_MESH_REFERENCE_BF; s->n_meshes = 1; s->meshes = st->meshes; }
  /* camera: scene.adb:212-215 */
  s->cam_pos[0] = 0.0f; s->cam_pos[1] = 2.55f; s->cam_pos[2] = 12.5f;
  mat_identity(s->cam_matrix);
}

/* ======================================================================================== */
/* BVH8 walk with counters.  NOT reference code: it mirrors the traversal order the product  */
/* publishes in include/art_hip.h / DESIGN.md so that B and T of SURVEY 8(d) can be counted  */
/* on the host for the identical BVH and ray set.                                            */
/* Node = 64 floats: child j: [4j..4j+3] = lo.xyz, ref(int bits); [32+4j..] = hi.xyz, cnt.   */
/*   ref = -1 empty; cnt = 0 inner (ref = node index); cnt > 0 leaf (ref = first triangle).  */
/* Tri  = 12 floats: A.xyz B.xyz C.xyz prim(int bits) pad pad.                               */
/* ======================================================================================== */

static void bvh_walk_one(const float* nodes, const float* tris, int width, const ray_t* rp, float tfar,
                         lite_hit* best, int32_t* best_prim_out, uint64_t* counters) {
  const int W = width, HB = 4 * width;          /* children per node; float offset of the {hi, count} half */
  const ray_t r = *rp;
  /* the product's slab arithmetic (csrc/art_isect.h slab_setup / slab_interval): finite inverse, one fma per plane */
  const float tiny = 1.0e-30f;
  float ddx = (fabsf(r.direction.x) < tiny) ? copysignf(tiny, r.direction.x) : r.direction.x;
  float ddy = (fabsf(r.direction.y) < tiny) ? copysignf(tiny, r.direction.y) : r.direction.y;
  float ddz = (fabsf(r.direction.z) < tiny) ? copysignf(tiny, r.direction.z) : r.direction.z;
  float idx = 1.0f / ddx, idy = 1.0f / ddy, idz = 1.0f / ddz;
  float nox = -(r.origin.x * idx), noy = -(r.origin.y * idy), noz = -(r.origin.z * idz);
  float best_t = tfar;
  int32_t best_prim = -1;
  lite_hit bh; bh.is_hit = 0; bh.tmin = 0.0f; bh.tmax = 0.0f; bh.u = 0.0f; bh.v = 0.0f;
  struct { int32_t ref, cnt; float tmin; } stack[192];
  int sp = 0;
  stack[sp].ref = 0; stack[sp].cnt = 0; stack[sp].tmin = 0.0f; sp++;
  while (sp > 0) {
    --sp;
    int32_t ref = stack[sp].ref, c = stack[sp].cnt; float etmin = stack[sp].tmin;
    if (etmin > best_t) continue;
    if (c == 0) {
      const float* nd = nodes + (size_t)ref * (size_t)(8 * W);
      uint32_t key[8]; int32_t cref[8], ccnt[8]; float ctm[8]; int nh = 0;
      counters[2]++;
      for (int j = 0; j < W; ++j) {
        int32_t rj = f2i(nd[4 * j + 3]);
        if (rj < 0) continue;
        counters[0]++;
        float t0x = fmaf(nd[4 * j + 0], idx, nox), t1x = fmaf(nd[HB + 4 * j + 0], idx, nox);
        float t0y = fmaf(nd[4 * j + 1], idy, noy), t1y = fmaf(nd[HB + 4 * j + 1], idy, noy);
        float t0z = fmaf(nd[4 * j + 2], idz, noz), t1z = fmaf(nd[HB + 4 * j + 2], idz, noz);
        float tmn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
        float tmx = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), best_t));
        if (tmn <= tmx) {
          uint32_t kb; memcpy(&kb, &tmn, 4);
          key[nh] = (kb & ~7u) | (uint32_t)j; cref[nh] = rj; ccnt[nh] = f2i(nd[HB + 4 * j + 3]); ctm[nh] = tmn; nh++;
        }
      }
      for (int a = 1; a < nh; ++a) {              /* ascending insertion sort by key */
        uint32_t k = key[a]; int32_t rr = cref[a], cc = ccnt[a]; float tt = ctm[a]; int b = a - 1;
        while (b >= 0 && key[b] > k) { key[b + 1] = key[b]; cref[b + 1] = cref[b]; ccnt[b + 1] = ccnt[b]; ctm[b + 1] = ctm[b]; --b; }
        key[b + 1] = k; cref[b + 1] = rr; ccnt[b + 1] = cc; ctm[b + 1] = tt;
      }
      for (int a = nh - 1; a >= 0; --a) { stack[sp].ref = cref[a]; stack[sp].cnt = ccnt[a]; stack[sp].tmin = ctm[a]; sp++; }   /* far-to-near */
    } else {
      counters[3]++;
      for (int j = 0; j < c; ++j) {
        const float* tr = tris + (size_t)(ref + j) * 12;
        counters[1]++;
        lite_hit h = intersect_triangle(&r, ld3(tr), ld3(tr + 3), ld3(tr + 6), 0.0f, 1000000.0f);
        int32_t prim = f2i(tr[9]);
        if (h.is_hit && (h.tmin < best_t || (h.tmin == best_t && best_prim >= 0 && prim < best_prim))) { best_t = h.tmin; best_prim = prim; bh = h; }
      }
    }
  }
  *best = bh; *best_prim_out = best_prim;
}

void orc_bvh_walk(const float* nodes, int32_t n_nodes, const float* tris, int32_t n_tris,
                  const float* origins, const float* dirs, const float* tfar, int64_t n,
                  float* out_t, int32_t* out_prim, orc_bvh_counters* cnt) {
  orc_bvh_walk_w(nodes, n_nodes, tris, n_tris, 8, origins, dirs, tfar, n, out_t, out_prim, cnt);
}

void orc_bvh_walk_w(const float* nodes, int32_t n_nodes, const float* tris, int32_t n_tris, int32_t width,
                    const float* origins, const float* dirs, const float* tfar, int64_t n,
                    float* out_t, int32_t* out_prim, orc_bvh_counters* cnt) {
  uint64_t boxes = 0, tritests = 0, nvis = 0, lvis = 0;
  (void)n_nodes; (void)n_tris;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : boxes, tritests, nvis, lvis)
  for (int64_t i = 0; i < n; ++i) {
    ray_t r; r.x = 0; r.y = 0; r.origin = ld3(origins + 3 * i); r.direction = ld3(dirs + 3 * i);
    lite_hit bh; int32_t prim; uint64_t c[4] = { 0, 0, 0, 0 };
    const float tf = tfar ? tfar[i] : ORC_INFINITY;   /* the product's unbounded rays start at Float'Last */
    bvh_walk_one(nodes, tris, width, &r, tf, &bh, &prim, c);
    boxes += c[0]; tritests += c[1]; nvis += c[2]; lvis += c[3];
    if (out_t) out_t[i] = (prim >= 0) ? bh.tmin : tf;
    if (out_prim) out_prim[i] = prim;
  }
  if (cnt) { cnt->rays += (uint64_t)n; cnt->box_tests += boxes; cnt->tri_tests += tritests; cnt->node_visits += nvis; cnt->leaf_visits += lvis; }
}

// art_shade.h -- light sampling, BSDFs and one bounce of the three integrators, written per path slot
// for the wavefront pipeline.  The reference recurses (ray_tracer-integrators.adb:82-301); here every
// bounce records (e_k, w_k) and the final kernel folds them inside-out so the floating-point
// association of  explicit + (|cos|*bxdf) * PathTrace(next)  (integrators.adb:299) is preserved.
//   lights.adb:42-96 (rect), :101-266 (sphere)      materials.adb:18-99, :142-410
//   vector_math.adb:175-326 (sampling helpers)      ray_tracer.adb:61-132 (camera rays, Compute_Shadow)
#pragma once
#include "art_isect.h"

namespace art {

constexpr float kGEpsilon = 1.0e-5f;      // ray_tracer.ads:29
constexpr float kGEpsilonDiv = 1.0e-20f;  // ray_tracer.ads:30
constexpr float kEpsDiv = 1.0e-20f;       // materials.adb:15 / lights.adb:39
constexpr float kEpsCos = 1.0e-6f;        // materials.adb:16
constexpr float kPhongClamp = 0x1.921eaep+0f;  // static M_PI*0.499995 (materials.adb:374)

// What a kernel may hand to the per-item code besides the scene: copies of the small scene tables it keeps in LDS (every ray tests
// every sphere and every shaded hit reads its material: a global load each is a round trip each -- that alone kept the first fused
// stage at 3.3 TB/s), and the LDS staging of the records a wave emits (emit_ray).  All optional: the host simulation passes none.
struct StageCtx {
  const DevSphere* spheres = nullptr; const DevLight* lights = nullptr; const DevMaterial* materials = nullptr;   // nullptr: the scene's own tables
  // the same two tables as LDS-qualified pointers (k_shade_compact, k_raygen: the kernel's __shared__ copies).  When set they are what the
  // per-item code reads (ds_read: no vmcnt wait, art_math.h ART_LDS); `spheres` / `lights` above then stay unused
  const ART_LDS DevSphere* lds_spheres = nullptr; const ART_LDS DevLight* lds_lights = nullptr;
  Rec4* stage = nullptr; int stage_pitch = 0; int stage_item = 0;     // LDS staging of the wave's records (emit_ray): k_raygen copies them out itself
  // k_shade_compact: stage_count > 0 lanes of the wave emit together (stage_item = this lane's rank among them) and copy the records out
  // themselves, one kind of ray at a time; rec_base[0 / 1]: first record of the wave's extension / shadow rays in the output bank
  int stage_count = 0; size_t rec_base[2] = {0, 0};
  bool hot_layout = false;                     // k_shade_compact: the banks are addressed through DevPaths::hot (one base pointer per bank)
  unsigned long long* tprobe = nullptr;        // -DART_TIME_PROBE: the wave's time-probe word (LDS)
  unsigned long long* lost = nullptr;          // the self-check counter (ArtStats::lost_paths): a staged record read before its lane wrote it is counted there
};

// the record schedule's blocks (art_scene.h HotField, DevPaths::cold): a field's base is wave-uniform -- scalar arithmetic on one base pointer
template <class T = float> ART_HD T* hotf(const DevPaths& q, int field) { return reinterpret_cast<T*>(q.hot + (size_t)field * (size_t)q.stride); }
ART_HD float* cold_e(const DevPaths& q, int c, int level) { return q.cold + ((size_t)c * (size_t)(q.depth + 1) + (size_t)level) * (size_t)q.P; }
ART_HD float* cold_w(const DevPaths& q, int c, int level) { return q.cold + ((size_t)3 * (size_t)(q.depth + 1) + (size_t)c * (size_t)q.depth + (size_t)level) * (size_t)q.P; }
ART_HD int32_t* cold_child(const DevPaths& q, int level) { return reinterpret_cast<int32_t*>(q.cold + ((size_t)3 * (size_t)(q.depth + 1) + (size_t)3 * (size_t)q.depth + (size_t)level) * (size_t)q.P); }

// The rays an item leaves a bounce with, for a caller that writes the trace records itself (k_shade_compact: the wave stages one kind of
// ray at a time and copies it out at a point every lane of the wave reaches).
struct RayOut { bool alive, shadow; f3 no, nd, so, sd; float s_tfar, sh_min; };

// Diagnostic builds only (wrong pictures; profiles/r4_shade/sensitivity.txt): -DART_DIAG_SKIP=<bits> leaves out one kind of the stage's traffic --
// 1 trace-record copy-out, 2 the triangle-normal gather, 4 the next ray as SoA words, 8 the hit-record (starting bound) stores, 16 the fold records
#ifndef ART_DIAG_SKIP
#define ART_DIAG_SKIP 0
#endif
#ifndef ART_SYNTH_DIR
#define ART_SYNTH_DIR 1           // DevPaths::synth0: 1 = bounce 0 recomputes the camera direction as well (raygen then stores a hit and a record, nothing else), 0 = raygen stores the direction (measured equal within noise on C3 / C4, 2 % slower on C5)
#endif
ART_HD void wave_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// ---------------------------------------------------------------- sampling helpers
ART_HD f3 perpendicular(f3 a) {   // GetPerpendicular
  const float xp = fabsf(a.x), yp = fabsf(a.y), zp = fabsf(a.z);
  f3 least;
  if ((xp <= yp + 1.0e-5f) && (xp <= zp + 1.0e-5f)) least = mk3(1.0f, 0.0f, 0.0f);
  else if ((yp < xp + 1.0e-5f) && (yp <= zp + 1.0e-5f)) least = mk3(0.0f, 1.0f, 0.0f);
  else least = mk3(0.0f, 0.0f, 1.0f);
  return normalize(cross(a, least));
}

// shared tail of MapSampleToCosineDist / MapSampleToCosineDistFixed: local frame + under-surface fix-up
ART_HD f3 lobe_to_world(f3 dev, f3 direction, f3 normal) {
  f3 ny = direction;
  f3 nx = perpendicular(ny);
  f3 nz = normalize(cross(nx, ny));
  { const f3 tmp = ny; ny = nz; nz = tmp; }
  f3 res = (nx * dev.x + ny * dev.y) + nz * dev.z;
  const float inv_sign = (dot(direction, normal) >= 0.0f) ? 1.0f : -1.0f;
  ART_PROBE(24);
  if (inv_sign * dot(res, normal) < 0.0f) {
    ART_PROBE(25);
    nx = normalize(cross(normal, direction));
    nz = normalize(cross(nx, ny));
    if (dot(nz, res) < 0.0f) nz = neg(nz);
    res = reflect(neg(res), nz);
    if (dot(res, normal) < 0.0f) res = direction;
  }
  return res;
}

ART_HD f3 sample_cosine(float r1, float r2, f3 direction, f3 normal, float power) {   // vector_math.adb:202-252
  float sp, cp;
  asincos_m1(2.0f * r1 * kPi, sp, cp);
  const float ct = apow(1.0f - r2, 1.0f / (power + 1.0f));
  const float st = sqrtf(1.0f - ct * ct);
  return lobe_to_world(mk3(st * cp, st * sp, ct), direction, normal);
}

ART_HD f3 sample_cosine_fixed(float r1, float r2, f3 direction, f3 normal, float power) {   // vector_math.adb:255-312
  const float h = sqrtf(1.0f - apow(r1, 2.0f / (power + 1.0f)));
  float s2, c2;
  asincos_m1((2.0f * kPi) * r2, s2, c2);
  return lobe_to_world(mk3(h * c2, h * s2, apow(r1, 1.0f / (power + 1.0f))), direction, normal);
}

// ---------------------------------------------------------------- lights
struct LightSample { f3 pos, dir, intensity; float pdf; };

ART_HD float pdf_area_to_solid(float pdfA, float dist, float cos_there) {   // PdfAtoW
  return pdfA * dist * dist / amax(cos_there, kEpsDiv);
}

ART_HD float dist2(f3 a, f3 b) { const f3 q = b - a; return dot(q, q); }

// (L: a pointer to the light, generic or LDS-qualified -- light_at() below hands the per-item code whichever the kernel provides)
template <class L>
ART_HD float sphere_light_pdf(L l, f3 p) {   // SphereLight.EvalPDF
  const f3 c = ld3(l->center);
  const float radius = l->radius;
  if (dist2(p, c) - radius * radius < 1.0e-4f) return 1.0f / l->surfaceArea;
  const float s2 = radius * radius / dist2(p, c);
  const float cmax = sqrtf(amax(0.0f, 1.0f - s2));
  return 1.0f / (2.0f * kPi * (1.0f - cmax));
}

template <class L>
ART_HD float light_eval_pdf(L l, f3 p, f3 ray_dir, float hit_dist) {
  if (l->shape == LIGHT_RECT) {
    const float ct = amax(dot(ray_dir, neg(ld3(l->normal))), 0.0f);
    return pdf_area_to_solid(1.0f / l->surfaceArea, hit_dist, ct);
  }
  return sphere_light_pdf(l, p);
}

template <class L>
ART_HD LightSample light_sample(L lp, float u1, float u2, f3 p) {
  LightSample r;
  r.pos = mk3(0.0f, 0.0f, 0.0f); r.dir = r.pos; r.pdf = 1.0f;
  r.intensity = ld3(lp->intensity);
  ART_PROBE(10);
  if (lp->shape == LIGHT_RECT) {                                 // AreaLight.Sample
    ART_PROBE(11);
    r.pos.x = lp->boxMin[0] + u1 * (lp->boxMax[0] - lp->boxMin[0]);
    r.pos.y = lp->boxMin[1];
    r.pos.z = lp->boxMin[2] + u2 * (lp->boxMax[2] - lp->boxMin[2]);
    r.dir = ld3(lp->normal);
    f3 rd = r.pos - p;
    const float dd = length(rd);
    rd = rd * (1.0f / dd);
    const float ct = amax(dot(rd, neg(ld3(lp->normal))), 0.0f);
    r.pdf = pdf_area_to_solid(1.0f / lp->surfaceArea, dd, ct);
    return r;
  }
  struct { float radius; } l = {lp->radius};                     // (the sphere light's radius, read once)
  const f3 c = ld3(lp->center);                                  // SphereLight.Sample
  if (dist2(p, c) - l.radius * l.radius < 1.0e-4f) {
    ART_PROBE(12);
    const float z = 1.0f - 2.0f * u1;                            // UniformSampleSphere
    const float rr = sqrtf(amax(0.0f, 1.0f - z * z));
    float sph, cph;
    asincos_m1(2.0f * kPi * u2, sph, cph);
    r.pos = c + l.radius * mk3(rr * cph, rr * sph, z);
    r.dir = normalize(r.pos - c);
    return r;                                                    // pdf stays 1.0 (lights.adb:210-214)
  }
  const f3 wc = normalize(c - p);
  f3 wx, wy;                                                     // CoordinateSystem
  ART_PROBE(13);
  if (fabsf(wc.x) > fabsf(wc.y)) {
    ART_PROBE(14);
    const float il = 1.0f / sqrtf(wc.x * wc.x + wc.z * wc.z);
    wx = mk3(-wc.z * il, 0.0f, wc.x * il);
  } else {
    const float il = 1.0f / sqrtf(wc.y * wc.y + wc.z * wc.z);
    wx = mk3(0.0f, wc.z * il, -wc.y * il);
  }
  wy = cross(wc, wx);
  ART_PROBE(15);
  const float s2 = l.radius * l.radius / dist2(p, c);
  const float cmax = sqrtf(amax(0.0f, 1.0f - s2));
  const float ct = alerp(u1, cmax, 1.0f);                        // UniformSampleCone
  const float st = sqrtf(1.0f - ct * ct);
  float sph, cph;
  asincos_m1(u2 * 2.0f * kPi, sph, cph);
  const f3 rdir = ((cph * st) * wx + (sph * st) * wy) + ct * wc;
  const f3 rpos = p + rdir * 1.0e-3f;
  float thit;
  {                                                              // RaySphereIntersect
    const f3 k = rpos - c;
    const float b = dot(k, rdir);
    const float cc = dot(k, k) - l.radius * l.radius;
    const float disc = b * b - cc;
    float hx;
    if (disc >= 0.0f) { ART_PROBE(16); const float sq = sqrtf(disc); hx = amin(-b - sq, -b + sq); }
    else hx = -kInfinity;
    thit = (hx < 0.0f) ? dot(c - p, normalize(rdir)) : hx;
  }
  ART_PROBE(17);
  r.pos = rpos + thit * rdir;
  r.dir = normalize(r.pos - c);
  r.pdf = sphere_light_pdf(lp, p);
  return r;
}

// ---------------------------------------------------------------- materials
struct BsdfSample { f3 color, dir; float pdf; bool specular; };

// Material sets (round 5).  The per-item code below is compiled once per SET of material types (template parameter MATS, one bit per
// MatType): a kernel instantiated for a set contains no instruction of the other materials, so that the Lambert majority of a scene does
// not carry the registers of Phong's binary64 pow or of the glass branch (k_shade_compact: one instantiation per register class).
// A material outside the set cannot reach the instantiation (the caller sorts the items by class first); should one ever, it shades
// as MaterialLight's default (black) and shade_item counts a lost path.
constexpr int mat_bit(int type) { return ((unsigned)type < 8u) ? (1 << type) : 0; }
constexpr int kMatsAll = mat_bit(MAT_NULL) | mat_bit(MAT_LIGHT) | mat_bit(MAT_LAMBERT) | mat_bit(MAT_MIRROR) | mat_bit(MAT_GLASS) | mat_bit(MAT_PHONG);

ART_HD float fresnel_unpolarised(float cos1, float eta_ext_in, float eta_int_in) {   // materials.adb:70-99
  float ext = eta_ext_in, in = eta_int_in;
  if (cos1 < 0.0f) { const float tmp = ext; ext = in; in = tmp; }
  const float sin2 = (ext / in) * sqrtf(amax(0.0f, 1.0f - cos1 * cos1));
  if (sin2 > 1.0f) return 1.0f;
  const float cos2 = sqrtf(amax(0.0f, 1.0f - sin2 * sin2));
  const float c1 = fabsf(cos1);
  // fresnelDielectric(cosTheta1 => |cos1|, cosTheta2, etaExt => in, etaInt => ext)
  const float rs = (in * c1 - ext * cos2) / (in * c1 + ext * cos2);
  const float rp = (ext * c1 - in * cos2) / (ext * c1 + in * cos2);
  return (rs * rs + rp * rp) / 2.0f;
}

template <int MATS = kMatsAll>
ART_HD BsdfSample bsdf_sample(const DevMaterial& m, float xi1, float xi2, f3 ray_dir, f3 n) {
  BsdfSample r;
  const int32_t type = (MATS & mat_bit(m.type)) ? m.type : (int32_t)MAT_LIGHT;      // outside the set: the default case
  switch (type) {
    case MAT_LAMBERT: if (MATS & mat_bit(MAT_LAMBERT)) {        // materials.adb:197-215
      ART_PROBE(20);
      const f3 nd = sample_cosine(xi1, xi2, n, n, 1.0f);
      const float ct = dot(nd, n);
      r.pdf = fabsf(ct) * kInvPi;
      r.color = ld3(m.p) * kInvPi;
      if (ct < kEpsCos) r.color = mk3(0.0f, 0.0f, 0.0f);
      r.dir = nd; r.specular = false;
      return r;
    } break;
    case MAT_MIRROR: if (MATS & mat_bit(MAT_MIRROR)) {          // :247-254
      ART_PROBE(21);
      const f3 nd = reflect(ray_dir, n);
      const float cdiv = 1.0f / amax(dot(nd, n), kEpsDiv);
      r.color = ld3(m.p) * cdiv; r.dir = nd; r.pdf = 1.0f; r.specular = true;
      return r;
    } break;
    case MAT_GLASS: if (MATS & mat_bit(MAT_GLASS)) {            // :285-331
      ART_PROBE(22);
      const float ior = m.p[6];
      const float f = fresnel_unpolarised(dot(ray_dir, n), ior, 1.0f);
      const f3 refl = f * ld3(m.p);
      const f3 trans = (1.0f - f) * ld3(m.p + 3);
      const float k_trans = length(trans) / (length(refl) + length(trans));
      const float k_refl = length(refl) / (length(refl) + length(trans));
      f3 nd, bx;
      if (xi1 > k_trans) {
        nd = reflect(ray_dir, n);
        bx = refl * (1.0f / k_refl);
      } else {
        bx = trans * (1.0f / k_trans);
        float ci = dot(neg(ray_dir), n);                        // TotalInternalReflection :18-32
        float eta = ior;
        if (ci < 0.0f) eta = 1.0f / eta;
        const bool tir = (1.0f - (1.0f - ci * ci) / (eta * eta)) < 0.0f;
        if (!tir) {                                             // refract :34-52
          f3 nn = n;
          const f3 wo = neg(ray_dir);
          if (ci < 0.0f) { ci = -ci; nn = neg(nn); }
          const float c2 = sqrtf(1.0f - (1.0f - ci * ci) / (eta * eta));
          nd = normalize((neg(wo) * (1.0f / eta)) - ((c2 - ci / eta) * nn));
        } else nd = reflect(ray_dir, n);
      }
      const float cdiv = 1.0f / amax(fabsf(dot(nd, n)), kEpsDiv);
      r.color = bx * cdiv; r.dir = nd; r.pdf = 1.0f; r.specular = true;
      return r;
    } break;
    case MAT_PHONG: if (MATS & mat_bit(MAT_PHONG)) {            // :363-387
      ART_PROBE(23);
      const float pw = m.p[3];
      const f3 rr = reflect(ray_dir, n);
      const f3 nd = sample_cosine_fixed(xi1, xi2, rr, n, pw);
      const float ct = aclamp(dot(nd, rr), 0.0f, kPhongClamp);
      const float lobe = apow(ct, pw);
      f3 col = (((ld3(m.p) * (pw + 2.0f)) * 0.5f) * kInvPi) * lobe;
      r.pdf = lobe * (pw + 1.0f) * (0.5f * kInvPi);
      const float cg = dot(nd, n);
      const float cdiv = 1.0f / amax(fabsf(cg), kEpsDiv);
      if (cg < kEpsCos) col = mk3(0.0f, 0.0f, 0.0f);
      r.color = col * cdiv; r.dir = nd; r.specular = false;
      return r;
    } break;
    default: break;
  }
  r.color = mk3(0.0f, 0.0f, 0.0f); r.dir = r.color; r.pdf = 1.0f; r.specular = false;      // MaterialLight :163-166
  return r;
}

template <int MATS = kMatsAll>
ART_HD void bsdf_eval(const DevMaterial& m, f3 l, f3 v, f3 n, f3& bxdf, float& pdf) {
  const int32_t type = (MATS & mat_bit(m.type)) ? m.type : (int32_t)MAT_LIGHT;
  switch (type) {
    case MAT_LAMBERT: if (MATS & mat_bit(MAT_LAMBERT)) {        // :217-226
      bxdf = ld3(m.p) * kInvPi;
      pdf = amax(dot(n, l), 0.0f) * kInvPi;
      return;
    } break;
    case MAT_PHONG: if (MATS & mat_bit(MAT_PHONG)) {            // :389-410
      const float pw = m.p[3];
      const f3 rr = reflect(neg(v), n);
      const float ct = aclamp(dot(l, rr), 0.0f, kPhongClamp);
      const float lobe = apow(ct, pw);
      const float cdiv = 1.0f / amax(dot(l, n), kEpsDiv);
      bxdf = ((((ld3(m.p) * (pw + 2.0f)) * 0.5f) * kInvPi) * lobe) * cdiv;
      pdf = lobe * (pw + 1.0f) * (0.5f * kInvPi);
      return;
    } break;
    default: break;
  }
  bxdf = mk3(0.0f, 0.0f, 0.0f); pdf = 1.0f;                       // light / mirror / glass
}

// shading record of triangle hit index `idx` (art_scene.h KEY_TRI): its own index, or -- instanced scenes -- the record of the mesh's triangle
ART_HD size_t shade_record(const DevScene& s, uint32_t idx) {
  if (s.n_inst > 0) return (size_t)s.inst[idx >> s.inst_shift].shade_base + (size_t)(idx & ((1u << s.inst_shift) - 1u));
  return (size_t)idx;
}
// material of rect light `idx` (the flat light's quad, geometry.adb:132-141)
ART_HD int32_t light_mat(const DevScene& s, const StageCtx& cx, uint32_t idx) { return cx.lds_lights ? cx.lds_lights[idx].mat : (cx.lights ? cx.lights : s.lights)[idx].mat; }

// ---------------------------------------------------------------- hit record -> shading frame
struct Surface { f3 normal; int32_t mat; int32_t mat_id; };

// pre.on: the three vertex normals of a BVH-mesh triangle's shading record, already fetched (shade_item's hinted path)
struct PreNormals { f3 a, b, c; bool on; };
ART_HD Surface surface_at(const DevScene& s, f3 o, f3 d, float t, uint32_t key, float u, float v, const StageCtx& cx = StageCtx(),
                          const PreNormals pre = PreNormals{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, false}) {
  Surface sf;
  const uint32_t cls = key & ~KEY_INDEX_MASK, idx = key & KEY_INDEX_MASK;
  if (cls == KEY_SPHERE) {                                      // geometry.adb:88,95-96
    const DevSphere sp = cx.lds_spheres ? load_sphere(cx.lds_spheres, (int)idx) : load_sphere(cx.spheres ? cx.spheres : s.spheres, (int)idx);
    sf.normal = normalize((o + d * t) - mk3(sp.x, sp.y, sp.z));
    sf.mat = s.sphere_mat[idx]; sf.mat_id = 0;
  } else if (cls == KEY_CORNELL) {                              // geometry.adb:215-224 + scene.adb:80-82
    sf.normal = ld3(s.cb_nrm[idx]);
    sf.mat_id = s.cb_mat[idx]; sf.mat = sf.mat_id;
  } else if (cls == KEY_QUAD) {                                 // geometry.adb:132-141
    sf.normal = mk3(0.0f, -1.0f, 0.0f);
    sf.mat = light_mat(s, cx, idx); sf.mat_id = 0;
  } else {
    const float w = 1.0f - u - v;                               // geometry.adb:301
    if (cls == KEY_BFTRI) {
      const int32_t* ix = s.bf_idx + 3 * (size_t)idx;
      const float* nr = s.bf_nrm;
      sf.normal = (w * ld3(nr + 3 * (size_t)ix[0]) + v * ld3(nr + 3 * (size_t)ix[1])) + u * ld3(nr + 3 * (size_t)ix[2]);
      sf.mat_id = 2;                                            // geometry.adb:311 hard-codes 2
    } else if (pre.on) {
      sf.normal = (w * pre.a + v * pre.b) + u * pre.c;
      sf.mat_id = -1;                                                          // the caller has the material index from its hint
    } else if (s.n_inst > 0) {                                                 // instanced scene: the mesh's record, its normals taken to world space
      const DevInstance& R = s.inst[idx >> s.inst_shift];
      const float* r = s.m_shade + (size_t)kTriShadeFloats * ((size_t)R.shade_base + (size_t)(idx & ((1u << s.inst_shift) - 1u)));
      sf.normal = (w * instance_normal(R.minv, ld3(r)) + v * instance_normal(R.minv, ld3(r + 3))) + u * instance_normal(R.minv, ld3(r + 6));
      sf.mat_id = __builtin_bit_cast(int32_t, r[9]);
    } else {
      const float* r = s.m_shade + (size_t)kTriShadeFloats * (size_t)idx;      // the triangle's own record: normals of A, B, C and the material id
      sf.normal = (w * ld3(r) + v * ld3(r + 3)) + u * ld3(r + 6);
      sf.mat_id = __builtin_bit_cast(int32_t, r[9]);
    }
    sf.mat = sf.mat_id;
  }
  return sf;
}

// ---------------------------------------------------------------- rays leave a stage as trace records (round 3)
// Candidates 1-4 of Scene.Find_Closest_Hit (scene.adb:62-69: spheres, Cornell box, rect lights, the reference's brute-force mesh):
// the starting bound of the BVH search.  The same calls in the same order as k_analytic and closest_hit().
template <class SP, class LP>
ART_HD Cand analytic_bound_t(const DevScene& s, SP sph, LP lgt, f3 o, f3 d, f3 rcp, float tfar) {
  Cand best = cand_init(tfar);
  for (int i = 0; i < s.n_spheres; ++i) isect_sphere(o, d, load_sphere(sph, i), (uint32_t)i, best);
  if (s.has_cornell) isect_cornell(o, d, rcp, s, best);
  for (int i = 0; i < s.n_lights; ++i)
    if (lgt[i].shape == LIGHT_RECT) isect_quad(o, d, lgt + i, (uint32_t)i, best);
  isect_bf_mesh(o, d, rcp, s, best);
  return best;
}
// rcp: the ray's reciprocals (ray_rcp), shared with the caller's slab set-up
ART_HD Cand analytic_bound(const DevScene& s, const StageCtx& cx, f3 o, f3 d, f3 rcp, float tfar) {
  if (cx.lds_spheres) return analytic_bound_t(s, cx.lds_spheres, cx.lds_lights, o, d, rcp, tfar);      // the kernel's LDS copies (both tables or neither)
  return analytic_bound_t(s, cx.spheres ? cx.spheres : s.spheres, cx.lights ? cx.lights : s.lights, o, d, rcp, tfar);
}

ART_HD size_t rec_slot(int mode, int w, bool shadow_ray) { return (mode == REC_BOTH) ? 2 * (size_t)w + (shadow_ray ? 1u : 0u) : (size_t)w; }

// One ray of an output item: its hit record (the starting bound: what stands if the BVH finds nothing nearer) and its 64-byte trace
// record (art_kernels.h) -- everything k_analytic does for the plain layout, done where the ray is born.  shm >= 0: a shadow ray under
// the visibility rule (art_isect.h shadow_rule).  A ray that does not exist (live = false) or is already decided leaves a record whose
// bound is negative: the trace kernel takes it and retires it at once.
struct TraceRec { Rec4 r0, r1, r2, r3; };
// hit_index: the ray's hit slot in qo.hit, or kShadowWord | item: a shadow ray of the record schedule, whose result is the word qo.sh_t[item]
ART_HD TraceRec make_record(const DevScene& s, const DevPaths& qo, size_t hit_index, bool live, f3 o, f3 d, float tfar, float shm, const StageCtx& cx = StageCtx()) {
  TraceRec t;
  const bool word = ((uint32_t)hit_index & kShadowWord) != 0u;
  float* const sh_t_out = cx.hot_layout ? hotf(qo, HF_SHT) : qo.sh_t;
  if (word && !live) put_s(sh_t_out, (int)((uint32_t)hit_index & ~kShadowWord), -1.0f);       // (an item without a shadow ray: never read, kept defined)
  t.r0 = Rec4{0.0f, 0.0f, 0.0f, -1.0f}; t.r1 = Rec4{0.0f, 0.0f, 0.0f, __builtin_bit_cast(float, KEY_MISS)}; t.r2 = Rec4{0.0f, 0.0f, 0.0f, -1.0f};
  t.r3 = Rec4{0.0f, __builtin_bit_cast(float, (uint32_t)hit_index), 0.0f, 0.0f};
  ART_PROBE(30);
  if (live) {
    ART_PROBE(31);
#ifndef ART_SHARE_RCP
#define ART_SHARE_RCP 1
#endif
    const f3 rcp = ray_rcp(d);                                   // three divisions per ray: the Cornell box, the brute-force mesh's box and the slab set-up below share them
    const Cand best = analytic_bound(s, cx, o, d, rcp, tfar);
    if (word) put_s(sh_t_out, (int)((uint32_t)hit_index & ~kShadowWord), (best.key != KEY_MISS) ? best.t : -1.0f);
    else if (cx.hot_layout) put_s(hotf<DevHit>(qo, HF_HIT), (int)hit_index, DevHit{best.t, best.key, best.u, best.v});
    else if (!(ART_DIAG_SKIP & 8)) qo.hit[hit_index] = DevHit{best.t, best.key, best.u, best.v};
    const bool near_done = (shm >= 0.0f) && (best.key != KEY_MISS) && (best.t <= shm);       // shadow_rule: decided
    if (!near_done && qo.has_bvh) {
      ART_PROBE(32);
      f3 inv, noi;
      if (ART_SHARE_RCP) slab_setup_rcp(o, d, rcp, inv, noi); else slab_setup(o, d, inv, noi);
      const bool far_found = (shm >= 0.0f) && (best.key != KEY_MISS);
      const float bt = far_found ? next_up_pos(shm) : best.t;
      const uint32_t bk = far_found ? KEY_MISS : best.key;
      const uint32_t sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
      const uint32_t sel_near = (sx ? 3u : 0u) | ((sy ? 4u : 1u) << 8) | ((sz ? 5u : 2u) << 16) | 0x0c000000u;
      t.r0 = Rec4{o.x, o.y, o.z, bt}; t.r1 = Rec4{d.x, d.y, d.z, __builtin_bit_cast(float, bk)}; t.r2 = Rec4{inv.x, inv.y, inv.z, shm};
      t.r3.x = __builtin_bit_cast(float, sel_near); t.r3.z = __builtin_bit_cast(float, far_found ? 1u : 0u);
    }
  }
  return t;
}

// ... stored at position `rq` of the output bank, or (cx.stage != nullptr: k_raygen) as four quarters in the wave's LDS staging area,
// stage[quarter * stage_pitch + stage_slot]; the kernel then writes the wave's records out as contiguous kilobytes (64 lanes storing 16
// bytes at a 64-byte stride cost the vector-memory path four times as much).  k_shade_compact stages and copies by itself (round 4).
ART_HD void emit_ray(const DevScene& s, const DevPaths& qo, size_t hit_index, size_t rq, bool live, f3 o, f3 d, float tfar, float shm,
                     const StageCtx& cx = StageCtx(), int stage_slot = 0, int kind = 0) {
  const TraceRec t = make_record(s, qo, hit_index, live, o, d, tfar, shm, cx);
  if (!qo.has_bvh) return;                              // no trace kernel will run: the hit record is the whole answer
  if (cx.stage) {
    Rec4* const stage = cx.stage; const int stage_pitch = cx.stage_pitch;
    stage[stage_slot] = t.r0; stage[stage_pitch + stage_slot] = t.r1; stage[2 * stage_pitch + stage_slot] = t.r2; stage[3 * stage_pitch + stage_slot] = t.r3;
    if (cx.stage_count > 0) {
      // k_shade_compact: the wave's stage_count emitting lanes are its lanes [0, stage_count) (sorted rounds: the survivors come first) and
      // all of them are here -- the call sits behind no material branch.  They copy their records out, 16 bytes per lane and pass,
      // consecutive lanes to consecutive addresses.  LDS operations of one wave execute in order.
      wave_fence();
      Rec4* out = qo.rec + 4 * cx.rec_base[kind];
      // the copy-out relies on every emitting lane having staged its record before any lane reads it, i.e. on the compiler
      // keeping this call at ONE place that all of them reach together.  Checked at run time: the last quarter of record q of the wave
      // must name hit slot (first item + q) of this kind, else the stage counts a lost path (tests assert the counter stays 0).
      const uint32_t hit0 = (uint32_t)hit_index - (uint32_t)stage_slot; (void)hit0;
      if (!(ART_DIAG_SKIP & 1)) for (int it = 0; it < 4; ++it) {
        const int g = it * cx.stage_count + stage_slot;
        const Rec4 v = stage[(g & 3) * stage_pitch + (g >> 2)];
#if (ART_NT & 1) && defined(__HIP_DEVICE_COMPILE__)
        { typedef float f4nt __attribute__((ext_vector_type(4))); __builtin_nontemporal_store(f4nt{v.x, v.y, v.z, v.w}, reinterpret_cast<f4nt*>(&out[g])); }      // experiment: written once, read once by the trace kernel
#else
        out[g] = v;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        if ((g & 3) == 3 && cx.lost != nullptr && __builtin_bit_cast(uint32_t, v.y) != hit0 + (uint32_t)(g >> 2)) atomicAdd(cx.lost, 1ull);
#endif
      }
      wave_fence();
    }
    return;
  }
  Rec4* out = qo.rec + 4 * rq;
  out[0] = t.r0; out[1] = t.r1; out[2] = t.r2; out[3] = t.r3;
}

// ---------------------------------------------------------------- camera (ray_tracer.adb:61-97, integrators.adb:37-58)
ART_HD void slot_to_sample(const DevPaths& q, int slot, uint32_t& pixel, uint32_t& sample) {
  const int sl = slot / q.npix, pl = slot - sl * q.npix;
  pixel = q.pixmap ? q.pixmap[pl] : (uint32_t)pl;
  sample = q.sample_base + (uint32_t)sl;
}

ART_HD f3 camera_dir(const DevFrame& f, const DevScene& s, uint32_t pixel, uint32_t sample) {
  const int x = (int)(pixel % (uint32_t)f.width), y = (int)(pixel / (uint32_t)f.width);
  float ox = 0.5f, oy = 0.5f;
  if (f.aa_on) {   // Generate4RayDirections: (1/3,1/3) (1/3,2/3) (2/3,1/3) (2/3,2/3)
    ox = (sample & 2u) ? 0x1.555556p-1f : 0x1.555556p-2f;
    oy = (sample & 1u) ? 0x1.555556p-1f : 0x1.555556p-2f;
  }
  f3 r;
  r.x = (float)x + ox - ((float)f.width / 2.0f);
  r.y = (float)y + oy - ((float)f.height / 2.0f);
  r.z = f.cam_z;
  return normalize(xform_point(s.cam_matrix, normalize(r)));
}

ART_HD void raygen_slot(const DevFrame& f, const DevScene& s, const DevPaths& q, int slot, const StageCtx& cx = StageCtx()) {
  uint32_t pixel, sample;
  slot_to_sample(q, slot, pixel, sample);
  const f3 d = camera_dir(f, s, pixel, sample);
  if (!q.synth0) { q.ray_ox[slot] = s.cam_pos[0]; q.ray_oy[slot] = s.cam_pos[1]; q.ray_oz[slot] = s.cam_pos[2]; }
  if (!q.synth0 || !ART_SYNTH_DIR) { q.ray_dx[slot] = d.x; q.ray_dy[slot] = d.y; q.ray_dz[slot] = d.z; }
  if (q.rec) emit_ray(s, q, (size_t)slot, (size_t)slot, true, ld3(s.cam_pos), d, kInfinity, -1.0f, cx, cx.stage_item);     // REC_EXT: record `slot`
  else {
    q.ray_tfar[slot] = kInfinity;
    q.ray_tfar[q.P + slot] = -1.0f;
  }
  if (!q.synth0) {                                   // (synth0: bounce 0 knows all of this without being told, DevPaths::synth0)
    q.prev_pdf[slot] = 1.0f;                         // StartSample (materials.ads:25)
    q.flags[slot] = FLAG_ALIVE | FLAG_PREV_SPEC;
  }
  if (!q.fold_dense) { q.term_r[slot] = 0.0f; q.term_g[slot] = 0.0f; q.term_b[slot] = 0.0f; }      // (dense fold records: a path's end is a record of its level)
}

// ---------------------------------------------------------------- one bounce
// Work items and slots (art_scene.h): item w of the INPUT set `qi` is path slot item_slot(qi, w); what the path needs for its next
// bounce is written to item `wo` of the OUTPUT set `qo` (wo < 0: the path is known to need nothing more).  With qi == qo and wo == w
// (identity layout) this is the plain in-place update; all reads of an item happen before its writes.
ART_HD int item_slot(const DevPaths& q, int w) { return q.slot_id ? (int)at(q.slot_id, w) : w; }

// the child word of a dense fold record: bits 0..27 the item at the next bounce, bits 28..29 the kind (0: that item exists; 1: the path ended
// here, the record's w is its value; 2: a surface whose successor was not kept), bit 30: the shadow test this item resolved for its
// PREDECESSOR came out "in shadow" (the predecessor's explicit light, e[this level][this item], then counts as 0)
constexpr int32_t kFoldIndexMask = (1 << 28) - 1;      // (a batch has at most 2^27 items: art_api.cpp refuses more than 2^28 rays per trace launch)
ART_HD int32_t fold_child_word(int32_t child, bool shadowed) {
  const int32_t kind = (child >= 0) ? 0 : (child == -1) ? 1 : 2;
  return ((child >= 0) ? child : 0) | (kind << 28) | (shadowed ? (1 << 30) : 0);
}

// What shade_item(bounce) will do with item w, decided from the hit alone (flags, hit key, material type) before anything is shaded:
// k_shade_compact sorts a workgroup's items by this class, so that a wave runs ONE material's code with (nearly) all its lanes -- measured
// in round 4 on the unsorted kernel: 31 of 64 lanes enabled per VALU instruction; on C4 71 % of the waves ran the Phong path (four binary64
// pow evaluations) for the 3 lanes that had hit the back wall, on C5 the glass and Phong paths ran at 10 of 64 lanes in 78 % of the waves.
// CLS_CHEAP: the path ends here (not alive, a miss, a light, no material): nothing is sampled, nothing is emitted.  The other classes are the
// surface materials; they survive the stage (need an output item) unless this is the last bounce of PT_STUPID -- shade_item reports an item
// that survives against the prediction (lost != nullptr), a needless yes only costs an idle item.
enum ItemCls : int32_t { CLS_LAMBERT = 0, CLS_PHONG = 1, CLS_GLASS = 2, CLS_MIRROR = 3, CLS_CHEAP = 4, kItemClasses = 5 };
// What the classification has already fetched for a SURFACE item (class < CLS_CHEAP): its hit key and its material index (in range, a
// surface material).  shade_item starts every load of the item from it at once instead of walking hit -> triangle record -> material.
struct ItemHint { uint32_t key; int32_t mat; };
// camera: 1 / 0 = the caller knows (k_shade_compact is compiled once for bounce 0 and once for the others, so that neither carries the
// other's loads and registers), -1 = look at the bank
ART_HD int32_t item_class(const DevScene& s, const DevPaths& qi, int w, const StageCtx& cx = StageCtx(), ItemHint* hint = nullptr, int camera_mode = -1) {
  const bool camera = (camera_mode >= 0) ? (camera_mode != 0) : (qi.synth0 && qi.slot_id == nullptr);         // raygen's bank: every item is a live camera ray (DevPaths::synth0)
  const uint32_t fl = camera ? (FLAG_ALIVE | FLAG_PREV_SPEC) : qi.flags[w];
  if (!(fl & FLAG_ALIVE)) return CLS_CHEAP;                       // only owed a shadow test: resolved now
  const uint32_t key = qi.hit[w].key;
  if (key == KEY_MISS) return CLS_CHEAP;
  const uint32_t cls = key & ~KEY_INDEX_MASK, idx = key & KEY_INDEX_MASK;
  int32_t mat;
  if (cls == KEY_SPHERE) mat = s.sphere_mat[idx];
  else if (cls == KEY_CORNELL) mat = s.cb_mat[idx];
  else if (cls == KEY_QUAD) mat = light_mat(s, cx, idx);
  else if (cls == KEY_BFTRI) mat = 2;
  else mat = __builtin_bit_cast(int32_t, s.m_shade[(size_t)kTriShadeFloats * shade_record(s, idx) + 9]);
  if (mat < 0 || mat >= s.n_materials) return CLS_CHEAP;
  const int32_t type = (cx.materials ? cx.materials : s.materials)[mat].type;
  if (hint) { hint->key = key; hint->mat = mat; }
  return (type == MAT_LAMBERT) ? CLS_LAMBERT : (type == MAT_PHONG) ? CLS_PHONG : (type == MAT_GLASS) ? CLS_GLASS : (type == MAT_MIRROR) ? CLS_MIRROR : CLS_CHEAP;
}
// item_class for N items at once, written so that the N independent chains of loads overlap (round 5).  item_class() returns early at
// every step, so the compiler emitted its four loads (flags, hit key -> triangle's shading record -> material type) as four waits one after
// the other, and the items of a thread one after the other: 20 memory round trips in a row per workgroup of k_shade_compact, during which
// the wave does nothing else (the stage spent 65 % of its wave-cycles in s_waitcnt, profiles/r4_final).  Here every step issues the loads of
// all N items with addresses made safe by selection (a lane without a triangle hit reads record 0's word; a lane without a material reads
// material 0), and looks at the values afterwards: three round trips.  Same results as item_class for every item.
// (device: an empty asm that names the N loaded values -- they must all have arrived there, so the N loads are issued before it, together)
template <class T, int N> ART_HD void pin_loads(T (&a)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (N == 4) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
  else if constexpr (N == 2) asm volatile("" : "+v"(a[0]), "+v"(a[1]));
  else if constexpr (N == 8) asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
  else for (int k = 0; k < N; ++k) asm volatile("" : "+v"(a[k]));
#else
  (void)a;
#endif
}
template <int N>
ART_HD void item_classes(const DevScene& s, const DevPaths& qi, const int (&w)[N], const bool (&on)[N], const StageCtx& cx, bool camera, int32_t (&cls_out)[N], ItemHint (&hint)[N]) {
  uint32_t fl[N], key[N];
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < N; ++k) {                                   // step 1: flags and hit keys
    const int wk = on[k] ? w[k] : 0;
    fl[k] = camera ? (FLAG_ALIVE | FLAG_PREV_SPEC) : at(hotf<const uint32_t>(qi, HF_FLAGS), wk);      // (cached: shade_item reads the line again)      // (k_shade_compact only: the record schedule's block)
    key[k] = ld_off(hotf<const uint32_t>(qi, HF_HIT), (uint32_t)wk * 16u + 4u);
  }
  if (!camera) pin_loads(fl);
  pin_loads(key);
  int32_t m_tri[N], m_sph[N]; bool surf[N], tri[N], sph[N];
  const float* const shade0 = s.m_shade ? s.m_shade : (const float*)(const void*)s.materials;      // (a scene without a BVH mesh: any readable word)
  const int32_t* const smat0 = s.sphere_mat ? s.sphere_mat : (const int32_t*)(const void*)s.materials;
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < N; ++k) {                                   // step 2: the material index (triangle: a word of its shading record; sphere: the sphere's)
    surf[k] = on[k] && (fl[k] & FLAG_ALIVE) && key[k] != KEY_MISS;
    const uint32_t c = key[k] & ~KEY_INDEX_MASK, idx = key[k] & KEY_INDEX_MASK;
    tri[k] = surf[k] && c == KEY_TRI; sph[k] = surf[k] && c == KEY_SPHERE;
    m_tri[k] = __builtin_bit_cast(int32_t, shade0[tri[k] ? (size_t)kTriShadeFloats * shade_record(s, idx) + 9 : (size_t)0]);
    m_sph[k] = smat0[sph[k] ? idx : 0u];
  }
  pin_loads(m_tri); pin_loads(m_sph);
  int32_t mat[N], type[N];
  const DevMaterial* const mats = cx.materials ? cx.materials : s.materials;
  const int32_t cb[6] = {s.cb_mat[0], s.cb_mat[1], s.cb_mat[2], s.cb_mat[3], s.cb_mat[4], s.cb_mat[5]};
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < N; ++k) {
    const uint32_t c = key[k] & ~KEY_INDEX_MASK, idx = key[k] & KEY_INDEX_MASK;
    int32_t m = tri[k] ? m_tri[k] : m_sph[k];
    // the Cornell box's six materials are scene header words (wave-uniform: selected, no memory access per item -- a wall is what most
    // rays of the Cornell scenes end on); the reference's brute-force mesh has material 2 (geometry.adb:311)
    const int32_t m_cb = (idx == 0u) ? cb[0] : (idx == 1u) ? cb[1] : (idx == 2u) ? cb[2] : (idx == 3u) ? cb[3] : (idx == 4u) ? cb[4] : cb[5];
    m = (c == KEY_CORNELL) ? m_cb : (c == KEY_BFTRI) ? 2 : m;
    if (surf[k] && c == KEY_QUAD) m = light_mat(s, cx, idx);      // a rect light (rare): the lights table
    mat[k] = m;
    surf[k] = surf[k] && m >= 0 && m < s.n_materials;
  }
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < N; ++k) type[k] = mats[surf[k] ? mat[k] : 0].type;      // step 3: the material's type (a loop of its own: no branch between the N loads)
  pin_loads(type);
#if defined(__HIPCC__)
#pragma unroll
#endif
  for (int k = 0; k < N; ++k) {
    int32_t c = CLS_CHEAP;
    if (surf[k]) {
      hint[k].key = key[k]; hint[k].mat = mat[k];
      c = (type[k] == MAT_LAMBERT) ? CLS_LAMBERT : (type[k] == MAT_PHONG) ? CLS_PHONG : (type[k] == MAT_GLASS) ? CLS_GLASS : (type[k] == MAT_MIRROR) ? CLS_MIRROR : CLS_CHEAP;
    }
    cls_out[k] = on[k] ? c : (int32_t)kItemClasses;
  }
}
ART_HD bool stage_keeps_surfaces(const DevFrame& f, int bounce) { return (f.render_type != PT_STUPID) || (bounce + 1 < f.max_depth); }
ART_HD bool item_survives(const DevFrame& f, const DevScene& s, const DevPaths& qi, int w, int bounce, const StageCtx& cx = StageCtx()) {
  return item_class(s, qi, w, cx) != CLS_CHEAP && stage_keeps_surfaces(f, bounce);
}

// Returns the number of rays the item emits (the closest-hit queries of the next trace: Mrays/s counts them).
// defer != nullptr (and qo.rec): the trace records are left to the caller, who gets the rays in *defer.
// hint != nullptr: the item is a surface item and *hint holds its hit key and material index (k_shade_compact's classification): the
// triangle's normals and the material record are then requested together with the item's own words, one round trip instead of three
// dependent ones (hit -> triangle shading record -> material).
template <int MATS = kMatsAll>
ART_HD int shade_item(const DevFrame& f, const DevScene& s, const DevPaths& qi, const DevPaths& qo, int w, int wo, int bounce, unsigned long long* lost = nullptr,
                      const StageCtx& cx = StageCtx(), RayOut* defer = nullptr, const ItemHint* hint_in = nullptr, int camera_mode = -1, int dense_mode = -1) {
  ART_PROBE(0);
  // dense_mode: 1 / 0 = the caller knows which fold records the schedule keeps (k_shade_compact: dense; the dead branches and their
  // pointer loads then drop out of the kernel), -1 = look at the bank
  const bool dense = (dense_mode >= 0) ? (dense_mode != 0) : (qi.fold_dense != 0);
  const bool batch = (dense_mode == 1);                // k_shade_compact: hints exist, the record schedule's blocks exist (DevPaths::hot / cold)
  const int slot_loaded = batch ? ((camera_mode == 1) ? w : (int)at_s(hotf<const uint32_t>(qi, HF_SLOT), w)) : item_slot(qi, w);
  const size_t P = (size_t)qi.P;
  const bool camera = (camera_mode >= 0) ? (camera_mode != 0) : (qi.synth0 && qi.slot_id == nullptr);         // bounce 0 of the compacted schedule: raygen stored the hit and nothing else (DevPaths::synth0)
  uint32_t fl = camera ? (FLAG_ALIVE | FLAG_PREV_SPEC) : at_s(batch ? hotf<const uint32_t>(qi, HF_FLAGS) : qi.flags, w);
  // ---- everything the item holds is read first: ONE batch of independent loads (round 5).  Written as conditional loads the compiler sank
  // each into the branch that uses it and waited for them one by one -- eight memory round trips in a row at the head of every item, at the
  // stage's 80-VGPR cap -- so the loads a hint makes possible are unconditional (an item without a hint reads record 0 / material 0 and
  // ignores them), and on the device an empty asm that names every loaded value pins them all before the first use.
  DevHit hw = at_s(batch ? hotf<const DevHit>(qi, HF_HIT) : qi.hit, w);
  // (hint_in may point at a record that says "no hint" (mat < 0): the caller then need not choose between a pointer and nullptr per lane --
  // which forced the record into scratch memory, with a scratch load and an s_waitcnt vmcnt(0) at each of its four uses, round 5)
  const bool hinted = (hint_in != nullptr) && (hint_in->mat >= 0);
  const ItemHint hint_v = hint_in ? *hint_in : ItemHint{KEY_MISS, -1};
  const uint32_t key = hinted ? hint_v.key : hw.key;
  PreNormals pre_n = PreNormals{{0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, false};
  DevMaterial pre_m = DevMaterial{MAT_NULL, 0, {0, 0, 0, 0, 0, 0, 0, 0}};
  if (hinted || batch) {
    pre_m = (cx.materials ? cx.materials : s.materials)[hinted ? hint_v.mat : 0];
    const bool tri = hinted && !(ART_DIAG_SKIP & 2) && (hint_v.key & ~KEY_INDEX_MASK) == KEY_TRI;
    if (tri || batch) {
      const float* r = (s.m_shade ? s.m_shade : (const float*)(const void*)s.materials) + (size_t)kTriShadeFloats * (tri ? shade_record(s, hint_v.key & KEY_INDEX_MASK) : (size_t)0);
      pre_n.a = ld3(r); pre_n.b = ld3(r + 3); pre_n.c = ld3(r + 6); pre_n.on = tri;
      if (s.n_inst > 0 && tri) {                                 // instanced scene: the record holds object-space normals
        const DevInstance& R = s.inst[(hint_v.key & KEY_INDEX_MASK) >> s.inst_shift];
        pre_n.a = instance_normal(R.minv, pre_n.a); pre_n.b = instance_normal(R.minv, pre_n.b); pre_n.c = instance_normal(R.minv, pre_n.c);
      }
    }
  }
  // the shadow test the item may owe (its words exist for every item; asked for now, used below if the flag says so)
  const bool may_owe = (bounce > 0) && (f.render_type != PT_STUPID);
  // (hs_t: t of the shadow ray's closest hit, -1 = none: the word sh_t[w] of the record schedule, else from the hit record at [P + w])
  float hs_t = -1.0f, owed_min = 0.0f; f3 owed = mk3(0.0f, 0.0f, 0.0f);
  auto shadow_t = [&]() { if (batch) return at_s(hotf<const float>(qi, HF_SHT), w); if (qi.sh_t) return at(qi.sh_t, w); const DevHit h = qi.hit[P + (size_t)w]; return (h.key != KEY_MISS) ? h.t : -1.0f; };
  if (may_owe || (batch && !camera)) { hs_t = shadow_t(); owed_min = at_s(batch ? hotf<const float>(qi, HF_SHMIN) : qi.sh_min_t, w); if (!dense) owed = mk3(qi.cand_r[w], qi.cand_g[w], qi.cand_b[w]); }
  // (the extension ray is kept as six SoA words next to its trace record: reading it back out of the record would pull the whole
  // 128-byte line of the item's two records for 24 useful bytes)
  f3 o, d; float prev_pdf;
  int slot_v = slot_loaded;
  if (camera) {
    o = ld3(s.cam_pos); prev_pdf = 1.0f;
    if (ART_SYNTH_DIR) {
      uint32_t cpix, csam;
      slot_to_sample(qi, slot_loaded, cpix, csam);
      d = camera_dir(f, s, cpix, csam);                                         // raygen_slot's own expression: the same bits
    } else d = mk3(at(qi.ray_dx, w), at(qi.ray_dy, w), at(qi.ray_dz, w));
  } else {
    if (batch) {
      o = mk3(at_s(hotf<const float>(qi, HF_OX), w), at_s(hotf<const float>(qi, HF_OY), w), at_s(hotf<const float>(qi, HF_OZ), w));
      d = mk3(at_s(hotf<const float>(qi, HF_DX), w), at_s(hotf<const float>(qi, HF_DY), w), at_s(hotf<const float>(qi, HF_DZ), w));
      prev_pdf = at_s(hotf<const float>(qi, HF_PDF), w);
    } else {
      o = mk3(at(qi.ray_ox, w), at(qi.ray_oy, w), at(qi.ray_oz, w));
      d = mk3(at(qi.ray_dx, w), at(qi.ray_dy, w), at(qi.ray_dz, w));
      prev_pdf = at(qi.prev_pdf, w);
    }
  }
#if defined(__HIP_DEVICE_COMPILE__)
  if (batch) {
    // (input operands only: tied in-out operands made the register allocator copy the loaded values around, and it put those copies --
    // hence a wait -- between the first loads and the last)
    if (camera) {
      asm volatile("; the item's loads are in flight together" :: "v"(slot_v), "v"(hw.t), "v"(hw.key), "v"(hw.u), "v"(hw.v),
                   "v"(pre_n.a.x), "v"(pre_n.a.y), "v"(pre_n.a.z), "v"(pre_n.b.x), "v"(pre_n.b.y), "v"(pre_n.b.z), "v"(pre_n.c.x), "v"(pre_n.c.y), "v"(pre_n.c.z),
                   "v"(pre_m.type), "v"(pre_m.light), "v"(pre_m.p[0]), "v"(pre_m.p[1]), "v"(pre_m.p[2]), "v"(pre_m.p[3]), "v"(pre_m.p[6]));
    } else {
      asm volatile("; the item's loads are in flight together" :: "v"(slot_v), "v"(fl), "v"(hw.t), "v"(hw.key), "v"(hw.u), "v"(hw.v), "v"(hs_t), "v"(owed_min), "v"(prev_pdf),
                   "v"(o.x), "v"(o.y), "v"(o.z), "v"(d.x), "v"(d.y), "v"(d.z),
                   "v"(pre_n.a.x), "v"(pre_n.a.y), "v"(pre_n.a.z), "v"(pre_n.b.x), "v"(pre_n.b.y), "v"(pre_n.b.z), "v"(pre_n.c.x), "v"(pre_n.c.y), "v"(pre_n.c.z),
                   "v"(pre_m.type), "v"(pre_m.light), "v"(pre_m.p[0]), "v"(pre_m.p[1]), "v"(pre_m.p[2]), "v"(pre_m.p[3]));
    }
  }
#endif
  const int slot = slot_v;
  ART_TPROBE(cx.tprobe, 71);      // the item's loads have arrived
  const float t = hw.t, hu = hw.u, hv = hw.v;
  // dense fold record of this item at this level (DevPaths::fold_dense): by default "the path ended here with value 0"; rec_shadowed: the
  // shadow test this item resolves for its predecessor's explicit light came out "in shadow"
  f3 rec_w = mk3(0.0f, 0.0f, 0.0f); int32_t rec_child = -1; bool rec_shadowed = false;
  if (fl & FLAG_SHADOW_PENDING) {
    ART_PROBE(1);
    // Compute_Shadow: hit and t < maxDist - eps2 (enforced by the ray's tfar clip) and t > 10*eps
    if (!may_owe && !(batch && !camera)) { hs_t = shadow_t(); owed_min = at(qi.sh_min_t, w); if (!dense) owed = mk3(qi.cand_r[w], qi.cand_g[w], qi.cand_b[w]); }
    const bool in_shadow = (hs_t >= 0.0f) && (hs_t > owed_min);       // hit, and beyond 10 eps (a hit's t is positive; -1: no hit)
    if (dense) rec_shadowed = in_shadow;      // dense fold records: the previous stage left the explicit colour itself at e[bounce][w]; this item's record says whether it counts
    else {                                            // e of the previous level at [bounce - 1][slot]
      const size_t li = (size_t)(bounce - 1) * P + (size_t)slot;
      qi.e_r[li] = in_shadow ? 0.0f : owed.x;
      qi.e_g[li] = in_shadow ? 0.0f : owed.y;
      qi.e_b[li] = in_shadow ? 0.0f : owed.z;
    }
    fl &= ~FLAG_SHADOW_PENDING;
  }
  // ---- what the output item will hold
  bool alive = (fl & FLAG_ALIVE) != 0u, shadow = false, null_shadow = false;
  f3 no = o, nd = d, so = o, sd = d;
  float s_tfar = -1.0f, sh_min = 0.0f, new_pdf = prev_pdf;
  f3 cand = mk3(0.0f, 0.0f, 0.0f);
  const f3 zero = mk3(0.0f, 0.0f, 0.0f);
  auto kill = [&](int levels, f3 terminal) {       // the deepest PathTrace call returned `terminal`; `levels` fold levels were recorded
    fl = (fl & ~(FLAG_ALIVE | 0xffffff00u)) | ((uint32_t)levels << 8);
    if (dense) { if (levels == bounce) { rec_w = terminal; rec_child = -1; } }     // levels == bounce + 1: a surface at the last bounce, its record stands
    else {
      qi.term_r[slot] = terminal.x; qi.term_g[slot] = terminal.y; qi.term_b[slot] = terminal.z;
      qi.final_flags[slot] = fl;
    }
    alive = false;
  };
  if (alive) {
    ART_PROBE(2);
    const Surface sf = (key != KEY_MISS) ? surface_at(s, o, d, t, key, hu, hv, cx, pre_n) : Surface{zero, -1, -1};
    const DevLight* const lights = cx.lights ? cx.lights : s.lights;      // (read only when the kernel gave no LDS copy)
    const bool mat_ok = hinted ? true : ((key != KEY_MISS) && sf.mat >= 0 && sf.mat < s.n_materials);
    const DevMaterial m = hinted ? pre_m : (mat_ok ? (cx.materials ? cx.materials : s.materials)[sf.mat] : DevMaterial{MAT_NULL, 0, {0, 0, 0, 0, 0, 0, 0, 0}});
    ART_PROBE(3);
    if (MATS != kMatsAll && mat_ok && !(MATS & mat_bit(m.type)) && lost != nullptr) {   // a material this instantiation was not compiled for: the caller's sort is broken
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(lost, 1ull);
#endif
    }
    if (!mat_ok || m.type == MAT_NULL) kill(bounce, zero);                            // integrators.adb:218-220
    else if ((MATS & mat_bit(MAT_LIGHT)) && m.type == MAT_LIGHT) {                    // :102-108 / :155-157 / :222-247
      ART_PROBE(4);
      const f3 n = sf.normal;
      const float sel_pdf = 1.0f / (float)s.n_lights;
      f3 out = zero;
      if (f.render_type != PT_SHADOW && !(dot(neg(d), n) < 0.0f)) {
        const f3 emit = (m.light >= 0 && m.light < s.n_lights) ? (cx.lds_lights ? ld3(cx.lds_lights[m.light].intensity) : ld3(lights[m.light].intensity)) : zero;
        if (f.render_type == PT_STUPID) out = emit;
        else {
          float mis = 1.0f;
          if (!(fl & FLAG_PREV_SPEC)) {
            const float lp = (cx.lds_lights ? light_eval_pdf(cx.lds_lights + m.light, o, d, t) : light_eval_pdf(lights + m.light, o, d, t)) * sel_pdf;
            const float bp = prev_pdf;
            mis = bp * bp / (lp * lp + bp * bp);
          }
          out = emit * mis;
        }
      }
      kill(bounce, out);
    } else {
      const f3 n = sf.normal;
      ART_PROBE(5);
      const float sel_pdf = 1.0f / (float)s.n_lights;
      uint32_t pixel, sample;
      slot_to_sample(qi, slot, pixel, sample);
      const u4 rnd = philox4x32_10(pixel, sample, (uint32_t)bounce, 0u, f.seed_lo, f.seed_hi);
      const f3 hpos = o + d * t;
      const size_t li = (size_t)bounce * P + (size_t)slot;      // this level's record by slot (fold_dense = 0)
      if (f.render_type != PT_STUPID) {                                                // explicit light sampling :159-178 / :251-287
        int light = 0;
        if (s.n_lights > 1) {
          const u4 r1 = philox4x32_10(pixel, sample, (uint32_t)bounce, 1u, f.seed_lo, f.seed_hi);
          light = (int)(u01(r1.x) * (float)s.n_lights);
          if (light > s.n_lights - 1) light = s.n_lights - 1;
        }
        ART_PROBE(6);
        ART_TPROBE(cx.tprobe, 72);  // surface + material + philox done
        const LightSample ls = cx.lds_lights ? light_sample(cx.lds_lights + light, u01(rnd.x), u01(rnd.y), hpos) : light_sample(lights + light, u01(rnd.x), u01(rnd.y), hpos);
        const f3 sdir = normalize(ls.pos - hpos);
        const float lp = ls.pdf * sel_pdf;
        ART_PROBE(7);
        f3 bx; float bp;
        bsdf_eval<MATS>(m, sdir, neg(d), n, bx, bp);
        const float c1 = amax(dot(sdir, n), 0.0f);
        if (f.render_type == PT_MIS) {
          const float mis = lp * lp / (lp * lp + bp * bp);
          cand = ((ls.intensity * (1.0f / amax(lp, kGEpsilonDiv))) * (c1 * bx)) * mis;
        } else {
          cand = (ls.intensity * (c1 * bx)) * (1.0f / amax(lp, kGEpsilonDiv));
        }
        // Compute_Shadow (ray_tracer.adb:100-132): the closest hit of this ray decides visibility
        float eps = amax3(fabsf(hpos.x), fabsf(hpos.y), fabsf(hpos.z)) * 0.000000001f;
        eps = amax(eps, 1.0e-30f);
        sd = normalize(ls.pos - hpos);
        so = hpos + sd * eps;
        const float max_dist = length(hpos - ls.pos);
        const float eps2 = amax(max_dist * 0.000001f, 1.0e-30f);
        // a NaN / negative bound can never be "in shadow": emit the ray with an empty interval
        const float bound = max_dist - eps2;
        s_tfar = (bound > 0.0f) ? bound : 0.0f;
        sh_min = 10.0f * eps;
        // (DevFrame::skip_null_shadow: a candidate of exactly zero stays zero under either verdict; NaN compares unequal and is traced)
        null_shadow = f.skip_null_shadow && cand.x == 0.0f && cand.y == 0.0f && cand.z == 0.0f;
        shadow = !null_shadow;
        if (shadow) fl |= FLAG_SHADOW_PENDING;
        else if (!dense) { qi.e_r[li] = cand.x; qi.e_g[li] = cand.y; qi.e_b[li] = cand.z; }      // (nobody will resolve it: the level's explicit light is this zero)
      } else if (!dense) {
        qi.e_r[li] = 0.0f; qi.e_g[li] = 0.0f; qi.e_b[li] = 0.0f;
      }
      ART_PROBE(8);
      ART_TPROBE(cx.tprobe, 73);    // light sample, bsdf_eval, shadow ray done
      const BsdfSample bs = bsdf_sample<MATS>(m, u01(rnd.z), u01(rnd.w), d, n);          // :116-124 / :183-191 / :291-299
      ART_PROBE(9);
      const f3 bxv = bs.color * (1.0f / amax(bs.pdf, kGEpsilonDiv));
      const float ct = dot(bs.dir, n);
      no = hpos + (asign(ct) * n) * kGEpsilon;
      nd = bs.dir;
      const f3 wv = fabsf(ct) * bxv;
      if (dense) { rec_w = wv; rec_child = -2; }          // a surface: its successor's index is known below
      else { qi.w_r[li] = wv.x; qi.w_g[li] = wv.y; qi.w_b[li] = wv.z; }
      new_pdf = bs.pdf;
      fl = bs.specular ? (fl | FLAG_PREV_SPEC) : (fl & ~FLAG_PREV_SPEC);
      if (bounce + 1 >= f.max_depth) kill(bounce + 1, zero);                             // next level returns 0 untraced (:212-214)
    }
  }
  ART_PROBE(40);
  ART_TPROBE(cx.tprobe, 74);        // bsdf_sample done (all lanes)
  if (dense && !(ART_DIAG_SKIP & 16)) {                      // dense stores: consecutive items, consecutive addresses
    const size_t l0 = (size_t)bounce * P;                   // (the level's base is wave-uniform: a scalar add; the item's offset stays 32 bits)
    const int32_t cw = fold_child_word((rec_child == -2 && wo >= 0) ? wo : rec_child, rec_shadowed);
    if (batch) { put_s(cold_w(qi, 0, bounce), w, rec_w.x); put_s(cold_w(qi, 1, bounce), w, rec_w.y); put_s(cold_w(qi, 2, bounce), w, rec_w.z); put_s(cold_child(qi, bounce), w, cw); }
    else { put(qi.w_r + l0, w, rec_w.x); put(qi.w_g + l0, w, rec_w.y); put(qi.w_b + l0, w, rec_w.z); put(qi.child + l0, w, cw); }
  }
  // ---- the output item
  if (wo < 0) {
    if (lost != nullptr && (alive || shadow)) {
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(lost, 1ull);
#endif
    }
    return 0;
  }
  ART_PROBE(41);
  const size_t so_i = (size_t)qo.P + (size_t)wo;
  if (batch) {                                           // the record schedule: every word through the output bank's ONE base pointer
    put_s(hotf<uint32_t>(qo, HF_SLOT), wo, (uint32_t)slot);
    put_s(hotf<uint32_t>(qo, HF_FLAGS), wo, fl);
    put_s(hotf<float>(qo, HF_PDF), wo, new_pdf);
    if (shadow) put_s(hotf<float>(qo, HF_SHMIN), wo, sh_min);
    if (shadow || null_shadow) {      // (null_shadow: the ray is not traced, DevFrame::skip_null_shadow -- the level's explicit light is still this exact zero)
      put_s(cold_e(qi, 0, bounce + 1), wo, cand.x); put_s(cold_e(qi, 1, bounce + 1), wo, cand.y); put_s(cold_e(qi, 2, bounce + 1), wo, cand.z);
    }
    if (alive && !(ART_DIAG_SKIP & 4)) {
      put_s(hotf<float>(qo, HF_OX), wo, no.x); put_s(hotf<float>(qo, HF_OY), wo, no.y); put_s(hotf<float>(qo, HF_OZ), wo, no.z);
      put_s(hotf<float>(qo, HF_DX), wo, nd.x); put_s(hotf<float>(qo, HF_DY), wo, nd.y); put_s(hotf<float>(qo, HF_DZ), wo, nd.z);
    }
  } else {
  if (qo.slot_id != nullptr) put(const_cast<uint32_t*>(qo.slot_id), wo, (uint32_t)slot);      // compacted banks: the output item's slot
  put(qo.flags, wo, fl);
  put(qo.prev_pdf, wo, new_pdf);
  if (shadow || null_shadow) {
    if (shadow) put(qo.sh_min_t, wo, sh_min);
    if (dense) {      // the explicit colour goes straight to where the fold reads e of this level: level bounce + 1, the successor's index
      const size_t l1 = (size_t)(bounce + 1) * P;
      put(qi.e_r + l1, wo, cand.x); put(qi.e_g + l1, wo, cand.y); put(qi.e_b + l1, wo, cand.z);
    } else if (shadow) { qo.cand_r[wo] = cand.x; qo.cand_g[wo] = cand.y; qo.cand_b[wo] = cand.z; }
  }
  if (alive && !(ART_DIAG_SKIP & 4)) {
    put(qo.ray_ox, wo, no.x); put(qo.ray_oy, wo, no.y); put(qo.ray_oz, wo, no.z);
    put(qo.ray_dx, wo, nd.x); put(qo.ray_dy, wo, nd.y); put(qo.ray_dz, wo, nd.z);
  }
  }
  ART_TPROBE(cx.tprobe, 75);        // fold record + output words stored
  if (qo.rec) {
    // the item's rays go out as trace records, at positions given by the item index (REC_BOTH: 2 wo and 2 wo + 1).  A ray the bank's
    // mode has no record for cannot exist (the modes follow the integrator: art_api.cpp); should it ever, the self-check counts it.
    const int mode = qo.rec_mode;
    if (defer) {
      defer->alive = alive; defer->shadow = shadow; defer->no = no; defer->nd = nd; defer->so = so; defer->sd = sd; defer->s_tfar = s_tfar; defer->sh_min = sh_min;
    } else {
      const int per = (mode == REC_BOTH) ? 2 : 1;
      const bool blocks = cx.stage_count > 0;      // k_shade_compact: each kind of ray is its own contiguous block of the wave's records
      if (mode != REC_SHADOW) emit_ray(s, qo, (size_t)wo, rec_slot(mode, wo, false), alive, no, nd, kInfinity, -1.0f, cx, blocks ? cx.stage_item : per * cx.stage_item, 0);
      ART_TPROBE(cx.tprobe, 76);    // extension ray's record out
      if (mode != REC_EXT) emit_ray(s, qo, (batch || qo.sh_t) ? (size_t)(kShadowWord | (uint32_t)wo) : so_i, rec_slot(mode, wo, true), shadow, so, sd, s_tfar, qo.shadow_rule ? sh_min : -1.0f, cx, blocks ? cx.stage_item : per * cx.stage_item + (per - 1), 1);
    }
    ART_TPROBE(cx.tprobe, 77);      // shadow ray's record out
    if (lost != nullptr && ((mode == REC_SHADOW && alive) || (mode == REC_EXT && shadow))) {
#if defined(__HIP_DEVICE_COMPILE__)
      atomicAdd(lost, 1ull);
#endif
    }
    return (alive ? 1 : 0) + (shadow ? 1 : 0);
  }
  qo.ray_tfar[wo] = alive ? kInfinity : -1.0f;
  qo.ray_tfar[so_i] = shadow ? s_tfar : -1.0f;
  if (shadow) {
    qo.ray_ox[so_i] = so.x; qo.ray_oy[so_i] = so.y; qo.ray_oz[so_i] = so.z;
    qo.ray_dx[so_i] = sd.x; qo.ray_dy[so_i] = sd.y; qo.ray_dz[so_i] = sd.z;
  }
  return (alive ? 1 : 0) + (shadow ? 1 : 0);
}

// the plain layout: one item per slot, updated in place
ART_HD void shade_slot(const DevFrame& f, const DevScene& s, const DevPaths& q, int slot, int bounce) { (void)shade_item(f, s, q, q, slot, slot, bounce); }

// after the last trace: resolve the shadow test item w still owes ...
ART_HD void resolve_last_shadow(const DevPaths& q, int w, int last_level) {
  const uint32_t fl = q.flags[w];
  if (!(fl & FLAG_SHADOW_PENDING)) return;
  const int slot = item_slot(q, w);
  float hs_t;
  if (q.sh_t) hs_t = q.sh_t[w];
  else { const DevHit hs = q.hit[(size_t)q.P + (size_t)w]; hs_t = (hs.key != KEY_MISS) ? hs.t : -1.0f; }
  const bool in_shadow = (hs_t >= 0.0f) && (hs_t > q.sh_min_t[w]);
  if (q.fold_dense) {           // the last stage left the explicit colour at e[last_level + 1][w]; there is no record of that level to carry the bit: zero it
    if (in_shadow) { const size_t li = (size_t)(last_level + 1) * (size_t)q.P + (size_t)w; q.e_r[li] = 0.0f; q.e_g[li] = 0.0f; q.e_b[li] = 0.0f; }
  } else {
    const size_t li = (size_t)last_level * (size_t)q.P + (size_t)slot;
    q.e_r[li] = in_shadow ? 0.0f : q.cand_r[w];
    q.e_g[li] = in_shadow ? 0.0f : q.cand_g[w];
    q.e_b[li] = in_shadow ? 0.0f : q.cand_b[w];
  }
  q.flags[w] = fl & ~FLAG_SHADOW_PENDING;
}

// ... and fold  L = e_k + w_k * L  from the deepest level out (the levels a path recorded are in the flags word it ended with)
ART_HD void fold_slot(const DevFrame& f, const DevPaths& q, int slot) {
  const int levels = (int)(q.final_flags[slot] >> 8);
  f3 L = mk3(q.term_r[slot], q.term_g[slot], q.term_b[slot]);
  for (int k = levels - 1; k >= 0; --k) {
    const size_t li = (size_t)k * q.P + slot;
    const f3 w = mk3(q.w_r[li], q.w_g[li], q.w_b[li]);
    if (f.render_type == PT_STUPID) L = w * L;
    else L = mk3(q.e_r[li], q.e_g[li], q.e_b[li]) + w * L;
  }
  q.rad_r[slot] = L.x; q.rad_g[slot] = L.y; q.rad_b[slot] = L.z;
}

// The same fold over dense records (DevPaths::fold_dense), one level per call from the deepest up: item w of bounce `level` gets
//   L = its path's terminal value (kind 1),  or  e_level + w_level * L(child)   -- L(child) = 0 at the deepest level and without a child,
//   e_level = 0 when the child's record says its shadow test failed --
// read from `nxt` (the level below, indexed by that level's items) and written to `cur`.  Level 0's items are the slots: its `cur` is rad.
// The operations on a path's values are those of fold_slot in the same order, so the bits are the same.
ART_HD void fold_level_item(const DevFrame& f, const DevPaths& q, int level, int w, bool deepest,
                            const float* nxt_r, const float* nxt_g, const float* nxt_b, float* cur_r, float* cur_g, float* cur_b) {
  const size_t lw = (size_t)level * (size_t)q.P + (size_t)w;
  const int32_t cw = q.child[lw];
  const int32_t kind = (cw >> 28) & 3, c = cw & kFoldIndexMask;
  f3 L = mk3(q.w_r[lw], q.w_g[lw], q.w_b[lw]);
  if (kind != 1) {
    const f3 wv = L;
    const bool has = (kind == 0);
    const f3 Ln = (has && !deepest) ? mk3(nxt_r[c], nxt_g[c], nxt_b[c]) : mk3(0.0f, 0.0f, 0.0f);
    if (f.render_type == PT_STUPID) L = wv * Ln;
    else {
      const size_t le = (size_t)(level + 1) * (size_t)q.P + (size_t)c;
      // e of this level sits at the successor's index one level down; whether it counts is in the successor's record (at the deepest
      // level k_resolve_last has zeroed it instead: there is no record below)
      const bool counts = has && (deepest || ((q.child[le] >> 30) & 1) == 0);
      const f3 ev = counts ? mk3(q.e_r[le], q.e_g[le], q.e_b[le]) : mk3(0.0f, 0.0f, 0.0f);
      L = ev + wv * Ln;
    }
  }
  cur_r[w] = L.x; cur_g[w] = L.y; cur_b[w] = L.z;
}

ART_HD void finish_slot(const DevFrame& f, const DevPaths& q, int slot, int last_level) {     // plain layout: both steps per slot
  resolve_last_shadow(q, slot, last_level);
  fold_slot(f, q, slot);
}

// DoPass accumulation (integrators.adb:42-52 / :60-64) for one local pixel, in sample order
ART_HD void accumulate_pixel(const DevFrame& f, const DevPaths& q, int pl, int samples_in_batch, float* accum /* row-major rgb */) {
  const uint32_t pixel = q.pixmap ? q.pixmap[pl] : (uint32_t)pl;
  float* a = accum + 3 * (size_t)pixel;
  f3 acc = mk3(a[0], a[1], a[2]);
  if (f.aa_on) {
    for (int t = 0; t + 3 < samples_in_batch; t += 4) {
      f3 color = ld3(f.background);
      for (int i = 0; i < 4; ++i) {
        const size_t si = (size_t)(t + i) * q.npix + pl;
        color = color + mk3(q.rad_r[si], q.rad_g[si], q.rad_b[si]);
      }
      acc = color + acc;
    }
  } else {
    for (int t = 0; t < samples_in_batch; ++t) {
      const size_t si = (size_t)t * q.npix + pl;
      acc = mk3(q.rad_r[si], q.rad_g[si], q.rad_b[si]) + acc;
    }
  }
  a[0] = acc.x; a[1] = acc.y; a[2] = acc.z;
}

// resolve (ray_tracer.adb:281-291, :19-57): gamma 2 -> clamp -> round-to-nearest pack R | G<<8 | B<<16
ART_HD uint32_t ada_round_u32(float v) {
  if (!(v > 0.0f)) return 0u;
  uint32_t u = (uint32_t)v;
  if (v - (float)u >= 0.5f) u += 1u;
  return u;
}

ART_HD uint32_t resolve_pixel(f3 acc, float norm_c) {
  f3 rgb = acc * norm_c;
  rgb.x = apow(rgb.x, 1.0f / 2.0f); rgb.y = apow(rgb.y, 1.0f / 2.0f); rgb.z = apow(rgb.z, 1.0f / 2.0f);
  const uint32_t r = ada_round_u32(amin(rgb.x, 1.0f) * 255.0f);
  const uint32_t g = ada_round_u32(amin(rgb.y, 1.0f) * 255.0f);
  const uint32_t b = ada_round_u32(amin(rgb.z, 1.0f) * 255.0f);
  return r | (g << 8) | (b << 16);
}

ART_HD f3 debug_palette(int32_t mat_id) {   // ray_tracer.adb:210-211
  switch (mat_id % 8) {
    case 0: return mk3(0.5f, 0.0f, 0.0f);   case 1: return mk3(0.0f, 0.5f, 0.0f);
    case 2: return mk3(0.0f, 0.0f, 0.5f);   case 3: return mk3(0.5f, 0.5f, 0.5f);
    case 4: return mk3(0.5f, 0.5f, 0.0f);   case 5: return mk3(0.5f, 0.0f, 0.5f);
    case 6: return mk3(0.0f, 0.5f, 0.5f);   default: return mk3(0.75f, 0.75f, 0.75f);
  }
}

}  // namespace art

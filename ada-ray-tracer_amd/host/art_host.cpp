// art_host.cpp -- see art_host.hpp.  Scene setup uses the same host+device math header as the kernels
// (csrc/art_math.h) so that e.g. RotationMatrix(-Pi/6) yields the bits the rest of the pipeline expects.
#include "art_host.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

#include "../csrc/art_math.h"

namespace art_host {

using art::f3;

void MatMul(const float a[16], const float b[16], float out[16]) {
  float t[16];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      t[4 * i + j] = a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j] + a[4 * i + 3] * b[12 + j];
  std::memcpy(out, t, sizeof t);
}

static void Identity(float M[16]) { std::memset(M, 0, 64); M[0] = M[5] = M[10] = M[15] = 1.0f; }

void RotationMatrix(float angle, const float axis[3], float M[16]) {
  Identity(M);
  const f3 v = art::normalize(art::mk3(axis[0], axis[1], axis[2]));
  float sin_t, cos_t;
  art::asincos_m1(angle, sin_t, cos_t);
  M[0] = (1.0f - cos_t) * v.x * v.x + cos_t;
  M[1] = (1.0f - cos_t) * v.x * v.y - sin_t * v.z;
  M[2] = (1.0f - cos_t) * v.x * v.z + sin_t * v.y;
  M[4] = (1.0f - cos_t) * v.y * v.x + sin_t * v.z;
  M[5] = (1.0f - cos_t) * v.y * v.y + cos_t;
  M[6] = (1.0f - cos_t) * v.y * v.z - sin_t * v.x;
  M[8] = (1.0f - cos_t) * v.x * v.z - sin_t * v.y;
  M[9] = (1.0f - cos_t) * v.z * v.y + sin_t * v.x;
  M[10] = (1.0f - cos_t) * v.z * v.z + cos_t;
}

bool LoadMeshFromVSGF(Mesh& self, const float T[16], const std::string& a_fileName, std::string& err) {
  std::ifstream f(a_fileName, std::ios::binary);
  if (!f) { err = "cannot open " + a_fileName; return false; }
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  struct Header { int64_t fileSizeInBytes; int32_t verticesNum, indicesNum, materialsNum, flags; } h;   // geometry.adb:499-507
  if (raw.size() < sizeof h) { err = "VSGF: short header"; return false; }
  std::memcpy(&h, raw.data(), sizeof h);
  const int nv = h.verticesNum, nt = h.indicesNum / 3;
  const size_t need = 24 + (size_t)nv * 40 + (h.flags != 0 ? (size_t)nv * 16 : 0) + (size_t)nt * 16;
  if (nv <= 0 || nt <= 0 || raw.size() < need) { err = "VSGF: truncated file"; return false; }
  const char* p = raw.data() + 24;
  self.vert_positions.resize(3 * (size_t)nv); self.vert_normals.resize(3 * (size_t)nv); self.vert_tex_coords.assign(2 * (size_t)nv, 0.0f);
  for (int i = 0; i < nv; ++i) { std::memcpy(&self.vert_positions[3 * i], p, 12); p += 16; }   // float4 -> xyz
  for (int i = 0; i < nv; ++i) { std::memcpy(&self.vert_normals[3 * i], p, 12); p += 16; }
  p += (size_t)nv * 8;                                     // texcoords are read and forced to 0 (geometry.adb:565-566)
  if (h.flags != 0) p += (size_t)nv * 16;                  // tangents skipped
  self.triangles.resize(3 * (size_t)nt); std::memcpy(self.triangles.data(), p, 12 * (size_t)nt); p += 12 * (size_t)nt;
  self.material_ids.resize(nt); std::memcpy(self.material_ids.data(), p, 4 * (size_t)nt);
  for (int a = 0; a < 3; ++a) { self.bbox_min[a] = art::kInfinity; self.bbox_max[a] = -art::kInfinity; }
  for (int i = 0; i < nv; ++i) {                           // geometry.adb:593-607 (positions only; true bbox)
    const f3 v = art::xform_point(T, art::mk3(self.vert_positions[3 * i], self.vert_positions[3 * i + 1], self.vert_positions[3 * i + 2]));
    const float c[3] = {v.x, v.y, v.z};
    for (int a = 0; a < 3; ++a) {
      self.vert_positions[3 * i + a] = c[a];
      self.bbox_min[a] = art::amin(self.bbox_min[a], c[a]);
      self.bbox_max[a] = art::amax(self.bbox_max[a], c[a]);
    }
  }
  return true;
}

bool Scene::Init(const std::string& a_vsgf_path, std::string& err) {   // Init_Cornell_Box, scene.adb:89-217
  const float intensity[3] = {20.0f, 20.0f, 20.0f};
  lights.assign(1, ArtLight());
  ArtLight& L = lights[0];
  std::memset(&L, 0, sizeof L);
  L.shape = ART_LIGHT_SPHERE;                                          // g_lightRef := sphLight (:128)
  L.center[0] = 0.0f; L.center[1] = 4.5f; L.center[2] = 1.0f; L.radius = 0.5f;
  for (int a = 0; a < 3; ++a) L.intensity[a] = 0.5f * intensity[a];    // intensity*0.5 (:121)
  L.surfaceArea = 4.0f * art::kPi * L.radius * L.radius;               // :122
  L.mat = 4;
  auto mat = [](int type, std::initializer_list<float> p, int light = 0) {
    ArtMaterial m; std::memset(&m, 0, sizeof m); m.type = type; m.light = light;
    int i = 0; for (float v : p) m.p[i++] = v;
    return m;
  };
  materials.assign(11, mat(ART_MAT_NULL, {}));                         // :169-180 (6, 7 stay null)
  materials[0] = mat(ART_MAT_GLASS, {0.75f, 0.75f, 0.75f, 0.85f, 0.85f, 0.85f, 1.75f});
  materials[1] = mat(ART_MAT_LAMBERT, {0.5f, 0.5f, 0.5f});
  materials[2] = mat(ART_MAT_LAMBERT, {0.25f, 0.5f, 0.0f});
  materials[3] = mat(ART_MAT_LAMBERT, {0.5f, 0.0f, 0.0f});
  materials[4] = mat(ART_MAT_LIGHT, {}, 0);
  materials[5] = mat(ART_MAT_MIRROR, {0.75f, 0.75f, 0.75f});
  materials[8] = mat(ART_MAT_PHONG, {0.75f, 0.75f, 0.75f, 80.0f});
  materials[9] = materials[1]; materials[10] = materials[1];
  auto sph = [](float x, float y, float z, float r, int m) { ArtSphere s; s.pos[0] = x; s.pos[1] = y; s.pos[2] = z; s.r = r; s.mat = m; return s; };
  spheres = {sph(-1.5f, 1.0f, 1.5f, 1.0f, 8), sph(1.4f, 1.0f, 3.0f, 1.0f, 0), sph(0.0f, 4.5f, 1.0f, 0.5f, 4)};   // :142-144, :182-192
  float mrot[16], mscale[16], mtans[16], tmp[16], T[16];               // :194-206
  const float axis[3] = {0.0f, 1.0f, 0.0f};
  RotationMatrix(-0x1.0c1524p-1f /* static -PI/6.0 */, axis, mrot);
  Identity(mscale); Identity(mtans);
  mtans[3] = -0.75f; mtans[7] = 0.1f; mtans[11] = 3.1f; mtans[15] = 1.0f;
  mscale[0] = mscale[5] = mscale[10] = 2.0f;
  MatMul(mtans, mrot, tmp); MatMul(tmp, mscale, T);
  if (!LoadMeshFromVSGF(mymesh, T, a_vsgf_path, err)) return false;
  std::memset(&mesh_desc, 0, sizeof mesh_desc);
  mesh_desc.mode = ART_MESH_REFERENCE_BF;
  mesh_desc.nverts = (int32_t)(mymesh.vert_positions.size() / 3); mesh_desc.ntris = (int32_t)(mymesh.triangles.size() / 3);
  mesh_desc.pos = mymesh.vert_positions.data(); mesh_desc.nrm = mymesh.vert_normals.data(); mesh_desc.uv = mymesh.vert_tex_coords.data();
  mesh_desc.idx = mymesh.triangles.data(); mesh_desc.matid = mymesh.material_ids.data();
  std::memcpy(mesh_desc.bbmin, mymesh.bbox_min, 12); std::memcpy(mesh_desc.bbmax, mymesh.bbox_max, 12);
  std::memset(&desc, 0, sizeof desc);
  desc.n_spheres = (int32_t)spheres.size(); desc.spheres = spheres.data();
  desc.has_cornell = 1;                                                // scene.ads:75-80
  const float bmin[3] = {-2.5f, 0.0f, 0.0f}, bmax[3] = {2.5f, 5.0f, 5.0f};
  std::memcpy(desc.cb_min, bmin, 12); std::memcpy(desc.cb_max, bmax, 12);
  const int32_t mi[6] = {2, 3, 1, 1, 8, 1}; std::memcpy(desc.cb_mat, mi, sizeof mi);
  const float nn[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}}; std::memcpy(desc.cb_nrm, nn, sizeof nn);
  desc.n_lights = 1; desc.lights = lights.data();
  desc.n_materials = (int32_t)materials.size(); desc.materials = materials.data();
  desc.n_meshes = 1; desc.meshes = &mesh_desc;
  desc.cam_pos[0] = 0.0f; desc.cam_pos[1] = 2.55f; desc.cam_pos[2] = 12.5f;   // :212
  Identity(desc.cam_matrix);                                                   // :215
  return true;
}

bool Ray_Tracer::Init_Render(Render_Type a_rendType) { g_rend_type = a_rendType; return true; }

bool Ray_Tracer::Resize_Viewport(int size_x, int size_y) {
  width = size_x; height = size_y;
  screen_buffer.assign((size_t)width * height, 0u);
  g_accBuff.assign(3 * (size_t)width * height, 0.0f);
  g_spp = 0;
  if (art_resize(width, height)) { last_error = art_last_error(); return false; }
  return true;
}

bool Ray_Tracer::Render_Pass() {
  ArtPassParams p; std::memset(&p, 0, sizeof p);
  p.render_type = (int32_t)g_rend_type; p.aa_on = Anti_Aliasing_On ? 1 : 0; p.max_depth = Max_Trace_Depth; p.vthreads = Threads_Num;
  std::memcpy(p.background, Background_Color, 12); p.seed = seed; p.layout = ART_LAYOUT_ADA_XY;
  int rc;
  if (g_rend_type == RT_DEBUG || g_rend_type == RT_WHITTED) {           // ray_tracer.adb:245-261
    rc = art_debug_hit_pass(&p, g_accBuff.data(), screen_buffer.data(), nullptr, nullptr, nullptr);
    g_finish = true;
  } else {
    rc = art_render_pass(&p, g_accBuff.data(), screen_buffer.data(), &g_spp);   // :264-291
  }
  if (rc) { last_error = art_last_error(); return false; }
  return true;
}

void Bitmap_Init(Image& im, int w, int h) { im.width = w; im.height = h; im.data.assign((size_t)w * h, 0u); }

std::vector<uint8_t> BMPBytes(const Image& im) {   // bitmap.adb:31-85: 14 + 40 byte headers field by field, then r,g,b = bits 16-23, 8-15, 0-7
  std::vector<uint8_t> out;
  auto put16 = [&](uint32_t v) { out.push_back(v & 255); out.push_back((v >> 8) & 255); };
  auto put32 = [&](uint32_t v) { put16(v & 0xffff); put16(v >> 16); };
  put16(0x4d42); put32(14 + 40 + (uint32_t)(im.width * im.height * 3)); put16(0); put16(0); put32(14 + 40);
  put32(40); put32((uint32_t)im.width); put32((uint32_t)im.height); put16(1); put16(24);
  put32(0); put32(0); put32(0); put32(0); put32(0); put32(0);
  for (uint32_t px : im.data) { out.push_back((px >> 16) & 255); out.push_back((px >> 8) & 255); out.push_back(px & 255); }
  return out;
}

bool SaveBMP(const Image& im, const std::string& a_fileName) {
  const std::vector<uint8_t> b = BMPBytes(im);
  std::ofstream f(a_fileName, std::ios::binary);
  if (!f) return false;
  f.write(reinterpret_cast<const char*>(b.data()), (std::streamsize)b.size());
  return (bool)f;
}

}  // namespace art_host

// C entry point used by the tests to compare the C++ Scene.Init with the oracle's (no GPU needed)
extern "C" int art_host_cornell_scene(const char* vsgf_path, float* spheres5 /*3x5*/, float* light16, float* materials10 /*11x10*/,
                                      float* mesh_pos /*cap 64x3*/, float* mesh_bbox6, int* counts4) {
  static art_host::Scene sc;
  std::string err;
  if (!sc.Init(vsgf_path, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
  for (size_t i = 0; i < sc.spheres.size(); ++i) { std::memcpy(spheres5 + 5 * i, sc.spheres[i].pos, 12); spheres5[5 * i + 3] = sc.spheres[i].r; spheres5[5 * i + 4] = (float)sc.spheres[i].mat; }
  const ArtLight& L = sc.lights[0];
  light16[0] = (float)L.shape; light16[1] = (float)L.mat; std::memcpy(light16 + 2, L.center, 12); light16[5] = L.radius; std::memcpy(light16 + 6, L.intensity, 12); light16[9] = L.surfaceArea;
  for (size_t i = 0; i < sc.materials.size(); ++i) { materials10[10 * i] = (float)sc.materials[i].type; materials10[10 * i + 1] = (float)sc.materials[i].light; std::memcpy(materials10 + 10 * i + 2, sc.materials[i].p, 32); }
  const size_t nv = sc.mymesh.vert_positions.size() / 3;
  if (nv > 64) return 2;
  std::memcpy(mesh_pos, sc.mymesh.vert_positions.data(), nv * 12);
  std::memcpy(mesh_bbox6, sc.mymesh.bbox_min, 12); std::memcpy(mesh_bbox6 + 3, sc.mymesh.bbox_max, 12);
  counts4[0] = (int)sc.spheres.size(); counts4[1] = (int)sc.materials.size(); counts4[2] = (int)nv; counts4[3] = (int)(sc.mymesh.triangles.size() / 3);
  return 0;
}

// The internal scene (Scene.Init, scene.adb:24-27 -> Init_Cornell_Box) as an ArtSceneDesc for callers that drive the C ABI directly
// (the Python layer, bench.py --scene c2): the product describes its own workload, no test code involved.  Returns nullptr on failure
// (message on stderr).  The descriptor and everything it points to stay valid until the next call.
extern "C" const ArtSceneDesc* art_host_scene_init(const char* vsgf_path) {
  static art_host::Scene sc;
  std::string err;
  if (!vsgf_path || !sc.Init(vsgf_path, err)) { std::fprintf(stderr, "art_host_scene_init: %s\n", err.c_str()); return nullptr; }
  return &sc.desc;
}

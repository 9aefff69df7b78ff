// art_host.hpp -- C++ mirror of the reference's Ada host side, above the C ABI (include/art_hip.h).
// The reference is compiled Ada with no toolchain in this image, so the host layer that an Ada maintainer
// would write (ada/*.ad?) is mirrored here in C++ with the same names, argument meaning and calling sequence:
//   Scene.Init / Init_Cornell_Box      scene.adb:24-27, 89-217     -> art_host::Scene
//   Geometry.LoadMeshFromVSGF          geometry.adb:499-609        -> art_host::LoadMeshFromVSGF
//   Ray_Tracer.Init_Render / Resize_Viewport / Render_Pass / GetSPP / Finished   ray_tracer.ads:40-48
//   Bitmap.Init / SaveBMP              bitmap.ads:21-25, bitmap.adb:31-85
// test_main.cpp replays test.adb:20-79 on top of it.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/art_hip.h"

namespace art_host {

struct Mesh {                       // geometry.ads:94-101
  std::vector<float> vert_positions, vert_normals, vert_tex_coords;
  std::vector<int32_t> triangles, material_ids;
  float bbox_min[3], bbox_max[3];
};

// geometry.adb:499-609: positions transformed by mTransform (row-major 4x4), normals untouched, texcoords zeroed
bool LoadMeshFromVSGF(Mesh& self, const float mTransform[16], const std::string& a_fileName, std::string& err);
void RotationMatrix(float angle, const float axis[3], float M[16]);   // vector_math.adb:85-111
void MatMul(const float a[16], const float b[16], float out[16]);     // generic_vector_math.adb:233-256

struct Scene {                      // scene.ads:61-73 flattened
  std::vector<ArtSphere> spheres;
  std::vector<ArtLight> lights;
  std::vector<ArtMaterial> materials;
  Mesh mymesh;
  ArtMesh mesh_desc;
  ArtSceneDesc desc;
  // Scene.Init (scene.adb:24-27): builds the internal Cornell scene; a_path is where data/pyramid2.vsgf lives
  bool Init(const std::string& a_vsgf_path, std::string& err);
};

enum Render_Type { RT_DEBUG = 0, RT_WHITTED = 1, PT_STUPID = 2, PT_SHADOW = 3, PT_MIS = 4 };   // ray_tracer.ads:40

struct Ray_Tracer {                 // package Ray_Tracer, ray_tracer.ads:18-48
  int width = 1024, height = 768;   // :20-21
  int Threads_Num = 14 * 2;         // :23
  bool Anti_Aliasing_On = true;     // :24
  int Max_Trace_Depth = 8;          // :25
  float Background_Color[3] = {0.0f, 0.0f, 0.0f};   // :27
  uint64_t seed = 1;
  std::vector<uint32_t> screen_buffer;   // ScreenBufferData(x, y): element (x,y) at x*height + y   (:35-37)
  std::vector<float> g_accBuff;          // AccumBuff(x, y) of float3                               (:54-55)

  bool Init_Render(Render_Type a_rendType);          // ray_tracer.adb:197-200
  bool Resize_Viewport(int size_x, int size_y);      // ray_tracer.adb:297-320
  bool Render_Pass();                                // ray_tracer.adb:240-293 -> art_render_pass / art_debug_hit_pass
  int GetSPP() const { return g_spp; }               // ray_tracer.adb:322-325
  bool Finished() const { return g_finish; }         // ray_tracer.adb:202-205
  std::string last_error;

 private:
  Render_Type g_rend_type = PT_MIS;
  int32_t g_spp = 0;
  bool g_finish = false;
};

struct Image { int width = 0, height = 0; std::vector<uint32_t> data; };   // bitmap.ads:15-19
void Bitmap_Init(Image& im, int w, int h);                                  // bitmap.adb:10-15
bool SaveBMP(const Image& im, const std::string& a_fileName);              // bitmap.adb:31-85
std::vector<uint8_t> BMPBytes(const Image& im);

}  // namespace art_host

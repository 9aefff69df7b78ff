// test_main.cpp -- replay of the reference's driver (test.adb:20-79) on the HIP backend:
//   Scene.Init -> Init_Render -> Resize_Viewport -> Bitmap.Init -> loop { Render_Pass; GetSPP; copy frame; SaveBMP }.
// usage: art_test <pyramid2.vsgf> <out.bmp> [width height passes threads_num render_type aa [hydra scene folder]]
// With a scene folder (the reference's SCN = "external_cpp" build, art.gpr:6-14: Scene.Init reads <folder>/statex_00001.xml,
// scene_hydra_embree.adb:303-390) the library's meshes and instances are rendered inside the internal scene (hydra_scene.hpp Build_Render_Desc).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "art_host.hpp"
#include "hydra_scene.hpp"

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <pyramid2.vsgf> <out.bmp> [w h passes threads type aa]\n", argv[0]); return 2; }
  const int w = argc > 3 ? atoi(argv[3]) : 1024, h = argc > 4 ? atoi(argv[4]) : 768;   // ray_tracer.ads:20-21
  const int passes = argc > 5 ? atoi(argv[5]) : 1;
  art_host::Scene g_scn; std::string err;
  if (art_init(-1)) { std::fprintf(stderr, "art_init: %s\n", art_last_error()); return 1; }
  if (!g_scn.Init(argv[1], err)) { std::fprintf(stderr, "Scene.Init: %s\n", err.c_str()); return 1; }      // test.adb:32
  art_host::Hydra_Scene hydra;
  const ArtSceneDesc* desc = &g_scn.desc;
  if (argc > 9) {
    if (!hydra.Load(argv[9], err) || !hydra.Build_Render_Desc(g_scn, err)) { std::fprintf(stderr, "Scene.Init (%s): %s\n", argv[9], err.c_str()); return 1; }
    desc = &hydra.r_desc;
  }
  if (art_upload_scene(desc)) { std::fprintf(stderr, "art_upload_scene: %s\n", art_last_error()); return 1; }
  art_host::Ray_Tracer rt;
  if (argc > 6) rt.Threads_Num = atoi(argv[6]);
  rt.Init_Render(argc > 7 ? (art_host::Render_Type)atoi(argv[7]) : art_host::PT_MIS);                        // test.adb:35
  if (argc > 8) rt.Anti_Aliasing_On = atoi(argv[8]) != 0;
  if (!rt.Resize_Viewport(w, h)) { std::fprintf(stderr, "Resize_Viewport: %s\n", rt.last_error.c_str()); return 1; }   // test.adb:36
  art_host::Image image; art_host::Bitmap_Init(image, w, h);                                                 // test.adb:38
  std::printf("render start\nthreads_num = %d\n", rt.Threads_Num);
  const auto t1 = std::chrono::steady_clock::now();
  int counter = 0;
  while (!rt.Finished() && counter < passes) {                                                               // test.adb:48
    if (!rt.Render_Pass()) { std::fprintf(stderr, "Render_Pass: %s\n", rt.last_error.c_str()); return 1; }
    const int spp = rt.GetSPP();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
    std::printf("pass %d - %.3fs elasped. spp = %d\n", counter, sec, spp);
    for (int y = 0; y < h; ++y)                                                                              // test.adb:63-67
      for (int x = 0; x < w; ++x) image.data[(size_t)y * w + x] = rt.screen_buffer[(size_t)x * h + y];
    if (!art_host::SaveBMP(image, argv[2])) { std::fprintf(stderr, "SaveBMP failed\n"); return 1; }          // test.adb:69
    ++counter;
  }
  std::printf("render finished\n");
  art_shutdown();
  return 0;
}

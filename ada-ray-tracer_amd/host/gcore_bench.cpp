// gcore_bench.cpp -- the reference's call pattern on the legacy geometry-core seam, natively: Threads_Num = 28 threads
// (ray_tracer.ads:23), each calling gcore_closest_hit for one ray at a time (scene_hydra_embree.adb:426-446), against the same rays as
// ONE gcore_closest_hit_n batch on the GPU.  Prints one JSON line: queries per second of both, and whether every HitCpp is the same
// bytes (the host walk and the GPU kernels test the same boxes and triangles with the same arithmetic).
//   usage: gcore_bench [triangles = 200000] [rays = 400000] [threads = 28] [instances = 1]
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "art_hip.h"

static uint64_t sm_state = 0xADA5EED0ull + 99;
static double u01() { uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; return (double)(z >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char** argv) {
  const int n_tris = argc > 1 ? std::atoi(argv[1]) : 200000, n_rays = argc > 2 ? std::atoi(argv[2]) : 400000, n_thr = argc > 3 ? std::atoi(argv[3]) : 28;
  const int n_inst = argc > 4 ? std::atoi(argv[4]) : 1;
  // a soup of small triangles in [-2, 2]^3 (the generator of the synthetic bench scenes, SURVEY 8d, in miniature)
  std::vector<float> v((size_t)n_tris * 9); std::vector<int> idx((size_t)n_tris * 3);
  const double s = 2.5 * std::pow((double)n_tris, -1.0 / 3.0);
  for (int t = 0; t < n_tris; ++t) {
    const double c[3] = {u01() * 4 - 2, u01() * 4 - 2, u01() * 4 - 2};
    for (int k = 0; k < 3; ++k) for (int a = 0; a < 3; ++a) v[(size_t)t * 9 + 3 * k + a] = (float)(c[a] + (k ? (u01() * 2 - 1) * s : 0.0));
    idx[(size_t)t * 3] = 3 * t; idx[(size_t)t * 3 + 1] = 3 * t + 1; idx[(size_t)t * 3 + 2] = 3 * t + 2;
  }
  gcore_init_and_clear();
  const int mid = gcore_add_mesh_3f(v.data(), 3 * n_tris, idx.data(), 3 * n_tris);
  std::vector<float> m((size_t)16 * n_inst, 0.0f);
  for (int i = 0; i < n_inst; ++i) { float* q = &m[(size_t)16 * i]; q[0] = q[5] = q[10] = q[15] = 1.0f; q[3] = 6.0f * (float)(i % 8); q[7] = 6.0f * (float)((i / 8) % 8); q[11] = 6.0f * (float)(i / 64); }
  gcore_instance_meshes(mid, m.data(), n_inst);
  gcore_commit_scene();
  std::vector<float> o((size_t)n_rays * 3), d((size_t)n_rays * 3);
  for (int i = 0; i < n_rays; ++i) {
    // rays across the whole grid of instances (8 x 8 x ... copies, 6 apart), aimed at the inside of one of them
    const double ex = n_inst > 1 ? 6.0 * 7 : 0.0, ey = n_inst > 8 ? 6.0 * 7 : 0.0, ez = n_inst > 64 ? 6.0 * ((n_inst - 1) / 64) : 0.0;
    const int ti = (int)(u01() * n_inst);
    const double cx = 6.0 * (ti % 8), cy = 6.0 * ((ti / 8) % 8), cz = 6.0 * (ti / 64);
    double p[3] = {u01() * (6 + ex) - 3, u01() * (6 + ey) - 3, u01() * (6 + ez) - 3}, q[3] = {cx + u01() * 3 - 1.5, cy + u01() * 3 - 1.5, cz + u01() * 3 - 1.5}, l = 0;
    for (int a = 0; a < 3; ++a) l += (q[a] - p[a]) * (q[a] - p[a]);
    l = std::sqrt(l);
    for (int a = 0; a < 3; ++a) { o[(size_t)i * 3 + a] = (float)p[a]; d[(size_t)i * 3 + a] = (float)((q[a] - p[a]) / l); }
  }
  std::vector<HitCpp> hb((size_t)n_rays), hs((size_t)n_rays); std::vector<unsigned char> fb((size_t)n_rays, 0), fs((size_t)n_rays, 0);
  std::memset(hb.data(), 0, hb.size() * sizeof(HitCpp)); std::memset(hs.data(), 0, hs.size() * sizeof(HitCpp));
  auto t0 = std::chrono::steady_clock::now();
  const int nb = gcore_closest_hit_n(n_rays, o.data(), d.data(), nullptr, nullptr, hb.data(), fb.data());
  const double t_batch = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  auto run_threads = [&](int threads) {
    std::atomic<int> next(0);
    auto t1 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < threads; ++k) th.emplace_back([&] {
      for (;;) {
        const int i0 = next.fetch_add(256);                         // a task's rays, one call each
        if (i0 >= n_rays) break;
        for (int i = i0; i < n_rays && i < i0 + 256; ++i) fs[(size_t)i] = gcore_closest_hit(&o[(size_t)i * 3], &d[(size_t)i * 3], 0.0f, 100000.0f, &hs[(size_t)i]) ? 1 : 0;
      }
    });
    for (auto& t : th) t.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
  };
  const double t_one = run_threads(1) ;
  const double t_thr = run_threads(n_thr);
  long long diff = 0, hits = 0;
  for (int i = 0; i < n_rays; ++i) {
    hits += fs[(size_t)i];
    if (fs[(size_t)i] != fb[(size_t)i] || (fs[(size_t)i] && std::memcmp(&hs[(size_t)i], &hb[(size_t)i], sizeof(HitCpp)) != 0)) ++diff;
  }
  std::printf("{\"triangles\": %d, \"instances\": %d, \"rays\": %d, \"threads\": %d, \"hardware_threads\": %u, \"hits\": %lld, \"batch_hits\": %d, \"different_from_gpu_batch\": %lld, "
              "\"single_thread_queries_per_s\": %.0f, \"threads_queries_per_s\": %.0f, \"gpu_batch_queries_per_s\": %.0f}\n",
              n_tris, n_inst, n_rays, n_thr, std::thread::hardware_concurrency(), hits, nb, diff, n_rays / t_one, n_rays / t_thr, n_rays / t_batch);
  gcore_destroy();
  art_shutdown();
  return diff == 0 ? 0 : 1;
}

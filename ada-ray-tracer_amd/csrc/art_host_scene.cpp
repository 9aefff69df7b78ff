// art_host_scene.cpp -- see art_host_scene.h.  Mirrors what Scene.Init hands to the renderer
// (scene.adb:89-217 for the internal scene, scene_hydra_embree.adb:251-296 for uploaded meshes).
#include "art_host_scene.h"

#include <chrono>
#include <cmath>
#include <cstring>

namespace art {

static bool finite3(const float* p) { return std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]); }

bool flatten_scene(const ArtSceneDesc& d, const BvhBuildParams& bp, HostScene& out, std::string& err) {
  out = HostScene();
  if (d.n_spheres < 0 || d.n_lights < 1 || d.n_materials < 1 || d.n_meshes < 0) { err = "scene: negative count, or no light / no material (Light_At(0) must exist, scene.adb:45-48)"; return false; }
  if (d.n_materials >= (1 << 22)) { err = "scene: more than 4 M materials (the shade stage packs a material index into 23 bits of its classification hint)"; return false; }
  if ((d.n_spheres && !d.spheres) || !d.lights || !d.materials || (d.n_meshes && !d.meshes)) { err = "scene: null array pointer"; return false; }
  if (d.n_spheres >= (1 << 28) || d.n_lights > 64) { err = "scene: too many spheres / lights (max 64 lights)"; return false; }
  auto mat_ok = [&](int32_t m) { return m >= 0 && m < d.n_materials; };

  for (int i = 0; i < d.n_materials; ++i) {
    const ArtMaterial& m = d.materials[i];
    if (m.type < ART_MAT_NULL || m.type > ART_MAT_PHONG) { err = "scene: unknown material type"; return false; }
    if (m.type == ART_MAT_LIGHT && (m.light < 0 || m.light >= d.n_lights)) { err = "scene: MaterialLight refers to a missing light"; return false; }
    DevMaterial dm; dm.type = m.type; dm.light = m.light; std::memcpy(dm.p, m.p, sizeof dm.p);
    out.materials.push_back(dm);
  }
  for (int i = 0; i < d.n_lights; ++i) {
    const ArtLight& l = d.lights[i];
    if (l.shape != ART_LIGHT_RECT && l.shape != ART_LIGHT_SPHERE) { err = "scene: unknown light shape"; return false; }
    if (!mat_ok(l.mat)) { err = "scene: light.mat out of range"; return false; }
    DevLight dl; std::memset(&dl, 0, sizeof dl);
    dl.shape = l.shape; dl.mat = l.mat;
    std::memcpy(dl.boxMin, l.boxMin, 12); std::memcpy(dl.boxMax, l.boxMax, 12); std::memcpy(dl.normal, l.normal, 12);
    std::memcpy(dl.center, l.center, 12); dl.radius = l.radius; std::memcpy(dl.intensity, l.intensity, 12); dl.surfaceArea = l.surfaceArea;
    out.lights.push_back(dl);
  }
  for (int i = 0; i < d.n_spheres; ++i) {
    const ArtSphere& s = d.spheres[i];
    if (!mat_ok(s.mat)) { err = "scene: sphere.mat out of range"; return false; }
    if (!finite3(s.pos) || !std::isfinite(s.r)) { err = "scene: non-finite sphere"; return false; }
    DevSphere ds; ds.x = s.pos[0]; ds.y = s.pos[1]; ds.z = s.pos[2]; ds.r = s.r;
    out.spheres.push_back(ds); out.sphere_mat.push_back(s.mat);
  }
  DevScene& h = out.hdr;
  std::memset(&h, 0, sizeof h);
  h.n_spheres = d.n_spheres; h.n_lights = d.n_lights; h.n_materials = d.n_materials;
  h.has_cornell = d.has_cornell ? 1 : 0;
  h.node_width = bp.width;
  std::memcpy(h.cb_min, d.cb_min, 12); std::memcpy(h.cb_max, d.cb_max, 12);
  std::memcpy(h.cb_mat, d.cb_mat, sizeof h.cb_mat); std::memcpy(h.cb_nrm, d.cb_nrm, sizeof h.cb_nrm);
  if (h.has_cornell) for (int i = 0; i < 6; ++i) if (!mat_ok(h.cb_mat[i])) { err = "scene: Cornell box material index out of range"; return false; }
  std::memcpy(h.cam_pos, d.cam_pos, 12); std::memcpy(h.cam_matrix, d.cam_matrix, 64);

  if (d.n_instances < 0 || (d.n_instances > 0 && !d.instances)) { err = "scene: bad instance list"; return false; }
  if (d.n_instances > 0) {
    // ---- instanced scene: meshes are object-space prototypes, the geometry is the instance list (embree_connect.cpp:147-184)
    if (bp.width != 4) { err = "scene: instanced scenes need bvh_width 4"; return false; }
    std::vector<InstMeshIn> meshes((size_t)d.n_meshes); std::vector<InstIn> insts((size_t)d.n_instances);
    std::vector<int32_t> shade_base((size_t)d.n_meshes, 0);
    int32_t max_tris = 0;
    for (int mi = 0; mi < d.n_meshes; ++mi) {
      const ArtMesh& m = d.meshes[mi];
      if (m.mode != ART_MESH_CLOSEST) { err = "scene: an instanced scene takes ART_MESH_CLOSEST meshes only"; return false; }
      if (m.nverts <= 0 || m.ntris <= 0 || !m.pos || !m.nrm || !m.idx || !m.matid) { err = "scene: empty mesh or null mesh array"; return false; }
      for (int64_t i = 0; i < 3 * (int64_t)m.ntris; ++i)
        if (m.idx[i] < 0 || m.idx[i] >= m.nverts) { err = "scene: triangle index out of range"; return false; }
      for (int64_t i = 0; i < (int64_t)m.nverts; ++i)
        if (!finite3(m.pos + 3 * i) || !finite3(m.nrm + 3 * i)) { err = "scene: non-finite mesh vertex"; return false; }
      for (int i = 0; i < m.ntris; ++i) if (!mat_ok(m.matid[i])) { err = "scene: triangle material id out of range"; return false; }
      meshes[(size_t)mi] = InstMeshIn{m.pos, (size_t)m.nverts, m.idx, (size_t)m.ntris};
      shade_base[(size_t)mi] = (int32_t)(out.m_shade.size() / kTriShadeFloats);
      const size_t b0 = out.m_shade.size();
      out.m_shade.resize(b0 + (size_t)kTriShadeFloats * (size_t)m.ntris, 0.0f);       // object-space vertex normals + material id, per triangle of the mesh
      for (int i = 0; i < m.ntris; ++i) {
        float* r = &out.m_shade[b0 + (size_t)kTriShadeFloats * (size_t)i];
        for (int k = 0; k < 3; ++k) std::memcpy(r + 3 * k, m.nrm + 3 * (size_t)m.idx[3 * (size_t)i + k], 12);
        std::memcpy(r + 9, &m.matid[i], 4);
      }
      max_tris = std::max(max_tris, m.ntris);
    }
    for (int i = 0; i < d.n_instances; ++i) {
      const ArtInstance& in = d.instances[i];
      if (in.mesh < 0 || in.mesh >= d.n_meshes) { err = "scene: instance of a missing mesh"; return false; }
      for (int k = 0; k < 12; ++k) if (!std::isfinite(in.m[k])) { err = "scene: non-finite instance matrix"; return false; }
      insts[(size_t)i].mesh = in.mesh; std::memcpy(insts[(size_t)i].m, in.m, 48);
    }
    // boxes of the meshes' trees padded for the object-space walk (art_instanced.h instanced_render_closest)
    // (scene_extent: what else a ray of this scene can start at -- the walls, the spheres and lights, the camera; the meshes' absolute pad
    // follows it and the instances' inverse matrices, art_instanced_build.cpp)
    float extent = 0.0f;
    auto reach = [&](float v) { if (std::isfinite(v)) extent = std::max(extent, std::fabs(v)); };
    for (int k = 0; k < 3; ++k) { reach(d.cam_pos[k]); if (d.has_cornell) { reach(d.cb_min[k]); reach(d.cb_max[k]); } }
    for (int i = 0; i < d.n_spheres; ++i) for (int k = 0; k < 3; ++k) reach(std::fabs(d.spheres[i].pos[k]) + std::fabs(d.spheres[i].r));
    for (int i = 0; i < d.n_lights; ++i) for (int k = 0; k < 3; ++k) { reach(d.lights[i].boxMin[k]); reach(d.lights[i].boxMax[k]); reach(std::fabs(d.lights[i].center[k]) + std::fabs(d.lights[i].radius)); }
    if (!build_two_level_host(meshes, insts, out.two, err, false, 1.0e-4f, 1.0e-5f, bp.inst_open, extent)) return false;
    int shift = 0; while ((1 << shift) < max_tris) ++shift;
    if (shift > 27 || ((uint64_t)d.n_instances << shift) > (1ull << 28)) { err = "scene: instances x triangles per mesh exceed the 28-bit hit index"; return false; }
    int64_t total = 0;
    out.inst.assign(out.two.entry.size(), DevInstance{});         // one record per entry point: the instances first (art_scene.h DevInstance)
    for (size_t e = 0; e < out.two.entry.size(); ++e) {
      const TwoLevelHost::EntryPoint& E = out.two.entry[e];
      const InstRec& R = out.two.inst[(size_t)E.inst];            // (one-sided builds keep every instance: index = the caller's)
      DevInstance& D = out.inst[e];
      std::memcpy(D.m, d.instances[E.inst].m, 48); std::memcpy(D.minv, R.minv, 48);
      D.node_base = R.node_base; D.tri_base = R.tri_base; D.shade_base = shade_base[(size_t)R.mesh];
      D.root_entry = E.root_entry; D.qroot = E.qroot; D.inst = E.inst;
    }
    for (int i = 0; i < d.n_instances; ++i) total += out.two.inst[(size_t)i].n_tris;
    if (total >= (1ll << 31)) { err = "scene: too many instanced triangles"; return false; }
    h.n_inst = d.n_instances; h.n_entry = (int32_t)out.two.entry.size(); h.inst_shift = shift; h.n_tris = (int32_t)total; h.n_nodes = out.two.tlas.n_nodes; h.node_width = 4;
    out.bvh.width = 4; out.bvh.max_stack = std::max(out.two.tlas.max_stack + 3 + out.two.blas_max_stack, 8);      // instance tree + the "leave" marker over two entries of saved world-space state + a mesh's tree, on one stack
    out.bvh.n_nodes = out.two.tlas.n_nodes + (int32_t)(out.two.blas_nodes.size() / node_floats(4)); out.bvh.n_tris = (int32_t)(out.two.blas_tris.size() / kTriFloats);      // (art_export_bvh's info: the two-level tree's sizes -- its nodes and the meshes' triangle records)
    return true;
  }
  bool have_bf = false, have_closest = false;
  for (int mi = 0; mi < d.n_meshes; ++mi) {
    const ArtMesh& m = d.meshes[mi];
    if (m.nverts <= 0 || m.ntris <= 0 || !m.pos || !m.nrm || !m.idx) { err = "scene: empty mesh or null mesh array"; return false; }
    if (m.ntris >= (1 << 27)) { err = "scene: mesh too large (max 2^27-1 triangles)"; return false; }
    for (int64_t i = 0; i < 3 * (int64_t)m.ntris; ++i)
      if (m.idx[i] < 0 || m.idx[i] >= m.nverts) { err = "scene: triangle index out of range"; return false; }
    for (int64_t i = 0; i < (int64_t)m.nverts; ++i)
      if (!finite3(m.pos + 3 * i) || !finite3(m.nrm + 3 * i)) { err = "scene: non-finite mesh vertex"; return false; }
    if (m.mode == ART_MESH_REFERENCE_BF) {
      if (have_bf) { err = "scene: at most one REFERENCE_BF mesh"; return false; }
      have_bf = true;
      if (!mat_ok(2)) { err = "scene: REFERENCE_BF mesh needs material 2 (geometry.adb:311)"; return false; }
      out.bf_pos.assign(m.pos, m.pos + 3 * (size_t)m.nverts);
      out.bf_nrm.assign(m.nrm, m.nrm + 3 * (size_t)m.nverts);
      if (m.uv) out.bf_uv.assign(m.uv, m.uv + 2 * (size_t)m.nverts); else out.bf_uv.assign(2 * (size_t)m.nverts, 0.0f);
      out.bf_idx.assign(m.idx, m.idx + 3 * (size_t)m.ntris);
      out.bf_ntris = m.ntris; h.bf_ntris = m.ntris;
      std::memcpy(h.bf_bbmin, m.bbmin, 12); std::memcpy(h.bf_bbmax, m.bbmax, 12);
    } else if (m.mode == ART_MESH_CLOSEST) {
      if (have_closest) { err = "scene: at most one CLOSEST mesh"; return false; }
      have_closest = true;
      if (!m.matid) { err = "scene: CLOSEST mesh needs material_ids"; return false; }
      for (int i = 0; i < m.ntris; ++i) if (!mat_ok(m.matid[i])) { err = "scene: triangle material id out of range"; return false; }
      out.m_pos.assign(m.pos, m.pos + 3 * (size_t)m.nverts);
      out.m_shade.assign((size_t)kTriShadeFloats * (size_t)m.ntris, 0.0f);        // texcoords are zeroed by the reference's loader and never read
      for (int i = 0; i < m.ntris; ++i) {
        float* r = &out.m_shade[(size_t)kTriShadeFloats * (size_t)i];
        for (int k = 0; k < 3; ++k) std::memcpy(r + 3 * k, m.nrm + 3 * (size_t)m.idx[3 * (size_t)i + k], 12);
        std::memcpy(r + 9, &m.matid[i], 4);
      }
      std::vector<float> tri9(9 * (size_t)m.ntris);
      for (int i = 0; i < m.ntris; ++i)
        for (int k = 0; k < 3; ++k) std::memcpy(&tri9[9 * (size_t)i + 3 * k], m.pos + 3 * (size_t)m.idx[3 * (size_t)i + k], 12);
      h.node_width = bp.width;
      const bool host_only = (bp.builder == 3 && bp.spatial_alpha >= 0.0f);        // reference splitting exists in the host builder only
      if (bp.builder >= 1 && m.ntris >= 2 && !host_only) { out.deferred_tri9 = std::move(tri9); h.n_tris = m.ntris; continue; }   // built on the GPU by the caller
      const auto t0 = std::chrono::steady_clock::now();
      if (!build_bvh8(tri9.data(), nullptr, m.ntris, bp, out.bvh, err)) return false;
      out.bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      h.n_tris = out.bvh.n_tris; h.n_nodes = out.bvh.n_nodes;
    } else { err = "scene: unknown mesh mode"; return false; }
  }
  return true;
}

bool flatten_instances(const ArtSceneDesc& d, std::vector<float>& pos, std::vector<float>& nrm, std::vector<int32_t>& idx, std::vector<int32_t>& matid, std::string& err) {
  pos.clear(); nrm.clear(); idx.clear(); matid.clear();
  std::vector<InstMeshIn> meshes; std::vector<InstIn> insts((size_t)d.n_instances);
  for (int mi = 0; mi < d.n_meshes; ++mi) meshes.push_back(InstMeshIn{d.meshes[mi].pos, (size_t)d.meshes[mi].nverts, d.meshes[mi].idx, (size_t)d.meshes[mi].ntris});
  for (int i = 0; i < d.n_instances; ++i) { insts[(size_t)i].mesh = d.instances[i].mesh; std::memcpy(insts[(size_t)i].m, d.instances[i].m, 48); }
  TwoLevelHost two;
  if (!build_two_level_host(meshes, insts, two, err, false, 1.0e-4f, 1.0e-5f)) return false;       // (for the inverse matrices: the instanced upload's own)
  for (int i = 0; i < d.n_instances; ++i) {
    const ArtMesh& m = d.meshes[d.instances[i].mesh];
    const float* M = d.instances[i].m; const float* Minv = two.inst[(size_t)i].minv;
    const int32_t v0 = (int32_t)(pos.size() / 3);
    for (int v = 0; v < m.nverts; ++v) {
      const f3 p = xform_point(M, mk3(m.pos[3 * v], m.pos[3 * v + 1], m.pos[3 * v + 2]));
      const f3 n = instance_normal(Minv, mk3(m.nrm[3 * v], m.nrm[3 * v + 1], m.nrm[3 * v + 2]));
      pos.insert(pos.end(), {p.x, p.y, p.z}); nrm.insert(nrm.end(), {n.x, n.y, n.z});
    }
    for (int t = 0; t < m.ntris; ++t) { for (int k = 0; k < 3; ++k) idx.push_back(v0 + m.idx[3 * t + k]); matid.push_back(m.matid[t]); }
  }
  return true;
}

std::vector<uint32_t> build_pixmap(int W, int H, int rank, int nranks, int T) {
  std::vector<uint32_t> pm;
  pm.reserve((size_t)W * H / (size_t)(nranks > 0 ? nranks : 1) + 1024);
  const int tx = (W + T - 1) / T, ty = (H + T - 1) / T;
  // The deal: tile (bx, by) belongs to rank (bx + s by) mod n, s odd and coprime to n -- diagonals, whatever the frame width.  Rounds 1-3
  // dealt tile_id mod n: with a frame of 128 tiles per row (4096 pixels) and n = 8 that is bx mod 8, i.e. every rank owns full-height
  // COLUMNS of tiles, and a bright object a column or two wide lands on one rank: 2.7 % (max over mean) on C5 in the per-rank rehearsal
  // of round 4 (profiles/r4_scaling.json), 1.6 % on C4 whose 60 tiles per row shift the columns by 4 from row to row.
  const int n = nranks > 0 ? nranks : 1;
  const int skew = (n % 3 != 0) ? 3 : (n % 5 != 0) ? 5 : 7;
  for (int by = 0; by < ty; ++by)
    for (int bx = 0; bx < tx; ++bx) {
      if ((bx + skew * by) % n != rank) continue;
      for (int y = by * T; y < (H < (by + 1) * T ? H : (by + 1) * T); ++y)
        for (int x = bx * T; x < (W < (bx + 1) * T ? W : (bx + 1) * T); ++x) pm.push_back((uint32_t)(y * W + x));
    }
  return pm;
}

void bind_host_pointers(HostScene& hs) {
  DevScene& h = hs.hdr;
  h.spheres = hs.spheres.data(); h.sphere_mat = hs.sphere_mat.data();
  h.lights = hs.lights.data(); h.materials = hs.materials.data();
  h.bf_pos = hs.bf_pos.data(); h.bf_nrm = hs.bf_nrm.data(); h.bf_uv = hs.bf_uv.data(); h.bf_idx = hs.bf_idx.data();
  h.nodes = hs.bvh.nodes.data(); h.tris = hs.bvh.tris.data();
  h.m_shade = hs.m_shade.data();
  if (h.n_inst > 0) {
    h.inst = hs.inst.data(); h.tlas_nodes = hs.two.tlas.nodes.data(); h.tlas_tris = hs.two.tlas.tris.data();
    h.blas_nodes = hs.two.blas_nodes.data(); h.blas_tris = hs.two.blas_tris.data();
  }
}

}  // namespace art

// hydra_scene.cpp -- see hydra_scene.hpp.
#include "hydra_scene.hpp"
#include <algorithm>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace art_host {

const XmlNode* XmlNode::child(const std::string& n) const {
  for (const XmlNode& c : children) if (c.name == n) return &c;
  return nullptr;
}
std::string XmlNode::attribute(const std::string& n) const {
  for (const auto& a : attrs) if (a.first == n) return a.second;
  return std::string();
}

namespace {

struct Parser {
  const std::string& s; size_t i = 0; std::string err;
  explicit Parser(const std::string& t) : s(t) {}
  bool eof() const { return i >= s.size(); }
  void skip_ws() { while (!eof() && std::isspace((unsigned char)s[i])) ++i; }
  bool starts(const char* lit) const { return s.compare(i, std::strlen(lit), lit) == 0; }
  bool skip_until(const char* lit) { const size_t p = s.find(lit, i); if (p == std::string::npos) { err = std::string("unterminated construct, expected ") + lit; return false; } i = p + std::strlen(lit); return true; }
  static std::string unescape(const std::string& v) {
    std::string o; o.reserve(v.size());
    for (size_t k = 0; k < v.size(); ++k) {
      if (v[k] != '&') { o += v[k]; continue; }
      static const struct { const char* e; char c; } tab[] = {{"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'}, {"&quot;", '"'}, {"&apos;", '\''}};
      bool done = false;
      for (const auto& t : tab) if (v.compare(k, std::strlen(t.e), t.e) == 0) { o += t.c; k += std::strlen(t.e) - 1; done = true; break; }
      if (!done) o += v[k];
    }
    return o;
  }
  std::string name() { const size_t b = i; while (!eof() && (std::isalnum((unsigned char)s[i]) || s[i] == '_' || s[i] == '-' || s[i] == ':' || s[i] == '.')) ++i; return s.substr(b, i - b); }
  // parses the content of `parent` up to its closing tag (or the end of the document when parent is the document node)
  bool content(XmlNode& parent, bool is_doc) {
    for (;;) {
      const size_t lt = s.find('<', i);
      if (lt == std::string::npos) {
        if (!is_doc) { err = "missing </" + parent.name + ">"; return false; }
        i = s.size(); return true;
      }
      parent.text += unescape(s.substr(i, lt - i));
      i = lt;
      if (starts("<!--")) { if (!skip_until("-->")) return false; continue; }
      if (starts("<?")) { if (!skip_until("?>")) return false; continue; }
      if (starts("<![CDATA[")) { const size_t b = i + 9; if (!skip_until("]]>")) return false; parent.text += s.substr(b, i - 3 - b); continue; }
      if (starts("<!")) { if (!skip_until(">")) return false; continue; }
      if (starts("</")) {
        i += 2; const std::string n = name(); skip_ws();
        if (eof() || s[i] != '>') { err = "malformed closing tag </" + n; return false; }
        ++i;
        if (is_doc || n != parent.name) { err = "unexpected </" + n + ">"; return false; }
        return true;
      }
      ++i;
      XmlNode el; el.name = name();
      if (el.name.empty()) { err = "malformed tag"; return false; }
      for (;;) {
        skip_ws();
        if (eof()) { err = "unterminated tag <" + el.name; return false; }
        if (s[i] == '>') { ++i; if (!content(el, false)) return false; break; }
        if (starts("/>")) { i += 2; break; }
        const std::string an = name();
        skip_ws();
        if (an.empty() || eof() || s[i] != '=') { err = "malformed attribute in <" + el.name + ">"; return false; }
        ++i; skip_ws();
        if (eof() || (s[i] != '"' && s[i] != '\'')) { err = "attribute value of " + an + " is not quoted"; return false; }
        const char q = s[i++]; const size_t e = s.find(q, i);
        if (e == std::string::npos) { err = "unterminated attribute value of " + an; return false; }
        el.attrs.emplace_back(an, unescape(s.substr(i, e - i))); i = e + 1;
      }
      parent.children.push_back(std::move(el));
    }
  }
};

bool read_floats(const std::string& str, float* out, int n) {
  const char* p = str.c_str();
  for (int k = 0; k < n; ++k) {
    char* end = nullptr;
    out[k] = std::strtof(p, &end);                    // Float'Value of the next blank-separated token
    if (end == p) return false;
    p = end;
  }
  return true;
}

std::string trim(const std::string& v) {
  size_t b = 0, e = v.size();
  while (b < e && std::isspace((unsigned char)v[b])) ++b;
  while (e > b && std::isspace((unsigned char)v[e - 1])) --e;
  return v.substr(b, e - b);
}

}  // namespace

bool ParseXml(const std::string& text, XmlNode& root, std::string& err) {
  root = XmlNode();
  Parser p(text);
  if (!p.content(root, true)) { err = "xml: " + p.err + " at offset " + std::to_string(p.i); return false; }
  return true;
}

bool Read_Float3_From_String(const std::string& s, float out[3]) { return read_floats(s, out, 3); }
bool Read_Float16_From_String(const std::string& s, float out[16]) { return read_floats(s, out, 16); }

bool Hydra_Scene::Load(const std::string& a_path, std::string& err) {
  *this = Hydra_Scene();
  const std::string xmlFileName = a_path + "/statex_00001.xml";                       // :320
  std::ifstream f(xmlFileName, std::ios::binary);
  if (!f) { err = "cannot open " + xmlFileName; return false; }
  std::stringstream ss; ss << f.rdbuf();
  XmlNode doc;
  if (!ParseXml(ss.str(), doc, err)) return false;
  // the libraries hang below the document's root element in Hydra files; the reference asks Document.Root for them (:337-347)
  const XmlNode* root = &doc;
  if (!doc.child("geometry_lib") && doc.children.size() == 1) root = &doc.children[0];
  const XmlNode* matlib = root->child("materials_lib");
  const XmlNode* lgtlib = root->child("lights_lib");
  const XmlNode* geolib = root->child("geometry_lib");
  const XmlNode* scnlib = root->child("scenes");
  if (!matlib || !lgtlib || !geolib || !scnlib) { err = "scene library lacks materials_lib / lights_lib / geometry_lib / scenes (scene_hydra_embree.adb:349-352)"; return false; }
  num_lights = (int)lgtlib->children.size();
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  for (const XmlNode& m : geolib->children) {                                           // Load_Meshes :85-103
    Mesh mesh;
    if (!LoadMeshFromVSGF(mesh, I, a_path + "/" + m.attribute("loc"), err)) return false;
    meshes.push_back(std::move(mesh));
  }
  for (const XmlNode& m : matlib->children) {                                           // Create_Material_From_Node :192-205
    HydraMaterial hm; hm.name = m.attribute("name"); hm.diffuse[0] = hm.diffuse[1] = hm.diffuse[2] = 0.0f;
    const XmlNode* diff = m.child("diffuse");
    const XmlNode* col = diff ? diff->child("color") : nullptr;
    if (col) {                                                                           // Read_Float3_Val :172-189: attribute 'val', else the text
      const std::string v = col->attribute("val");
      if (!Read_Float3_From_String(v.empty() ? trim(col->text) : v, hm.diffuse)) { err = "material '" + hm.name + "': bad diffuse colour"; return false; }
    }
    materials.push_back(hm);
  }
  if (meshes.empty() || materials.empty()) { err = "scene library without meshes or materials (scene_hydra_embree.adb:372-373)"; return false; }
  const XmlNode* scene = scnlib->child("scene");
  if (!scene) { err = "scenes/scene missing"; return false; }
  for (const XmlNode& n : scene->children) {                                             // Instance_All_Meshes :272-296
    if (n.name != "instance") continue;
    HydraInstance in;
    in.mesh_id = std::atoi(n.attribute("mesh_id").c_str());
    if (!Read_Float16_From_String(n.attribute("matrix"), in.matrix)) { err = "instance: matrix needs 16 numbers"; return false; }
    if (in.mesh_id < 0 || in.mesh_id >= (int)meshes.size()) { err = "instance: mesh_id out of range"; return false; }
    instances.push_back(in);
  }
  return true;
}

bool Hydra_Scene::Init(const std::string& a_path, std::string& err) {
  if (!Load(a_path, err)) return false;
  gcore_init_and_clear();                                                                 // :377
  for (const Mesh& m : meshes) {                                                          // Add_Meshes_To_GCore :252-270 (element counts)
    int id = -1;
    if (!m.triangles.empty())
      id = gcore_add_mesh_3f(m.vert_positions.data(), (int)(m.vert_positions.size() / 3), m.triangles.data(), (int)m.triangles.size());
    geom_ids.push_back(id);
  }
  for (const HydraInstance& in : instances) {
    if (geom_ids[in.mesh_id] < 0) { err = "instance of an empty mesh"; return false; }
    gcore_instance_meshes(geom_ids[in.mesh_id], in.matrix, 1);                            // :290
  }
  gcore_commit_scene();                                                                   // :383
  return true;
}

bool Hydra_Scene::Build_Render_Desc(const Scene& base, std::string& err) {
  if (meshes.empty() || instances.empty() || materials.empty()) { err = "scene library without meshes, instances or materials"; return false; }
  r_materials = base.materials;
  const int32_t first = (int32_t)r_materials.size();
  for (const HydraMaterial& hm : materials) {
    ArtMaterial m; std::memset(&m, 0, sizeof m);
    m.type = ART_MAT_LAMBERT; std::memcpy(m.p, hm.diffuse, 12);
    r_materials.push_back(m);
  }
  r_meshes.assign(meshes.size(), ArtMesh()); r_matids.assign(meshes.size(), {});
  clamped_material_ids = 0;
  for (size_t i = 0; i < meshes.size(); ++i) {
    const Mesh& me = meshes[i];
    ArtMesh& d = r_meshes[i]; std::memset(&d, 0, sizeof d);
    if (me.triangles.empty()) { err = "mesh " + std::to_string(i) + " has no triangles"; return false; }
    const size_t nt = me.triangles.size() / 3;
    r_matids[i].resize(nt);
    for (size_t t = 0; t < nt; ++t) {
      const int32_t id = t < me.material_ids.size() ? me.material_ids[t] : 0;
      const int32_t kept = std::min(std::max(id, 0), (int32_t)materials.size() - 1);
      if (kept != id) clamped_material_ids += 1;
      r_matids[i][t] = first + kept;
    }
    d.mode = ART_MESH_CLOSEST;
    d.nverts = (int32_t)(me.vert_positions.size() / 3); d.ntris = (int32_t)nt;
    d.pos = me.vert_positions.data(); d.nrm = me.vert_normals.data(); d.uv = nullptr;
    d.idx = me.triangles.data(); d.matid = r_matids[i].data();
    std::memcpy(d.bbmin, me.bbox_min, 12); std::memcpy(d.bbmax, me.bbox_max, 12);
  }
  r_instances.clear();
  for (const HydraInstance& in : instances) {
    ArtInstance a; a.mesh = in.mesh_id; std::memcpy(a.m, in.matrix, 48);
    r_instances.push_back(a);
  }
  if (clamped_material_ids > 0)                                // once per build, not per triangle
    std::fprintf(stderr, "Hydra_Scene::Build_Render_Desc: %lld triangle(s) name a material outside the library's %zu; they take its nearest one\n",
                 (long long)clamped_material_ids, materials.size());
  r_desc = base.desc;                                          // box, spheres, light, camera of the internal scene -- NOT its own mesh (hydra_scene.hpp)
  r_desc.n_materials = (int32_t)r_materials.size(); r_desc.materials = r_materials.data();
  r_desc.n_meshes = (int32_t)r_meshes.size(); r_desc.meshes = r_meshes.data();
  r_desc.n_instances = (int32_t)r_instances.size(); r_desc.instances = r_instances.data();
  return true;
}

void Hydra_Scene::Destroy() { gcore_destroy(); }

bool Hydra_Scene::Find_Closest_Hit(const float origin[3], const float direction[3], HitCpp& hit) const {
  return gcore_closest_hit(origin, direction, 0.0f, 100000.0f, &hit);                     // :433-434
}

}  // namespace art_host

// C entry points for the tests
static art_host::Hydra_Scene g_hydra;
extern "C" int art_host_hydra_load(const char* dir, int* counts4 /* meshes, materials, instances, lights */, float* first_diffuse3, float* matrices /* cap 16*64 */) {
  std::string err;
  if (!g_hydra.Load(dir, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
  counts4[0] = (int)g_hydra.meshes.size(); counts4[1] = (int)g_hydra.materials.size(); counts4[2] = (int)g_hydra.instances.size(); counts4[3] = g_hydra.num_lights;
  std::memcpy(first_diffuse3, g_hydra.materials[0].diffuse, 12);
  for (size_t i = 0; i < g_hydra.instances.size() && i < 64; ++i) std::memcpy(matrices + 16 * i, g_hydra.instances[i].matrix, 64);
  return 0;
}
// The scene library as a render scene (Hydra_Scene::Build_Render_Desc) behind a handle that OWNS everything the descriptor points into
// (round 6, ADVICE r5: the first form returned pointers into function-local statics that the next call invalidated under a caller still
// holding them).  vsgf_path: where the internal scene finds data/pyramid2.vsgf (Scene.Init).  nullptr on failure, message on stderr.
namespace { struct HydraRenderScene { art_host::Scene base; art_host::Hydra_Scene hy; }; }
extern "C" void* art_host_hydra_scene_create(const char* dir, const char* vsgf_path) {
  HydraRenderScene* h = new HydraRenderScene;
  std::string err;
  if (!dir || !vsgf_path || !h->base.Init(vsgf_path, err) || !h->hy.Load(dir, err) || !h->hy.Build_Render_Desc(h->base, err)) {
    std::fprintf(stderr, "art_host_hydra_scene_create: %s\n", err.c_str());
    delete h;
    return nullptr;
  }
  return h;
}
// the descriptor art_upload_scene takes (it copies): valid until art_host_hydra_scene_destroy(handle)
extern "C" const ArtSceneDesc* art_host_hydra_scene_desc(void* handle) { return handle ? &static_cast<HydraRenderScene*>(handle)->hy.r_desc : nullptr; }
extern "C" long long art_host_hydra_scene_clamped_ids(void* handle) { return handle ? static_cast<HydraRenderScene*>(handle)->hy.clamped_material_ids : -1; }
extern "C" void art_host_hydra_scene_destroy(void* handle) { delete static_cast<HydraRenderScene*>(handle); }
extern "C" int art_host_hydra_init(const char* dir) {
  std::string err;
  if (!g_hydra.Init(dir, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
  return 0;
}
extern "C" int art_host_hydra_closest_hits(const float* origins, const float* dirs, int n, HitCpp* out, int* hit_flags) {
  for (int i = 0; i < n; ++i) hit_flags[i] = g_hydra.Find_Closest_Hit(origins + 3 * i, dirs + 3 * i, out[i]) ? 1 : 0;
  return 0;
}
extern "C" void art_host_hydra_destroy() { g_hydra.Destroy(); }

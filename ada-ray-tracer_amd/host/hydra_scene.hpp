// hydra_scene.hpp -- C++ mirror of the reference's "external_cpp" Scene body, scene_hydra_embree.adb:
// Init (:303-390) reads <folder>/statex_00001.xml (Hydra legacy scene library), loads every mesh of <geometry_lib> from its
// VSGF file (Load_Meshes :85-103), reads <materials_lib> (Load_Materials :192-225: diffuse colour only, as in the reference),
// hands the meshes to the geometry core (Add_Meshes_To_GCore :252-270) and instances them with the 16-float matrices of
// <scenes>/<scene>/<instance> (Instance_All_Meshes :272-296), then commits.  Find_Closest_Hit (:426-446) is one
// gcore_closest_hit call.  The geometry core is this repository's libart_hip.so instead of cpp/embree_connect.cpp.
//
// The reference parses XML with pugixml through an Ada binding; this mirror carries a ~100-line reader for the subset the
// scene library uses (elements, quoted attributes, text, comments, declarations).
#pragma once
#include <string>
#include <vector>
#include "art_host.hpp"

namespace art_host {

struct XmlNode {
  std::string name, text;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<XmlNode> children;
  const XmlNode* child(const std::string& n) const;             // first child called n, or nullptr  (XML_Node.child)
  std::string attribute(const std::string& n) const;            // "" when absent                     (XML_Node.attribute(..).value)
};
bool ParseXml(const std::string& text, XmlNode& root, std::string& err);     // root = synthetic document node

bool Read_Float3_From_String(const std::string& s, float out[3]);           // scene_hydra_embree.adb:106-133
bool Read_Float16_From_String(const std::string& s, float out[16]);         // :137-165

struct HydraMaterial { std::string name; float diffuse[3]; };
struct HydraInstance { int mesh_id; float matrix[16]; };

struct Hydra_Scene {                    // Render_Scene of scene_hydra_embree.adb
  std::vector<Mesh> meshes;
  std::vector<HydraMaterial> materials;
  std::vector<HydraInstance> instances;
  std::vector<int> geom_ids;            // value returned by gcore_add_mesh_3f per mesh
  int num_lights = 0;

  // The library's geometry as a scene for Ray_Tracer.Render_Pass (round 5; art_hip.h ArtSceneDesc::instances): every mesh of geometry_lib an
  // object-space prototype, every <instance> an ArtInstance (the first 12 floats of its matrix, as rtcSetGeometryTransform reads them,
  // embree_connect.cpp:169) -- rendered through the two-level tree, nothing flattened.  The reference's own "external_cpp" body stops short
  // of this (Find_Closest_Hit returns matId -1, lights are only counted, no camera is read: scene_hydra_embree.adb:303-390, :426-446), so
  // the surroundings are the internal scene's (`base`: box, spheres, light, camera of Scene.Init) and the materials are its table 0..10
  // followed by ONE Lambert per <material> (the diffuse colour is all Load_Materials reads, :192-225); a triangle takes material
  // 11 + its VSGF material id, ids outside the library its nearest one (counted in clamped_material_ids, one line on stderr per build).
  // NOT in the picture: the internal scene's own mesh (data/pyramid2.vsgf, a brute-force mesh in world space) -- an instanced render
  // scene holds closest-hit prototypes only (art_hip.h ArtSceneDesc::instances), so `base` contributes its analytic surroundings and nothing else.
  long long clamped_material_ids = 0;
  std::vector<ArtMaterial> r_materials; std::vector<ArtMesh> r_meshes; std::vector<std::vector<int32_t>> r_matids; std::vector<ArtInstance> r_instances;
  ArtSceneDesc r_desc;
  bool Build_Render_Desc(const Scene& base, std::string& err);   // after Load; the descriptor points into this object and into `base`

  bool Load(const std::string& a_path, std::string& err);      // parsing + VSGF loading only (no GPU)
  bool Init(const std::string& a_path, std::string& err);      // Load + gcore_init_and_clear / add / instance / commit
  void Destroy();                                              // gcore_destroy (:392-397)
  // Scene.Find_Closest_Hit (:426-446): tnear 0, tfar 1e5
  bool Find_Closest_Hit(const float origin[3], const float direction[3], HitCpp& hit) const;
};

}  // namespace art_host

--  scene_hip.adb -- third body for package Scene (spec scene.ads unchanged), selected in art.gpr by
--      type Scene_Type is ("internal_ada", "external_cpp", "external_hip");   SCN : Scene_Type := "external_hip";
--      when "external_hip" => SCN_File := "scene_hip.adb";   Linker switches: "-Lada-ray-tracer_amd", "-lart_hip"
--  It builds the internal Cornell scene exactly like scene.adb:89-217 (Init_Cornell_Box is textually shared),
--  then flattens the tagged Material / Light objects into the tables of Art_Hip.Art_Scene_Desc and uploads them once.
--  Find_Closest_Hit stays available for debugging through the internal Ada intersectors.
--  Syntax-reviewed only (no GNAT in the build image); the flattening rules are the ones exercised by
--  host/art_host.cpp (C++ mirror) and tests/conv.py.
with Art_Hip;      use Art_Hip;
with Interfaces.C; use Interfaces.C;
with System;

separate (Scene)
procedure Upload_To_Hip (a_scn : in out Render_Scene) is

  use type Materials.MaterialRef;

  n_mat : constant Integer := a_scn.materials'Length;
  mats  : array (0 .. n_mat - 1) of aliased Art_Material;
  sphs  : array (a_scn.spheres'Range) of aliased Art_Sphere;
  lgt   : aliased Art_Light;
  msh   : aliased Art_Mesh;
  desc  : aliased Art_Scene_Desc;

  function F (x : float) return C_float is (C_float (x));
  function F3 (v : float3) return Float3_C is ((F (v.x), F (v.y), F (v.z)));

  --  index of a material object inside a_scn.materials (the sphere-light material is equal to materials(4))
  function Index_Of (m : Materials.MaterialRef) return int is
  begin
    for i in a_scn.materials'Range loop
      if a_scn.materials (i) = m then return int (i); end if;
    end loop;
    return 4;
  end Index_Of;

begin
  for i in a_scn.materials'Range loop
    declare
      m : Art_Material := (kind => ART_MAT_NULL, light => 0, p => (others => 0.0));
      r : constant Materials.MaterialRef := a_scn.materials (i);
    begin
      if r /= null then
        if r.all in Materials.MaterialLight'Class then
          m.kind := ART_MAT_LIGHT;
        elsif r.all in Materials.MaterialLambert'Class then
          m.kind := ART_MAT_LAMBERT;
          m.p (0) := F (Materials.MaterialLambert (r.all).kd.x);
          m.p (1) := F (Materials.MaterialLambert (r.all).kd.y);
          m.p (2) := F (Materials.MaterialLambert (r.all).kd.z);
        elsif r.all in Materials.MaterialMirror'Class then
          m.kind := ART_MAT_MIRROR;
          m.p (0) := F (Materials.MaterialMirror (r.all).reflection.x);
          m.p (1) := F (Materials.MaterialMirror (r.all).reflection.y);
          m.p (2) := F (Materials.MaterialMirror (r.all).reflection.z);
        elsif r.all in Materials.MaterialFresnelDielectric'Class then
          m.kind := ART_MAT_GLASS;
          m.p (0) := F (Materials.MaterialFresnelDielectric (r.all).reflection.x);
          m.p (1) := F (Materials.MaterialFresnelDielectric (r.all).reflection.y);
          m.p (2) := F (Materials.MaterialFresnelDielectric (r.all).reflection.z);
          m.p (3) := F (Materials.MaterialFresnelDielectric (r.all).transparency.x);
          m.p (4) := F (Materials.MaterialFresnelDielectric (r.all).transparency.y);
          m.p (5) := F (Materials.MaterialFresnelDielectric (r.all).transparency.z);
          m.p (6) := F (Materials.MaterialFresnelDielectric (r.all).ior);
        elsif r.all in Materials.MaterialPhong'Class then
          m.kind := ART_MAT_PHONG;
          m.p (0) := F (Materials.MaterialPhong (r.all).reflection.x);
          m.p (1) := F (Materials.MaterialPhong (r.all).reflection.y);
          m.p (2) := F (Materials.MaterialPhong (r.all).reflection.z);
          m.p (3) := F (Materials.MaterialPhong (r.all).cosPower);
        end if;
      end if;
      mats (i) := m;
    end;
  end loop;

  for i in a_scn.spheres'Range loop
    sphs (i) := (pos => F3 (a_scn.spheres (i).pos), r => F (a_scn.spheres (i).r), mat => Index_Of (a_scn.spheres (i).mat));
  end loop;

  --  the single light of the reference (scene.adb:45-48)
  if Lights.GetShapeType (a_scn.g_lightRef) = Lights.Light_Shape_Sphere then
    declare
      s : Lights.SphereLight renames Lights.SphereLight (a_scn.g_lightRef.all);
    begin
      lgt := (shape => ART_LIGHT_SPHERE, mat => 4, boxMin | boxMax | normal => (others => 0.0),
              center => F3 (s.center), radius => F (s.radius), intensity => F3 (s.intensity), surfaceArea => F (s.surfaceArea));
    end;
  else
    declare
      a : Lights.AreaLight renames Lights.AreaLight (a_scn.g_lightRef.all);
    begin
      lgt := (shape => ART_LIGHT_RECT, mat => 4, boxMin => F3 (a.boxMin), boxMax => F3 (a.boxMax), normal => F3 (a.normal),
              center => (others => 0.0), radius => 0.0, intensity => F3 (a.intensity), surfaceArea => F (a.surfaceArea));
    end;
  end if;

  msh := (mode => ART_MESH_REFERENCE_BF,
          nverts => int (a_scn.mymesh.vert_positions'Length), ntris => int (a_scn.mymesh.triangles'Length),
          pos => a_scn.mymesh.vert_positions (0)'Address, nrm => a_scn.mymesh.vert_normals (0)'Address,
          uv => a_scn.mymesh.vert_tex_coords (0)'Address, idx => a_scn.mymesh.triangles (0)'Address,
          matid => a_scn.mymesh.material_ids (0)'Address,
          bbmin => F3 (a_scn.mymesh.bbox.min), bbmax => F3 (a_scn.mymesh.bbox.max));

  desc.n_spheres := int (sphs'Length);  desc.spheres := sphs (sphs'First)'Address;
  desc.has_cornell := 1;
  desc.cb_min := F3 (my_cornell_box.box.min);  desc.cb_max := F3 (my_cornell_box.box.max);
  for k in 0 .. 5 loop
    desc.cb_mat (k) := int (my_cornell_box.mat_indices (k));
    desc.cb_nrm (k) := F3 (my_cornell_box.normals (k));
  end loop;
  desc.n_lights := 1;  desc.lights := lgt'Address;
  desc.n_materials := int (n_mat);  desc.materials := mats (0)'Address;
  desc.n_meshes := 1;  desc.meshes := msh'Address;
  desc.n_instances := 0;  desc.instances := System.Null_Address;   --  the internal scene has no instances (the Hydra scene's <instance> nodes become Art_Instance records: INTEGRATION.md section 2)
  desc.cam_pos := F3 (a_scn.g_cam.pos);
  for i in 0 .. 3 loop
    for j in 0 .. 3 loop
      desc.cam_matrix (i * 4 + j) := F (a_scn.g_cam.matrix (i, j));   --  row-major float4x4 (generic_vector_math.ads:64)
    end loop;
  end loop;

  if art_init (-1) /= 0 or else art_upload_scene (desc'Access) /= 0 then
    Put_Line ("art_hip: " & Interfaces.C.Strings.Value (art_last_error));
  end if;
end Upload_To_Hip;

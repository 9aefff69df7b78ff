--  art_hip.ads -- thin Ada binding of libart_hip.so (include/art_hip.h), the MI355X render backend.
--  NOTE: written against the reference sources without an Ada toolchain in the build image (no GNAT):
--  syntax-reviewed only.  The same calling sequence is exercised by host/test_main.cpp (C++ mirror).
--
--  Record layouts are Convention C and match include/art_hip.h field for field.
with Interfaces;   use Interfaces;
with Interfaces.C; use Interfaces.C;
with System;
with Interfaces.C.Strings;

package Art_Hip is

  type Float3_C is array (0 .. 2) of aliased C_float;   pragma Convention (C, Float3_C);
  type Float8_C is array (0 .. 7) of aliased C_float;   pragma Convention (C, Float8_C);
  type Float16_C is array (0 .. 15) of aliased C_float; pragma Convention (C, Float16_C);
  type Int6_C is array (0 .. 5) of aliased int;         pragma Convention (C, Int6_C);
  type Normals6_C is array (0 .. 5) of Float3_C;        pragma Convention (C, Normals6_C);

  ART_MAT_NULL    : constant := 0;  ART_MAT_LIGHT  : constant := 1;  ART_MAT_LAMBERT : constant := 2;
  ART_MAT_MIRROR  : constant := 3;  ART_MAT_GLASS  : constant := 4;  ART_MAT_PHONG   : constant := 5;
  ART_LIGHT_RECT  : constant := 0;  ART_LIGHT_SPHERE : constant := 1;
  ART_MESH_REFERENCE_BF : constant := 0;  ART_MESH_CLOSEST : constant := 1;
  ART_LAYOUT_ADA_XY : constant := 0;      ART_LAYOUT_ROW_MAJOR : constant := 1;

  type Art_Material is record          --  materials.ads:58-130 flattened
    kind  : int;
    light : int;
    p     : Float8_C;
  end record;
  pragma Convention (C, Art_Material);

  type Art_Light is record             --  lights.ads:36-55
    shape       : int;
    mat         : int;
    boxMin, boxMax, normal : Float3_C;
    center      : Float3_C;
    radius      : C_float;
    intensity   : Float3_C;
    surfaceArea : C_float;
  end record;
  pragma Convention (C, Art_Light);

  type Art_Sphere is record            --  geometry.ads:21-25
    pos : Float3_C;
    r   : C_float;
    mat : int;
  end record;
  pragma Convention (C, Art_Sphere);

  type Art_Mesh is record              --  geometry.ads:94-101
    mode   : int;
    nverts : int;
    ntris  : int;
    pos, nrm, uv : System.Address;     --  Float3_Array / Float2_Array storage ('Address of element 0)
    idx, matid   : System.Address;     --  Triangle_Array (3 ints each) / MaterialsId_Array
    bbmin, bbmax : Float3_C;
  end record;
  pragma Convention (C, Art_Mesh);

  type Float12_C is array (0 .. 11) of aliased C_float;
  pragma Convention (C, Float12_C);
  type Art_Instance is record          --  embree_connect.cpp:147-184: one instance of a mesh, object -> world 3x4 row-major
    mesh : int;
    m    : Float12_C;
  end record;
  pragma Convention (C, Art_Instance);

  type Art_Scene_Desc is record
    n_spheres   : int;  spheres   : System.Address;
    has_cornell : int;
    cb_min, cb_max : Float3_C;
    cb_mat      : Int6_C;
    cb_nrm      : Normals6_C;
    n_lights    : int;  lights    : System.Address;
    n_materials : int;  materials : System.Address;
    n_meshes    : int;  meshes    : System.Address;
    cam_pos     : Float3_C;
    cam_matrix  : Float16_C;
    n_instances : int;  instances : System.Address;   --  > 0: meshes are object-space prototypes, the geometry is the instance list
  end record;
  pragma Convention (C, Art_Scene_Desc);

  type Art_Pass_Params is record       --  the package variables Render_Pass reads, ray_tracer.ads:20-32
    render_type : int;                 --  Render_Type'Pos (g_rend_type)
    aa_on       : int;
    max_depth   : int;
    vthreads    : int;                 --  Threads_Num
    background  : Float3_C;
    seed        : Unsigned_64;
    layout      : int;
  end record;
  pragma Convention (C, Art_Pass_Params);

  type Art_Hit is record               --  geometry.ads:57-67 flattened (include/art_hip.h ArtHit, 44 bytes)
    t          : C_float;
    is_hit     : int;
    prim_type  : int;                  --  Primitive'Pos: 0 plane, 1 sphere, 2 triangle, 3 quad; -1 miss
    prim_index : int;
    mat_id     : int;
    mat        : int;
    normal     : Float3_C;
    u, v       : C_float;              --  triangle barycentrics (weight of C, weight of B; geometry.adb:245-246)
  end record;
  pragma Convention (C, Art_Hit);

  ART_TRACE_COOP : constant := 0;  ART_TRACE_SIMPLE : constant := 1;
  --  option "bvh_builder": 3 binned SAH on the GPU (default), 0 binned SAH on the host, 1 LBVH, 2 PLOC on the GPU
  ART_BVH_GPU_SAH : constant := 3;  ART_BVH_HOST_SAH : constant := 0;  ART_BVH_GPU_LBVH : constant := 1;  ART_BVH_GPU_PLOC : constant := 2;

  function art_init (device_ordinal : int) return int;
  pragma Import (C, art_init, "art_init");

  --  one process, n GPUs of the node (instead of art_init): ordinals = Null_Address means devices 0 .. n-1
  function art_init_devices (n : int; ordinals : System.Address) return int;
  pragma Import (C, art_init_devices, "art_init_devices");

  function art_upload_scene (scene : access constant Art_Scene_Desc) return int;
  pragma Import (C, art_upload_scene, "art_upload_scene");

  function art_resize (width, height : int) return int;
  pragma Import (C, art_resize, "art_resize");

  function art_render_pass (p : access constant Art_Pass_Params; accum_host : System.Address;
                            screen_host : System.Address; spp_inout : access int) return int;
  pragma Import (C, art_render_pass, "art_render_pass");

  function art_debug_hit_pass (p : access constant Art_Pass_Params; accum_host, screen_host : System.Address;
                               prim_index, mat_id, prim_type : System.Address) return int;
  pragma Import (C, art_debug_hit_pass, "art_debug_hit_pass");

  --  Tuning / build options by name (include/art_hip.h lists them), e.g.
  --    rc := art_set_option (Interfaces.C.Strings.New_String ("bvh_builder"), ART_BVH_HOST_SAH);
  --  BVH options take effect at the next art_upload_scene.
  function art_set_option (name : Interfaces.C.Strings.chars_ptr; value : Interfaces.Integer_64) return int;
  pragma Import (C, art_set_option, "art_set_option");

  --  Scene.Find_Closest_Hit for a list of rays -- SURVEY 8(b)'s "per-ray fallback for debugging": origins / dirs are 3 * n C floats
  --  ('Address of element 0), tfar may be Null_Address (unbounded), hits points at n Art_Hit records, stats may be Null_Address.
  function art_trace_rays (origins, dirs, tfar : System.Address; n : Interfaces.Integer_64; hits : System.Address;
                           kernel : int; stats : System.Address) return int;
  pragma Import (C, art_trace_rays, "art_trace_rays");

  function art_last_error return Interfaces.C.Strings.chars_ptr;
  pragma Import (C, art_last_error, "art_last_error");

  procedure art_shutdown;
  pragma Import (C, art_shutdown, "art_shutdown");

end Art_Hip;

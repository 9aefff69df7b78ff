--  Replacement body of Ray_Tracer.Render_Pass (ray_tracer.adb:240-293) that forwards one pass to the HIP backend.
--  Everything the reference does between "first call only: create tasks" (:264) and the LDR resolve (:281-291)
--  happens inside art_render_pass; g_accBuff / screen_buffer keep their Ada (x, y) layout (ray_tracer.ads:35, 54),
--  the backend writes them in that layout (ART_LAYOUT_ADA_XY).  Syntax-reviewed only (no GNAT in the build image).
--
--  In ray_tracer.adb:   with Art_Hip; with Interfaces.C; use Interfaces.C;   and replace the body of Render_Pass by:

  procedure Render_Pass is
    use type Interfaces.C.int;
    p   : aliased Art_Hip.Art_Pass_Params;
    spp : aliased Interfaces.C.int := Interfaces.C.int (g_spp.all);
    rc  : Interfaces.C.int;
  begin
    p.render_type := Interfaces.C.int (Render_Type'Pos (g_rend_type));
    p.aa_on       := Boolean'Pos (Anti_Aliasing_On);
    p.max_depth   := Interfaces.C.int (Max_Trace_Depth);
    p.vthreads    := Interfaces.C.int (Threads_Num);
    p.background  := (Interfaces.C.C_float (Background_Color.x), Interfaces.C.C_float (Background_Color.y),
                      Interfaces.C.C_float (Background_Color.z));
    p.seed        := 1;                               --  replaces Float_Random.Reset per task (ray_tracer.adb:147)
    p.layout      := Art_Hip.ART_LAYOUT_ADA_XY;

    if g_rend_type = RT_DEBUG or g_rend_type = RT_WHITTED then
      rc := Art_Hip.art_debug_hit_pass (p'Access, g_accBuff (0, 0)'Address, screen_buffer (0, 0)'Address,
                                        System.Null_Address, System.Null_Address, System.Null_Address);
      g_finish := true;                               --  ray_tracer.adb:259
    else
      rc := Art_Hip.art_render_pass (p'Access, g_accBuff (0, 0)'Address, screen_buffer (0, 0)'Address, spp'Access);
      g_spp.all := Integer (spp);                     --  spp advanced by Threads_Num * (4 | 1), ray_tracer.adb:168-175
    end if;

    if rc /= 0 then
      Put_Line ("art_hip: " & Interfaces.C.Strings.Value (Art_Hip.art_last_error));
    end if;
  end Render_Pass;

#!/usr/bin/env python3
"""Where k_shade_compact loses its lanes: run a bench workload on the -DART_LANE_PROBE build and print, per probe point of art_shade.h /
art_isect.h, how many waves passed and with how many enabled lanes on average.
  make -C ada-ray-tracer_amd OUT=libart_hip_probe.so BUILD=build_probe EXTRA=-DART_LANE_PROBE libart_hip_probe.so
  ART_LIB=ada-ray-tracer_amd/libart_hip_probe.so python profiles/lane_probe.py [c3|c4|c5]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

NAMES = {0: "shade_item entry", 1: "resolve pending shadow", 2: "alive", 3: "material known", 4: "hit a light", 5: "surface path (light sampling + BSDF)",
         6: "before light_sample", 7: "before bsdf_eval", 8: "before bsdf_sample", 9: "after bsdf_sample", 10: "light_sample entry", 11: "rect light",
         12: "inside the sphere light", 13: "cone sampling", 14: "frame branch x > y", 15: "frame built", 16: "ray-sphere disc >= 0", 17: "light_sample end",
         20: "bsdf Lambert", 21: "bsdf Mirror", 22: "bsdf Glass", 23: "bsdf Phong", 24: "lobe_to_world", 25: "lobe fix-up", 30: "emit_ray entry", 31: "emit_ray live",
         32: "emit_ray slab_setup", 40: "output decision", 41: "output item written", 50: "isect_sphere", 51: "isect_sphere disc >= 0",
         52: "isect_cornell", 53: "isect_cornell hit"}


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "c4"
    art = ge.load_package()
    from ada_ray_tracer_amd import scenes
    be = art.Backend(0)
    if scene == "c4":
        sd, W, H, T = scenes.synthetic_scene(1000000, 4), 1920, 1080, 16
    elif scene == "c3":
        sd, W, H, T = scenes.synthetic_scene(100000, 3), 1024, 1024, 16
    else:
        sd, W, H, T = scenes.mixed_scene(20000, 5), 4096, 4096, 2
    be.upload_scene(sd)
    be.resize(W, H)
    lib = be.lib
    buf = (C.c_uint64 * 128)()
    lib.art_debug_lane_probe(buf, 128)            # clear
    be.render_pass_device(art.Backend.pass_params(art.PT_MIS, True, 8, T, seed=1), 0)
    be.synchronize()
    assert lib.art_debug_lane_probe(buf, 128) == 0
    print("scene", scene)
    for k in range(64):
        lanes, visits = buf[2 * k], buf[2 * k + 1]
        if visits:
            print("  %2d %-40s waves %12d  lanes %5.1f" % (k, NAMES.get(k, "?"), visits, lanes / visits))
    be.shutdown()


if __name__ == "__main__":
    main()

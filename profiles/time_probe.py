#!/usr/bin/env python3
"""Where a wave of k_shade_compact spends its wall time: run a bench workload on the -DART_TIME_PROBE build and print, per time probe, the
shader-clock cycles the waves spent between the previous probe and this one (sum, share, average per visit).
  make -C ada-ray-tracer_amd OUT=libart_hip_tprobe.so BUILD=build_tprobe EXTRA=-DART_TIME_PROBE libart_hip_tprobe.so
  ART_LIB=ada-ray-tracer_amd/libart_hip_tprobe.so python profiles/time_probe.py [c3|c4|c5|s4]
(The probes add two launches around every stage launch and a few instructions per probe: the build is not timing evidence for anything else.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

NAMES = {64: "classification loads (3 batches)", 65: "class ranks (ballots, LDS atomics)", 66: "barrier 1", 67: "sort + output reservation + barrier 2",
         70: "round bookkeeping / tail of the previous item", 71: "item loads arrive", 72: "surface, material, Philox", 73: "light sample, bsdf_eval, shadow ray",
         74: "bsdf_sample (to reconvergence)", 75: "fold record + output words stored", 76: "extension ray: analytic bound, record, copy-out",
         77: "shadow ray: analytic bound, record, copy-out", 68: "after the last round"}


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "c4"
    art = ge.load_package()
    from ada_ray_tracer_amd import scenes
    be = art.Backend(0)
    if scene == "c4":
        sd, W, H, T = scenes.synthetic_scene(1000000, 4), 1920, 1080, 16
    elif scene == "c3":
        sd, W, H, T = scenes.synthetic_scene(100000, 3), 1024, 1024, 16
    elif scene == "s4":
        sd, W, H, T = scenes.structured_scene(1000000), 1920, 1080, 16
    else:
        sd, W, H, T = scenes.mixed_scene(20000, 5), 4096, 4096, 2
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        be.set_option(k, int(v))
    be.upload_scene(sd)
    be.resize(W, H)
    lib = be.lib
    buf = (C.c_uint64 * 192)()
    lib.art_debug_lane_probe(buf, 192)            # clear
    be.render_pass_device(art.Backend.pass_params(art.PT_MIS, True, 8, T, seed=1), 0)
    be.synchronize()
    assert lib.art_debug_lane_probe(buf, 192) == 0
    tot = sum(buf[2 * k] for k in range(64, 96))
    print("scene", scene, " total wave-cycles between probes %.3e" % tot)
    for k in list(range(64, 68)) + list(range(70, 78)) + [68]:
        cyc, visits = buf[2 * k], buf[2 * k + 1]
        if visits:
            print("  %2d %-58s visits %10d  avg %9.0f cycles  share %5.1f %%" % (k, NAMES.get(k, "?"), visits, cyc / visits, 100.0 * cyc / tot))
    be.shutdown()


if __name__ == "__main__":
    main()

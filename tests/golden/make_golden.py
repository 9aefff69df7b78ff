"""Regenerates the committed fixtures in tests/golden/.  Run from the repo root in the BUILD container
(it reads /root/reference for the data files only; nothing here executes reference code -- the reference is
Ada and cannot be built in this image):

    python tests/golden/make_golden.py

  pyramid2.vsgf               copy of the reference's data file data/pyramid2.vsgf (a mesh, not source)
  vsgf_decode.json            header + decoded arrays of that file, by an independent numpy decoder
  reference_image_patches.json  mean LDR colour of five mid-surface patches of the reference's own output
                              picture image.png (README.md:2) -- the statistical pin of the oracle
  reference_image_blocks.npz  the same picture reduced to 128x128 block means (8x8 pixels each, value * 64 as uint16) and the
                              per-block standard deviation -- the whole-picture pin: the picture turned out to be the HEAD scene
                              seen from (0, 2.55, 11) instead of scene.adb:212's (0, 2.55, 12.5)
  cornell_debug_64.npz        RT_DEBUG ids of the internal scene at 64x64 from the oracle (regression pin)
  c1_debug_64.npz             RT_DEBUG ids of the C1 eight-sphere scene (scenes.eight_sphere_scene) at 64x64 from the oracle
  cornell_mis_32.npz          PT_MIS accum of the internal scene, 32x32, 2 passes x 4 spp, seed 1 (regression pin)
"""
import json
import os
import shutil
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference"


def main():
    shutil.copyfile(os.path.join(REF, "data", "pyramid2.vsgf"), os.path.join(HERE, "pyramid2.vsgf"))
    raw = open(os.path.join(HERE, "pyramid2.vsgf"), "rb").read()
    size, nv, ni, nm, flags = struct.unpack_from("<qiiii", raw, 0)
    off = 24
    pos = np.frombuffer(raw, "<f4", nv * 4, off).reshape(nv, 4); off += nv * 16
    nrm = np.frombuffer(raw, "<f4", nv * 4, off).reshape(nv, 4); off += nv * 16
    uv = np.frombuffer(raw, "<f4", nv * 2, off).reshape(nv, 2); off += nv * 8
    if flags:
        off += nv * 16
    idx = np.frombuffer(raw, "<i4", ni, off); off += ni * 4
    mid = np.frombuffer(raw, "<i4", ni // 3, off); off += (ni // 3) * 4
    assert off == len(raw) == size
    json.dump(dict(fileSizeInBytes=size, verticesNum=nv, indicesNum=ni, materialsNum=nm, flags=flags,
                   positions=pos[:, :3].tolist(), normals=nrm[:, :3].tolist(), indices=idx.tolist(), material_ids=mid.tolist()),
              open(os.path.join(HERE, "vsgf_decode.json"), "w"), indent=1)

    from PIL import Image
    ref = np.asarray(Image.open(os.path.join(REF, "image.png")).convert("RGB")).astype(np.float64)

    def patch(x, y, r):
        return [round(float(v), 3) for v in ref[y - r:y + r, x - r:x + r].mean((0, 1))]
    # pixel positions in image.png (1024x1024, origin top-left) of mid-surface points of diffuse walls
    patches = dict(green_wall=patch(180, 520, 20), red_wall=patch(845, 520, 20), floor_front=patch(512, 900, 20), ceiling_front=patch(512, 130, 15))
    json.dump(dict(source="image.png (1024x1024), README.md:2", note="camera of this picture differs from scene.adb:212 at HEAD; "
                   "diffuse-wall radiance is view independent, so mid-wall patches are compared", patches=patches),
              open(os.path.join(HERE, "reference_image_patches.json"), "w"), indent=1)

    blk = ref.reshape(128, 8, 128, 8, 3)
    np.savez_compressed(os.path.join(HERE, "reference_image_blocks.npz"), mean64=np.round(blk.mean((1, 3)) * 64.0).astype(np.uint16),
                        std64=np.round(blk.std((1, 3)) * 64.0).astype(np.uint16))

    import orc
    cs = orc.CornellScene()
    _, prim, mat, ptype = orc.debug_pass(cs.scene, orc.make_params(64, 64, orc.RT_DEBUG, False))
    np.savez_compressed(os.path.join(HERE, "cornell_debug_64.npz"), prim=prim, mat=mat, ptype=ptype)
    import __graft_entry__ as ge
    import conv
    art = ge.load_package()
    from ada_ray_tracer_amd import scenes
    c1 = conv.OracleScene(scenes.eight_sphere_scene())
    _, prim, mat, ptype = orc.debug_pass(c1.scene, orc.make_params(64, 64, orc.RT_DEBUG, False))
    np.savez_compressed(os.path.join(HERE, "c1_debug_64.npz"), prim=prim, mat=mat, ptype=ptype)
    acc, spp, cnt = orc.render(cs.scene, orc.make_params(32, 32, orc.PT_MIS, True, 8, 1, seed=1), passes=2)
    np.savez_compressed(os.path.join(HERE, "cornell_mis_32.npz"), accum_bits=acc.view(np.uint32), spp=spp, rays=cnt.rays)
    print("golden fixtures written")


if __name__ == "__main__":
    main()

"""Statistical pin of the oracle against the reference's own output picture (image.png, README.md:2).
The picture's camera differs from scene.adb:212 at HEAD, so whole-image comparison is impossible; diffuse wall radiance is
view independent, so the LDR colour of mid-surface patches is compared.  fixture: tests/golden/reference_image_patches.json."""
import ctypes as C
import json

import numpy as np

import orc
import picture_pin


def test_oracle_reproduces_the_reference_picture():
    """The whole picture, block by block and region by region (tests/picture_pin.py): glass sphere, Phong sphere and wall, sphere
    light (cone sampling + MIS), the pyramid mesh, shadows, the caustic.  256x256 at 256 spp, a few seconds."""
    cs = orc.CornellScene()
    cs.scene.cam_pos = (C.c_float * 3)(*picture_pin.PICTURE_CAMERA)
    acc, spp, _ = orc.render(cs.scene, orc.make_params(256, 256, orc.PT_MIS, True, 8, 64, seed=1))
    assert spp == 256
    picture_pin.compare_with_reference_picture(picture_pin.ldr_rgb_top_left(orc.resolve(acc, spp)), "oracle 256x256x256spp")


def test_converged_cornell_render_matches_reference_picture_patches():
    ref = json.load(open(orc.GOLDEN + "/reference_image_patches.json"))["patches"]
    cs = orc.CornellScene()
    acc, spp, _ = orc.render(cs.scene, orc.make_params(256, 256, orc.PT_MIS, True, 8, 24, seed=1))
    img = orc.resolve(acc, spp)
    rgb = np.stack([img & 255, (img >> 8) & 255, (img >> 16) & 255], -1).astype(np.float64)[::-1]     # top-left origin like the PNG

    def patch(x, y, r):
        return rgb[y - r:y + r, x - r:x + r].mean((0, 1))
    mine = dict(green_wall=patch(59, 128, 5), red_wall=patch(196, 128, 5), floor_front=patch(128, 205, 5), ceiling_front=patch(128, 50, 4))
    for k, want in ref.items():
        got = mine[k]
        for c in range(3):
            if want[c] < 1.0:
                assert got[c] < 2.0, (k, got, want)
            else:
                assert abs(got[c] - want[c]) / want[c] < 0.08, (k, got, want)

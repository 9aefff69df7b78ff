"""The PRODUCT against the reference's own output picture (image.png, README.md:2; fixture + tolerances in tests/picture_pin.py):
the HEAD scene rendered by libart_hip.so at the picture's own resolution, 1024x1024, PT_MIS, AA on, depth 8, from the camera the
picture was taken with, resolved to LDR on the device, compared block by block and region by region.  This is the only check in the
suite that does not go through the oracle: a misreading of the Ada source shared by the oracle and the product would show here."""
import numpy as np
import pytest

import picture_pin

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,vthreads", [(1024, 256), (512, 64)])
def test_product_reproduces_the_reference_picture(art, backend, size, vthreads):
    from ada_ray_tracer_amd import scenes
    sd = scenes.reference_scene(cam_pos=picture_pin.PICTURE_CAMERA)      # Scene.Init by the product's own host layer: no oracle anywhere in this test
    backend.upload_scene(sd.desc)
    backend.resize(size, size)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, vthreads, seed=1)
    _, screen, spp = backend.render_pass(p, 0, want_accum=False, want_screen=True)
    assert spp == 4 * vthreads
    stats = picture_pin.compare_with_reference_picture(picture_pin.ldr_rgb_top_left(screen), "product %dx%d x %d spp" % (size, size, spp))
    print(stats)

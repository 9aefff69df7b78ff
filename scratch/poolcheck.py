import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
import __graft_entry__ as ge
art = ge.load_package()
import conv, orc
from ada_ray_tracer_amd import scenes
from test_gpu_parity import _assert_hits_equal, _random_rays, bits
be = art.Backend(0)
for n in (1, 3, 300, 20000):
    sd = scenes.synthetic_scene(n, 3); osc = conv.OracleScene(sd)
    be.upload_scene(sd)
    o, d = _random_rays(30000, n)
    _assert_hits_equal(be.trace_rays(o, d, kernel=art.TRACE_POOL), orc.closest_hits(osc.scene, o, d))
    print("hits ok", n, flush=True)
be.set_option("trace_kernel", art.TRACE_POOL)
for rt in ("PT_MIS", "PT_SHADOW", "PT_STUPID"):
    for sd, seed in ((scenes.synthetic_scene(2000, 3), 3), (scenes.mixed_scene(1500, 5), 6)):
        osc = conv.OracleScene(sd)
        be.upload_scene(sd); be.resize(64, 64)
        accum, _, spp = be.render_pass(art.Backend.pass_params(getattr(art, rt), True, 8, 1, seed=seed), 0)
        ref, _, cnt = orc.render(osc.scene, orc.make_params(64, 64, getattr(orc, rt), True, 8, 1, seed=seed))
        assert np.array_equal(bits(accum), bits(ref)) and be.stats().rays == cnt.rays
    print("render ok", rt, flush=True)
be.set_option("lds_stack_cap", 5)
sd = scenes.synthetic_scene(20000, 3); osc = conv.OracleScene(sd)
be.upload_scene(sd)
o, d = _random_rays(30000, 5)
_assert_hits_equal(be.trace_rays(o, d, kernel=art.TRACE_POOL), orc.closest_hits(osc.scene, o, d))
print("overflow ok")

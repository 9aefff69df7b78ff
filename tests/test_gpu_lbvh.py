"""SURVEY 8(f) rank 1 -- the BVH built ON the GPU (option bvh_builder = 1, csrc/art_lbvh.hip) in place of Embree's
rtcCommitScene (embree_connect.cpp:241-244).  The bar is the same as for the host build: the tree may be any tree, the
search result may not change -- hits bit-identical to the oracle's brute-force scan, images bit-identical to the ones
rendered through the host-built tree, and the exported tree is structurally sound (every triangle exactly once, every
child box encloses what hangs below it)."""
import numpy as np
import pytest

import conv
import orc
from test_gpu_parity import _assert_hits_equal, _random_rays, bits

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(1, 4), (1, 8), (2, 4), (2, 8), (3, 4), (3, 8)], ids=["lbvh-w4", "lbvh-w8", "ploc-w4", "ploc-w8", "sah-w4", "sah-w8"])
def gpu_builder(art, backend, request):
    """The GPU builders (1 = LBVH radix tree, 2 = PLOC clustering, 3 = binned SAH: the default since round 3) for both node widths"""
    backend.set_option("bvh_builder", request.param[0])
    backend.set_option("bvh_width", request.param[1])
    backend.builder_under_test = request.param[0]
    yield backend
    backend.set_option("bvh_builder", art.DEFAULT_BVH_BUILDER)
    backend.set_option("bvh_width", 4)


def _check_tree(nodes, tris, info, mesh_pos, mesh_idx):
    W = info.node_width
    nodes = nodes.reshape(-1, 8 * W); tris = tris.reshape(-1, 12)
    n = tris.shape[0]
    prim = tris[:, 9].view(np.int32)
    assert np.array_equal(np.sort(prim), np.arange(n)), "triangle records are not a permutation of the input"
    corners = mesh_pos.reshape(-1, 3)[mesh_idx.reshape(-1, 3)[prim]].reshape(n, 9)
    assert np.array_equal(tris[:, :9].view(np.uint32), corners.astype(np.float32).view(np.uint32))
    tlo = tris[:, :9].reshape(n, 3, 3).min(axis=1); thi = tris[:, :9].reshape(n, 3, 3).max(axis=1)
    seen = np.zeros(n, np.int32)
    visited = np.zeros(nodes.shape[0], np.int32)
    max_depth_stack = 0

    def walk(ni, stack_before):
        nonlocal max_depth_stack
        visited[ni] += 1
        nd = nodes[ni]
        ref = nd[3:4 * W:4].view(np.int32); cnt = nd[4 * W + 3:8 * W:4].view(np.int32)
        nch = int((ref >= 0).sum())
        assert nch >= 1 and np.all(ref[:nch] >= 0) and np.all(ref[nch:] < 0), "children are not packed to the front"
        here = stack_before + nch - 1
        max_depth_stack = max(max_depth_stack, here + 1)
        lo_all = np.full(3, np.inf, np.float32); hi_all = np.full(3, -np.inf, np.float32)
        for j in range(nch):
            lo = nd[4 * j:4 * j + 3]; hi = nd[4 * W + 4 * j:4 * W + 4 * j + 3]
            if cnt[j] > 0:
                assert cnt[j] <= W
                r = slice(int(ref[j]), int(ref[j]) + int(cnt[j]))
                seen[r] += 1
                clo, chi = tlo[r].min(axis=0), thi[r].max(axis=0)
            else:
                clo, chi = walk(int(ref[j]), here)
            assert np.all(lo < clo) and np.all(hi > chi), "child box does not strictly enclose its subtree"
            lo_all = np.minimum(lo_all, clo); hi_all = np.maximum(hi_all, chi)
        return lo_all, hi_all

    import sys
    sys.setrecursionlimit(10000)
    walk(0, 0)
    assert np.all(seen == 1), "a triangle is referenced %s times" % np.unique(seen)
    assert np.all(visited == 1) and nodes.shape[0] == info.n_nodes
    assert max_depth_stack <= info.max_stack, "stack bound %d below the real worst case %d" % (info.max_stack, max_depth_stack)


@pytest.mark.parametrize("ntris", [2, 9, 1000, 50000])
def test_lbvh_tree_is_well_formed(art, gpu_builder, ntris):
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(ntris, 3)
    gpu_builder.upload_scene(sd)
    nodes, tris, info = gpu_builder.export_bvh()
    assert info.n_tris == ntris and info.build_ms > 0.0
    pos, _, idx, _, _ = [a for a, m in zip(sd._mesh_arrays, sd.meshes) if m.mode == art.MESH_CLOSEST][0]
    _check_tree(nodes, tris, info, pos, idx)


@pytest.mark.parametrize("m", [4, 16, 128])
def test_lbvh_regular_grid_with_one_triangle_leaves(art, gpu_builder, m):
    """A regular tessellation with n = 2 m^2 = 2^k triangles gives a perfectly balanced radix tree; with 1-triangle leaves the wide
    tree then needs about 2n/3 nodes (21 for n = 32 at width 4, 21845 for n = 32768) -- more than the n/2 + 2 the builder
    used to allocate.  The tree must be complete and sound, and the hits equal the brute-force scan."""
    from ada_ray_tracer_amd import scenes
    mesh = scenes.grid_mesh(m)
    mats = scenes.cornell_materials()
    lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    sd = art.SceneDesc([], lights, mats, [mesh], None, scenes.REFERENCE_CAMERA)
    gpu_builder.set_option("bvh_max_leaf", 1)
    try:
        gpu_builder.upload_scene(sd)
    finally:
        gpu_builder.set_option("bvh_max_leaf", 0)
    nodes, tris, info = gpu_builder.export_bvh()
    n = 2 * m * m
    assert info.n_tris == n
    _check_tree(nodes, tris, info, np.asarray(mesh["pos"], np.float32), np.asarray(mesh["idx"], np.int32))
    if info.node_width == 4 and m >= 16 and gpu_builder.builder_under_test in (1, 2):      # the SAH collapse fills its 4-wide nodes: about n / 3
        assert info.n_nodes > n // 2 + 2, "this case is meant to exceed the old capacity (got %d nodes for %d triangles)" % (info.n_nodes, n)
    o, d = _random_rays(4000, m)
    _assert_hits_equal(gpu_builder.trace_rays(o, d), orc.closest_hits(conv.OracleScene(sd).scene, o, d))


@pytest.mark.parametrize("kernel", ["TRACE_COOP", "TRACE_SIMPLE"])
@pytest.mark.parametrize("ntris", [2, 9, 300, 20000])
def test_lbvh_hits_match_brute_force(art, gpu_builder, kernel, ntris):
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(ntris, 3)
    osc = conv.OracleScene(sd)
    gpu_builder.upload_scene(sd)
    o, d = _random_rays(30000, ntris + 1)
    _assert_hits_equal(gpu_builder.trace_rays(o, d, kernel=getattr(art, kernel)), orc.closest_hits(osc.scene, o, d))


def test_lbvh_equal_morton_codes_and_duplicate_triangles(art, gpu_builder):
    """Many triangles with the same centroid (equal Morton codes: the radix tree falls back to the sorted position) and exact
    duplicates (equal t: the lowest triangle index must win, as in the oracle's index-order scan)."""
    from ada_ray_tracer_amd import scenes
    mesh = scenes.random_triangles(64, 5)
    pos = np.asarray(mesh["pos"], np.float32).reshape(-1, 3); idx = np.asarray(mesh["idx"], np.int32).reshape(-1, 3)
    reps = 40
    pos2 = np.tile(pos, (reps, 1)); idx2 = np.concatenate([idx + k * pos.shape[0] for k in range(reps)])
    m2 = dict(mesh); m2["pos"] = pos2.ravel(); m2["nrm"] = np.tile(np.asarray(mesh["nrm"], np.float32).reshape(-1, 3), (reps, 1)).ravel()
    m2["idx"] = idx2.ravel(); m2["matid"] = np.tile(np.asarray(mesh["matid"], np.int32), reps)
    mats = scenes.cornell_materials()
    lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    sd = art.SceneDesc([], lights, mats, [m2], None, scenes.REFERENCE_CAMERA)
    osc = conv.OracleScene(sd)
    gpu_builder.upload_scene(sd)
    o, d = _random_rays(20000, 4)
    gh = gpu_builder.trace_rays(o, d); oh = orc.closest_hits(osc.scene, o, d)
    _assert_hits_equal(gh, oh)
    prim = np.array([h.prim_index for h in gh if h.is_hit])
    assert prim.size > 100 and prim.max() < 64, "a duplicate with a higher index won a tie"


def test_lbvh_counters_match_oracle_walk_of_the_exported_tree(art, gpu_builder):
    from ada_ray_tracer_amd import scenes
    mesh = scenes.random_triangles(20000, 78)
    mats = scenes.cornell_materials()
    lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    sd = art.SceneDesc([], lights, mats, [mesh], None, scenes.REFERENCE_CAMERA)
    gpu_builder.upload_scene(sd)
    nodes, tris, info = gpu_builder.export_bvh()
    o, d = _random_rays(40000, 9)
    t, prim, cnt = orc.bvh_walk(nodes, tris, o, d, width=info.node_width)
    for kernel in (art.TRACE_COOP, art.TRACE_SIMPLE):
        hits, st = gpu_builder.trace_rays(o, d, kernel=kernel, want_stats=True)
        gprim = np.array([h.prim_index if h.is_hit else -1 for h in hits], np.int32)
        assert np.array_equal(gprim, prim)
        assert (st.box_tests, st.tri_tests, st.node_visits, st.leaf_visits) == (cnt.box_tests, cnt.tri_tests, cnt.node_visits, cnt.leaf_visits)


@pytest.mark.parametrize("config", ["c3", "c5"])
def test_image_does_not_depend_on_the_builder(art, backend, config):
    """Same scene through the host SAH tree and the GPU LBVH tree: the accum buffers are bit-identical and so are the ray
    counts (the closest hit is a minimum over (t, key) whatever the traversal order)."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(100000, 3) if config == "c3" else scenes.mixed_scene(20000, 5)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=21)
    out = []
    try:
        for builder in (0, 1, 2, 3):
            backend.set_option("bvh_builder", builder)
            backend.upload_scene(sd)
            backend.resize(160, 90)
            accum, screen, spp = backend.render_pass(p, 0, want_screen=True)
            out.append((accum.copy(), screen.copy(), backend.stats().rays))
    finally:
        backend.set_option("bvh_builder", art.DEFAULT_BVH_BUILDER)
    for k in (1, 2, 3):
        assert np.array_equal(bits(out[0][0]), bits(out[k][0])) and np.array_equal(out[0][1], out[k][1]) and out[0][2] == out[k][2]


def test_spatial_split_builder_same_image(art, backend):
    """Host builder with reference splitting (option bvh_spatial_splits): duplicated triangle references, same image."""
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(100000, 3)
    p = art.Backend.pass_params(art.PT_MIS, True, 8, 2, seed=22)
    out = []
    try:
        backend.set_option("bvh_builder", 0)                       # reference splitting is an option of the host builder
        for spatial in (0, 1):
            backend.set_option("bvh_spatial_splits", spatial)
            backend.upload_scene(sd)
            backend.resize(160, 90)
            accum, _, spp = backend.render_pass(p, 0)
            out.append((accum.copy(), backend.stats().rays, backend.bvh_info().n_tris))
    finally:
        backend.set_option("bvh_spatial_splits", 0)
        backend.set_option("bvh_builder", art.DEFAULT_BVH_BUILDER)
    assert out[1][2] > out[0][2] == 100000
    assert np.array_equal(bits(out[0][0]), bits(out[1][0])) and out[0][1] == out[1][1]


# ---- round 3: the GPU SAH builder (csrc/art_sah.hip) builds the HOST builder's tree -------------------------------------------------
def _tree_signature(nodes, tris, info):
    """Numbering-independent fingerprint of a wide tree: bottom-up, a node's hash mixes, in slot order, every child's box (bit
    patterns) with the prim ids of a leaf or the hash of an inner child.  Two trees get the same root hash iff they hold the same
    boxes, the same leaves and the same slot order everywhere (up to hash collisions)."""
    W = info.node_width
    nodes = np.asarray(nodes, np.float32).reshape(-1, 8 * W); tris = np.asarray(tris, np.float32).reshape(-1, 12)
    N = nodes.shape[0]
    prim = tris[:, 9].view(np.int32).astype(np.uint64)
    ref = nodes[:, 3:4 * W:4].view(np.int32); cnt = nodes[:, 4 * W + 3:8 * W:4].view(np.int32)
    box = np.concatenate([nodes[:, :4 * W].reshape(N, W, 4)[:, :, :3], nodes[:, 4 * W:].reshape(N, W, 4)[:, :, :3]], axis=2).view(np.uint32).astype(np.uint64)
    used = ref >= 0; inner = used & (cnt == 0); leaf = used & (cnt > 0)
    M = np.uint64(0x9E3779B97F4A7C15)

    def mix(h, v):
        h = (h ^ v) * M
        return h ^ (h >> np.uint64(29))

    order = [np.array([0])]
    while True:
        f = order[-1]
        kids = ref[f][inner[f]]
        if kids.size == 0:
            break
        order.append(kids)
    H = np.zeros(N, np.uint64)
    with np.errstate(over="ignore"):
        for lvl in reversed(order):
            h = np.full(lvl.size, 1469598103934665603, np.uint64)
            for j in range(W):
                u = used[lvl, j]
                hj = np.full(lvl.size, 7, np.uint64)
                for a in range(6):
                    hj = mix(hj, box[lvl, j, a])
                lf = leaf[lvl, j]; inn = inner[lvl, j]
                for k in range(W):
                    m = lf & (cnt[lvl, j] > k)
                    pk = np.zeros(lvl.size, np.uint64)
                    pk[m] = prim[(ref[lvl, j][m] + k)] + np.uint64(1)
                    hj = mix(hj, pk)
                ch = np.zeros(lvl.size, np.uint64)
                ch[inn] = H[ref[lvl, j][inn]]
                hj = mix(hj, ch)
                h = np.where(u, mix(h, hj), mix(h, np.uint64(3)))
            H[lvl] = h
    return int(H[0]), N, int(leaf.sum())


@pytest.mark.parametrize("width", [4, 8])
@pytest.mark.parametrize("scene", ["soup-9", "soup-2047", "soup-2048", "soup-2049", "soup-3000", "soup-4097", "soup-6145", "soup-100k", "soup-1000k", "mixed-20k", "grid-128", "torus-49k"])
def test_gpu_sah_builder_builds_the_host_builders_tree(art, backend, scene, width):
    """bvh_builder = 3 restates art_bvh.cpp breadth-first on the GPU.  Every split decision depends only on minima, maxima and counts over
    a node's SET of references, so the binary tree -- hence the wide tree, slot for slot -- must be the host's: same boxes (bits),
    same leaves, same slot order, same node count and stack bound; only the node numbering (breadth-first) and the order of the
    triangle records differ."""
    from ada_ray_tracer_amd import scenes
    if scene == "soup-1000k" and width == 8:
        pytest.skip("the C4 mesh is compared at the default width")
    if scene.startswith("soup"):
        sd = scenes.synthetic_scene(int(scene.split("-")[1].replace("k", "000")), 4 if scene == "soup-1000k" else 3)     # 2047 .. 6145: around the builder's 2048-reference chunks
    elif scene == "mixed-20k":
        sd = scenes.mixed_scene(20000, 5)
    else:
        mesh = scenes.grid_mesh(128) if scene.startswith("grid") else scenes.torus_mesh()
        lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
        sd = art.SceneDesc([], lights, scenes.cornell_materials(), [mesh], None, scenes.REFERENCE_CAMERA)
    sig = []
    try:
        backend.set_option("bvh_width", width)
        for builder in (0, 3):
            backend.set_option("bvh_builder", builder)
            backend.upload_scene(sd)
            nodes, tris, info = backend.export_bvh()
            sig.append((_tree_signature(nodes, tris, info), info.n_nodes, info.n_tris, info.max_stack))
    finally:
        backend.set_option("bvh_builder", art.DEFAULT_BVH_BUILDER)
        backend.set_option("bvh_width", 4)
    assert sig[0] == sig[1], "host %s, GPU %s" % (sig[0], sig[1])


def test_gpu_sah_builder_is_deterministic_and_fast_at_1m_triangles(art, backend):
    """The C4 mesh: two builds give the same bytes (scan-based numbering, no atomics in the layout), the tree is sound (the vectorised
    structural check of tests/bvh_check.py), and the build stays far below the 50 ms the review asked for."""
    import bvh_check
    from ada_ray_tracer_amd import scenes
    sd = scenes.synthetic_scene(1000000, 3)
    backend.set_option("bvh_builder", 3)
    out = []
    for _ in range(2):
        backend.upload_scene(sd)
        nodes, tris, info = backend.export_bvh()
        out.append((nodes.copy(), tris.copy(), info.n_nodes, info.max_stack, info.build_ms))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    pos, _, idx, _, _ = [a for a, m in zip(sd._mesh_arrays, sd.meshes) if m.mode == art.MESH_CLOSEST][0]
    rep = bvh_check.check_tree(out[0][0], out[0][1], out[0][2], out[0][3], 4, pos, idx)
    print("GPU SAH build of 1M triangles: %.2f / %.2f ms, %s" % (out[0][4], out[1][4], rep))
    assert max(out[0][4], out[1][4]) < 50.0


def test_gpu_sah_tree_of_a_structured_mesh_finds_the_brute_force_hits(art, backend):
    """A torus (49 k triangles: a closed, connected surface with shared vertices and triangles of very different sizes) through the default
    builder: the tree is sound, and the hits of 6000 rays equal the oracle's brute-force scan bit for bit.  The random soups of the bench
    scenes never produce coplanar neighbours or exact ties of the SAH costs; this does."""
    import bvh_check
    from ada_ray_tracer_amd import scenes
    mesh = scenes.torus_mesh()
    lights = [dict(shape=art.LIGHT_SPHERE, mat=4, center=(0.0, 4.5, 1.0), radius=0.5, intensity=(10.0, 10.0, 10.0), surfaceArea=3.14159)]
    sd = art.SceneDesc([], lights, scenes.cornell_materials(), [mesh], None, scenes.REFERENCE_CAMERA)
    backend.set_option("bvh_builder", 3)
    backend.upload_scene(sd)
    nodes, tris, info = backend.export_bvh()
    bvh_check.check_tree(nodes, tris, info.n_nodes, info.max_stack, info.node_width, mesh["pos"], mesh["idx"])
    o, d = _random_rays(6000, 31)
    _assert_hits_equal(backend.trace_rays(o, d), orc.closest_hits(conv.OracleScene(sd).scene, o, d))
    hit = [h for h in backend.trace_rays(o, d) if h.is_hit]
    assert len(hit) > 300

"""Structural check of an exported wide BVH, vectorised (numpy, level by level) so that it also runs on the 1M-triangle tree of C4.
Independent of every traversal: it only reads the node / triangle packets (layout: csrc/art_scene.h) and the caller's mesh.
  * the triangle records are a permutation of the input triangles (or, with spatial splits, cover every input triangle), corner for corner
  * every triangle record is referenced by exactly one leaf slot, every node except the root by exactly one inner slot
  * children are packed to the front of a node, leaves hold 1..W triangles
  * every child box STRICTLY encloses everything that hangs below it (true triangle bounds, not the stored boxes)
  * the traversal-stack bound the product reports is not below the real worst case of the published push order"""
import numpy as np


def check_tree(nodes, tris, n_nodes, max_stack, W, mesh_pos, mesh_idx, allow_duplicates=False):
    nodes = np.asarray(nodes, np.float32).reshape(-1, 8 * W); tris = np.asarray(tris, np.float32).reshape(-1, 12)
    N, n = nodes.shape[0], tris.shape[0]
    assert N == n_nodes
    prim = tris[:, 9].view(np.int32)
    ntri_mesh = np.asarray(mesh_idx).reshape(-1, 3).shape[0]
    if allow_duplicates:
        assert prim.min() >= 0 and prim.max() < ntri_mesh and np.unique(prim).size == ntri_mesh, "an input triangle has no record"
    else:
        assert n == ntri_mesh and np.array_equal(np.sort(prim), np.arange(n)), "triangle records are not a permutation of the input"
    corners = np.asarray(mesh_pos, np.float32).reshape(-1, 3)[np.asarray(mesh_idx, np.int32).reshape(-1, 3)[prim]].reshape(n, 9)
    assert np.array_equal(tris[:, :9].view(np.uint32), corners.view(np.uint32)), "a triangle record does not hold its triangle's corners"
    tlo = tris[:, :9].reshape(n, 3, 3).min(axis=1); thi = tris[:, :9].reshape(n, 3, 3).max(axis=1)

    ref = nodes[:, 3:4 * W:4].view(np.int32)                    # [N, W]
    cnt = nodes[:, 4 * W + 3:8 * W:4].view(np.int32)
    lo = nodes[:, :4 * W].reshape(N, W, 4)[:, :, :3]; hi = nodes[:, 4 * W:].reshape(N, W, 4)[:, :, :3]
    used = ref >= 0
    nch = used.sum(1)
    assert (nch >= 1).all(), "a node without children"
    assert (used == (np.arange(W)[None, :] < nch[:, None])).all(), "children are not packed to the front"
    leaf = used & (cnt > 0); inner = used & (cnt == 0)
    assert (cnt[used] >= 0).all() and (cnt[leaf] <= W).all()
    # every triangle record in exactly one leaf slot
    cover = np.zeros(n + 1, np.int64)
    np.add.at(cover, ref[leaf], 1); np.add.at(cover, ref[leaf] + cnt[leaf], -1)
    assert (ref[leaf] + cnt[leaf] <= n).all()
    seen = np.cumsum(cover)[:n]
    assert (seen == 1).all(), "a triangle record is referenced %s times" % np.unique(seen)
    # every node but the root referenced exactly once
    refs = np.bincount(ref[inner], minlength=N)
    assert refs[0] == 0 and (refs[1:] == 1).all() and ref[inner].max(initial=0) < N, "node references are not a tree"

    # depth of every node + stack occupancy before it is expanded (published order: a node's hits are pushed, one is popped)
    parent = np.full(N, -1, np.int64); parent[ref[inner]] = np.nonzero(inner)[0]
    depth = np.zeros(N, np.int64); before = np.zeros(N, np.int64)
    frontier = np.array([0]); level = 0; order = [frontier]
    while True:
        kids = ref[frontier][inner[frontier]]
        if kids.size == 0:
            break
        level += 1
        depth[kids] = level
        before[kids] = before[parent[kids]] + nch[parent[kids]] - 1
        frontier = kids; order.append(kids)
        assert level <= N
    assert sum(o.size for o in order) == N, "unreachable nodes"
    worst = int((before + nch).max())                          # entries on the stack right after a node's hits were pushed
    assert worst <= max_stack, "stack bound %d below the real worst case %d" % (max_stack, worst)

    # true bounds of what hangs below every slot, bottom-up
    sub_lo = np.full((N, 3), np.inf, np.float32); sub_hi = np.full((N, 3), -np.inf, np.float32)
    for lvl in reversed(order):
        s_lo = np.full((lvl.size, W, 3), np.inf, np.float32); s_hi = np.full((lvl.size, W, 3), -np.inf, np.float32)
        r, c, lf, inn = ref[lvl], cnt[lvl], leaf[lvl], inner[lvl]
        for k in range(W):                                      # leaf slots: min / max over their <= W triangles
            m = lf & (c > k)
            if m.any():
                t = r[m] + k
                s_lo[m] = np.minimum(s_lo[m], tlo[t]); s_hi[m] = np.maximum(s_hi[m], thi[t])
        s_lo[inn] = sub_lo[r[inn]]; s_hi[inn] = sub_hi[r[inn]]
        u = used[lvl]
        assert (lo[lvl][u] < s_lo[u]).all() and (hi[lvl][u] > s_hi[u]).all(), "a child box does not strictly enclose its subtree"
        sub_lo[lvl] = s_lo.min(1); sub_hi[lvl] = s_hi.max(1)
    return dict(nodes=N, records=n, depth=int(depth.max()), worst_stack=worst, leaf_slots=int(leaf.sum()), tris_per_leaf=float(n / max(1, leaf.sum())))

"""Structure checks of the GPU-built BVH (downloaded through the C ABI), run before any
traversal test: a malformed tree must show up here, not as a hung kernel."""
import numpy as np
import pytest

import scenes
from mipt import abi, host

pytestmark = pytest.mark.gpu


def validate_bvh(scene_host, dev_scene):
    info = dev_scene.info()
    nodes, v0, e1, e2, root, (qmin, qcell) = dev_scene.download_bvh()
    verts = scene_host.vertices()[:, :, :3]
    n = scene_host.num_triangles
    assert info.num_triangles == n and v0.shape[0] == n
    ids = v0[:, 3].copy().view(np.uint32).astype(np.int64)
    assert np.array_equal(np.sort(ids), np.arange(n)), "every triangle must sit in exactly one leaf slot"
    # leaf slots hold v0 and the fp32 edge differences of the input, bit for bit
    p0, p1, p2 = verts[ids, 0], verts[ids, 1], verts[ids, 2]
    assert np.array_equal(v0[:, :3], p0)
    assert np.array_equal(e1[:, :3], p1 - p0) and np.array_equal(e2[:, :3], p2 - p0)
    last = e1[:, 3].copy().view(np.uint32)
    assert set(np.unique(last)) <= {0, 1} and last[-1] == 1
    tri_lo = np.minimum(np.minimum(p0, p1), p2)
    tri_hi = np.maximum(np.maximum(p0, p1), p2)
    W = info.node_width
    assert nodes.shape[1] == 4 * W
    codes = nodes[:, 3 * W:]
    lo, hi = (nodes[:, :3 * W] & 0xFFFF).reshape(-1, W, 3), (nodes[:, :3 * W] >> 16).reshape(-1, W, 3)   # word = min | max << 16 per child and axis
    q = np.stack([lo, hi], axis=2).reshape(-1, 6 * W).astype(np.float64)   # per child: min xyz, max xyz
    qmin, qcell = qmin.astype(np.float64), qcell.astype(np.float64)
    seen = np.zeros(n, dtype=bool)
    max_depth = 0
    leaves = 0
    inner = 0
    if root == 0xFFFFFFFF:
        assert n == 0
        return info
    # child codes index one space: node slots first, leaf slots behind them
    leaf_base = nodes.shape[0]
    stack = [(int(root), 0, None, None)]
    while stack:
        code, depth, lo, hi = stack.pop()
        max_depth = max(max_depth, depth)
        if code >= leaf_base:
            s = code - leaf_base
            e = s
            while last[e] == 0:
                e += 1
            sl = slice(s, e + 1)
            assert not seen[sl].any(), "leaf ranges overlap"
            seen[sl] = True
            leaves += 1
            if lo is not None:      # the parent's (padded) box of this child contains the leaf's triangles
                assert (tri_lo[sl] >= lo).all() and (tri_hi[sl] <= hi).all()
            continue
        inner += 1
        nd = q[code]
        used = 0
        for k in range(W):
            ck = int(codes[code, k])
            clo, chi = qmin + nd[6 * k:6 * k + 3] * qcell, qmin + nd[6 * k + 3:6 * k + 6] * qcell
            if ck == 0xFFFFFFFF:
                assert (clo > chi).all(), "an unused child must have an empty box"
                continue
            used += 1
            assert (clo <= chi).all()
            if lo is not None:      # nested up to the padding both levels carry
                assert (clo >= lo - 1.01 * qcell).all() and (chi <= hi + 1.01 * qcell).all()
            stack.append((ck, depth + 1, clo, chi))
        assert used >= 2
    assert seen.all(), "some leaf slot is unreachable from the root"
    assert max_depth == info.max_depth and leaves == info.num_leaves and inner == info.num_nodes
    return info


def test_bvh_cornell(gpu, cornell):
    sc = gpu.Scene(cornell.desc)
    info = validate_bvh(cornell, sc)
    assert info.num_triangles == 32 and info.num_leaves >= 8


def test_bvh_atrium_small(gpu, atrium_small):
    sc = gpu.Scene(atrium_small.desc)
    info = validate_bvh(atrium_small, sc)
    assert info.num_leaves * 4 >= info.num_triangles          # leaves hold at most 4 triangles unless centroids coincide
    assert info.sah_cost > 0


def test_largest_nodes_sit_in_the_cached_slots(gpu, atrium_small, monkeypatch):
    """The traversal keeps the first 512 node slots in LDS; the builder exchanges the nodes with the largest boxes
    (surface of the union of a node's child boxes, which goes with the number of rays that visit it) into them.
    With MIPT_BUILD_NO_PROMOTE the slots stay as the level-by-level collapse hands them out; both trees are valid."""
    def areas(dev_scene):
        nodes, _, _, _, root, (qmin, qcell) = dev_scene.download_bvh()
        W = dev_scene.info().node_width
        codes = nodes[:, 3 * W:]
        w = nodes[:, :3 * W].reshape(-1, W, 3).astype(np.int64)
        valid = codes != 0xFFFFFFFF
        lo = np.where(valid[..., None], w & 0xFFFF, 1 << 20).min(axis=1).astype(np.float64)
        hi = np.where(valid[..., None], w >> 16, -1).max(axis=1).astype(np.float64)
        d = np.maximum(hi - lo, 0) * qcell.astype(np.float64)
        a = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
        return a, root

    sc = gpu.Scene(atrium_small.desc)
    validate_bvh(atrium_small, sc)
    a, root = areas(sc)
    assert a.shape[0] > 512, "the scene must have more node slots than the copy holds"
    assert a[:512].min() >= a[512:].max(), "a node beyond the cached slots has a larger box than one inside"
    assert root < 512 and a[root] == a.max()
    monkeypatch.setenv("MIPT_BUILD_NO_PROMOTE", "1")
    plain = gpu.Scene(atrium_small.desc)
    monkeypatch.delenv("MIPT_BUILD_NO_PROMOTE")
    validate_bvh(atrium_small, plain)
    b, _ = areas(plain)
    assert b[:512].min() < b[512:].max(), "without the exchange the top slots are not the largest nodes (else this test shows nothing)"
    assert np.isclose(np.sort(a).sum(), np.sort(b).sum(), rtol=1e-9)      # the same nodes, in other slots


def test_bvh_degenerate_inputs(gpu):
    # one triangle; many identical triangles (coincident centroids force splits by position);
    # zero-area and sliver triangles; an empty scene
    scene_from = scenes.scene_from_triangles

    import ctypes as C
    one = [[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]
    desc, keep = scene_from(one)
    sc = gpu.Scene(C.pointer(desc))
    tri, tuv, anyhit = sc.trace_rays(np.array([[0.2, 0.2, 1, 0, 0, -1], [2, 2, 1, 0, 0, -1]], dtype=np.float32))
    assert tri.tolist() == [0, -1] and anyhit.tolist() == [1, 0] and np.allclose(tuv[0], [1.0, 0.2, 0.2], atol=1e-7)

    same = one * 100
    desc, keep = scene_from(same)
    sc = gpu.Scene(C.pointer(desc))
    assert sc.info().num_triangles == 100
    tri, _, _ = sc.trace_rays(np.array([[0.2, 0.2, 1, 0, 0, -1]], dtype=np.float32))
    assert tri.tolist() == [0]                       # 100 exact ties: the lowest id wins

    odd = one + [[[0, 0, 0], [1, 1, 0], [2, 2, 0]], [[0, 0, -1], [1, 0, -1], [1, 1e-7, -1]], [[5, 5, 5], [5, 5, 5], [5, 5, 5]]]
    desc, keep = scene_from(odd)
    sc = gpu.Scene(C.pointer(desc))
    tri, _, _ = sc.trace_rays(np.array([[0.2, 0.2, 1, 0, 0, -1]], dtype=np.float32))
    assert tri.tolist() == [0]

    desc, keep = scene_from(np.zeros((0, 3, 3)))
    sc = gpu.Scene(C.pointer(desc))
    tri, _, anyhit = sc.trace_rays(np.array([[0.2, 0.2, 1, 0, 0, -1]], dtype=np.float32))
    assert tri.tolist() == [-1] and anyhit.tolist() == [0]

"""Closest-hit and any-hit parity of the HIP traversal against the CPU oracle's brute-force
search, through mipt_trace_rays.  Integer results (triangle ids, any-hit flags) and the
fp32 (t, u, v) of the winning triangle must be identical bit for bit."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import scenes
from conftest import GOLDEN
from mipt import abi, host

pytestmark = pytest.mark.gpu


def assert_trace_parity(dev_scene, oracle_scene, rays, eps=1e-20):
    tri, tuv, anyhit = dev_scene.trace_rays(rays, eps=eps)
    otri, otuv, oany = oracle_scene.trace_rays(rays, eps=eps, accel=ob.ACCEL_BRUTE)
    bad = np.nonzero(tri != otri)[0]
    assert bad.size == 0, f"{bad.size} of {len(tri)} closest-hit ids differ, first: ray {bad[:5]}, gpu {tri[bad[:5]]}, oracle {otri[bad[:5]]}"
    assert np.array_equal(anyhit, oany), f"{(anyhit != oany).sum()} any-hit flags differ"
    assert np.array_equal(tuv.view(np.uint32), otuv.view(np.uint32)), "t/u/v differ in their bits"
    return tri


def test_golden_ray_batch(gpu, cornell):
    g = np.load(os.path.join(GOLDEN, "cornell_rays_4096.npz"))
    sc = gpu.Scene(cornell.desc)
    tri, tuv, anyhit = sc.trace_rays(g["rays"])
    assert np.array_equal(tri, g["tri"]) and np.array_equal(anyhit, g["anyhit"])
    assert np.array_equal(tuv.view(np.uint32), g["tuv"].view(np.uint32))


def test_cornell_random_and_aimed_rays(gpu, cornell, cornell_oracle):
    sc = gpu.Scene(cornell.desc)
    tri = assert_trace_parity(sc, cornell_oracle, scenes.random_rays(cornell, 200000, seed=1))
    assert (tri >= 0).mean() > 0.2
    # both flavours of the determinant threshold (xnacollision.cpp:1073 vs PathTracer.cu:153)
    assert_trace_parity(sc, cornell_oracle, scenes.random_rays(cornell, 50000, seed=2), eps=1e-5)


def test_atrium_small_rays(gpu, atrium_small, atrium_small_oracle):
    sc = gpu.Scene(atrium_small.desc)
    tri = assert_trace_parity(sc, atrium_small_oracle, scenes.random_rays(atrium_small, 30000, seed=3))
    assert (tri >= 0).mean() > 0.5


def test_build_options_do_not_change_results(gpu, atrium_small, atrium_small_oracle):
    rays = scenes.random_rays(atrium_small, 8000, seed=4)
    for bins, leaf in [(4, 1), (32, 8), (8, 2)]:
        sc = gpu.Scene(atrium_small.desc, sah_bins=bins, max_leaf_triangles=leaf)
        assert_trace_parity(sc, atrium_small_oracle, rays)


def test_atrium_262k_rays(gpu):
    big = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 0, 1)
    osc = ob.OracleScene(big.desc)
    sc = gpu.Scene(big.desc)
    info = sc.info()
    assert info.num_triangles == big.num_triangles
    assert_trace_parity(sc, osc, scenes.random_rays(big, 3000, seed=7))


def test_empty_and_ragged_batches(gpu, cornell, cornell_oracle):
    sc = gpu.Scene(cornell.desc)
    tri, tuv, anyhit = sc.trace_rays(np.zeros((0, 6), dtype=np.float32))
    assert tri.size == 0
    for n in (1, 63, 64, 65, 127, 1000):           # sizes around the 64-lane wave
        assert_trace_parity(sc, cornell_oracle, scenes.random_rays(cornell, n, seed=100 + n))
    # non-finite and zero directions must not hang or fault, and must agree with the oracle
    rays = scenes.random_rays(cornell, 64, seed=9)
    rays[0, 3:] = 0
    rays[1, 3] = np.inf
    rays[2, 0] = np.nan
    rays[3, 3:] = [0, -1, 0]
    rays[4, 3:] = [1e-38, -1, 0]
    assert_trace_parity(sc, cornell_oracle, rays)


def test_rays_that_start_on_box_planes(gpu, atrium_small, atrium_small_oracle):
    """Origins exactly on planes of node boxes (entry distance +0 or -0 on that axis), directions along the axes and
    oblique: the sort key of a box the ray starts on must not turn into the farthest one (the sign bit of a -0 is
    no part of the key), so nothing behind such a box is dropped."""
    sc = gpu.Scene(atrium_small.desc)
    nodes, _, _, _, root, (qmin, qcell) = sc.download_bvh()
    W = sc.info().node_width
    words = nodes[:256, :3 * W].reshape(-1, W, 3)
    codes = nodes[:256, 3 * W:]
    rng = np.random.default_rng(21)
    rays = []
    for n in range(words.shape[0]):
        for k in range(W):
            if codes[n, k] == 0xFFFFFFFF:
                continue
            lo = qmin + (words[n, k] & 0xFFFF).astype(np.float32) * qcell
            hi = qmin + (words[n, k] >> 16).astype(np.float32) * qcell
            for axis in range(3):
                for plane in (lo, hi):
                    o = (lo + (hi - lo) * rng.random(3).astype(np.float32)).astype(np.float32)
                    o[axis] = plane[axis]
                    for d in (np.eye(3, dtype=np.float32)[axis], -np.eye(3, dtype=np.float32)[axis], rng.normal(0, 1, 3).astype(np.float32)):
                        rays.append(np.concatenate([o, d / np.linalg.norm(d)]))
    rays = np.asarray(rays, dtype=np.float32)[:60000]
    tri = assert_trace_parity(sc, atrium_small_oracle, rays)
    assert (tri >= 0).mean() > 0.5


def test_axis_parallel_rays_in_a_large_scene(gpu):
    """Coordinates in the millions and rays along the axes, exactly and with components of 1e-19 ... 1e-38: the box
    test replaces a direction component below 1e-20 and multiplies the grid's extent by the reciprocal, which has to
    stay finite (a product that overflowed would turn a box the ray runs through into a miss)."""
    import ctypes as C
    rng = np.random.default_rng(31)
    tris = []
    for i in range(40):           # small triangles scattered over +-3e6, facing every axis
        c = rng.uniform(-3e6, 3e6, 3)
        axis = i % 3
        a, b = (axis + 1) % 3, (axis + 2) % 3
        h = rng.uniform(1e3, 1e5)
        p0, p1, p2 = c.copy(), c.copy(), c.copy()
        p0[a] -= h; p0[b] -= h; p1[a] += h; p1[b] -= h; p2[b] += h
        tris.append([p0, p1, p2])
    desc, keep = scenes.scene_from_triangles(tris)
    sc = gpu.Scene(C.pointer(desc))
    osc = ob.OracleScene(C.pointer(desc))
    t = np.asarray(tris, dtype=np.float32)
    rays = []
    for i in range(t.shape[0]):
        axis = i % 3
        target = t[i].mean(axis=0)
        for sign in (-1.0, 1.0):
            for tiny in (0.0, 1e-19, -1e-19, 1e-21, -1e-25, 1e-30, -1e-38):
                d = np.zeros(3, dtype=np.float32)
                d[axis] = sign
                d[(axis + 1) % 3] = tiny
                d[(axis + 2) % 3] = -tiny
                o = target.copy()
                o[axis] -= sign * 5e6          # starts outside the scene's box, runs through the triangle
                rays.append(np.concatenate([o, d]))
    rays = np.asarray(rays, dtype=np.float32)
    tri = assert_trace_parity(sc, osc, rays)
    assert (tri >= 0).mean() > 0.9


def test_rays_from_far_outside_the_scene(gpu, atrium_small, atrium_small_oracle):
    """Origins 3 to 300 scene sizes away, aimed at vertices, edge midpoints and interior points of triangles: the
    rounding errors of the triangle test and of the box test grow with the distance, and so must the margin of the
    box test (it is a number of units in the last place of the ray's reach; built without that part,
    -DMIPT_BOX_MARGIN_ULPS=0, the search loses 95 of the 12 000 rays from 300 and 3000 scene sizes away).  From
    3000 scene sizes away, where one unit in the last place of the origin is a 4000th of the scene and the triangle
    test's own barycentric coordinates are off by 0.1, 3 of 6000 rays still differ from the brute-force search: not tested."""
    rng = np.random.default_rng(41)
    v = atrium_small.vertices()[:, :, :3]
    lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
    size = float(np.linalg.norm(hi - lo))
    rays = []
    for dist in (3.0, 30.0, 300.0):
        n = 6000
        tri = v[rng.integers(0, v.shape[0], n)]
        w = rng.random((n, 3)); w /= w.sum(axis=1, keepdims=True)
        kind = rng.integers(0, 3, n)
        target = np.einsum("nk,nkc->nc", w, tri)                                   # interior point
        target = np.where((kind == 1)[:, None], tri[:, 0], target)                 # a vertex
        target = np.where((kind == 2)[:, None], 0.5 * (tri[:, 0] + tri[:, 1]), target)   # an edge midpoint
        dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        o = (target - dirs * dist * size).astype(np.float32)
        d = target - o.astype(np.float64)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays.append(np.concatenate([o, d.astype(np.float32)], axis=1))
    rays = np.concatenate(rays).astype(np.float32)
    sc = gpu.Scene(atrium_small.desc)
    tri = assert_trace_parity(sc, atrium_small_oracle, rays)
    assert (tri >= 0).mean() > 0.5


def test_deep_tree_uses_the_stack_beyond_lds(gpu):
    """Nested shells of triangles whose sizes fall geometrically give a tall, thin tree; rays
    through the common centre enter every child box on the way down, so the per-lane stack grows
    past the part that is kept in LDS."""
    import ctypes as C
    tris = []
    for i in range(120):
        r = 100.0 * 0.8 ** i
        for sx in (-1.0, 1.0):          # a small triangle on either side of the centre, at distance r
            c = np.array([sx * r, 0.0, 0.0])
            h = 0.2 * r
            tris.append([c + [0, -h, -h], c + [0, h, -h], c + [0, 0, h]])
    desc, keep = scenes.scene_from_triangles(tris)
    sc = gpu.Scene(C.pointer(desc))
    osc = ob.OracleScene(C.pointer(desc))
    info = sc.info()
    assert (info.node_width - 1) * info.max_depth + 2 > 22, "tree too shallow to leave the LDS part of the stack"
    # rays start near the common centre and leave along +-x: the nearest children are the deepest
    # ones, so every level leaves its outer (far) children on the stack
    rng = np.random.default_rng(5)
    n = 4096
    rays = np.zeros((n, 6), dtype=np.float32)
    rays[:, :3] = rng.normal(0, 1e-12, (n, 3))
    d = np.stack([rng.choice([-1.0, 1.0], n), rng.normal(0, 0.1, n), rng.normal(0, 0.1, n)], axis=1)
    rays[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    tri = assert_trace_parity(sc, osc, rays)
    assert (tri >= 0).mean() > 0.1


def test_stack_slice_in_global_memory(gpu, atrium_small, atrium_small_oracle, monkeypatch):
    """With only two stack entries in LDS nearly every ray keeps part of its stack in the
    global-memory slice of its wave; results must not change."""
    monkeypatch.setenv("MIPT_STACK_LDS", "2")
    sc = gpu.Scene(atrium_small.desc)
    monkeypatch.delenv("MIPT_STACK_LDS")
    assert_trace_parity(sc, atrium_small_oracle, scenes.random_rays(atrium_small, 20000, seed=11))


def test_ray_count_beyond_the_queue_index_is_refused(gpu, cornell):
    # the queue head is a 32-bit counter that the persistent waves push past the ray count by up to one
    # reservation each: counts that could make it wrap are refused before anything is allocated or read
    import ctypes as C
    from mipt import abi
    from mipt.device import lib
    sc = gpu.Scene(cornell.desc)
    rays = np.zeros(6, dtype=np.float32)
    for n in ((1 << 31) - 64, (1 << 31), (1 << 40)):
        rc = lib().mipt_trace_rays(sc._h, rays.ctypes.data, C.c_int64(n), C.c_float(1e-20), None, None, None)
        assert rc == abi.E_LIMIT, (n, rc)
    assert b"too many rays" in lib().mipt_last_error()

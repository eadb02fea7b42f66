"""Known-answer tests that pin the CPU oracle (the reference itself holds no tests or golden
vectors for this path: parity unpinned, so the oracle is anchored on analytic answers)."""
import math

import numpy as np
import pytest

import oracle_binding as ob
import scenes
from mipt import abi, host

V0, V1, V2 = (0, 0, 0), (1, 0, 0), (0, 1, 0)


def test_mt_front_hit_barycentrics():
    # xnacollision.cpp:1068-1168: u along e1, v along e2, t distance
    hit, t, u, v = ob.intersect_ray_triangle((0.25, 0.5, 2), (0, 0, -1), V0, V1, V2)
    assert hit and t == 2.0 and u == 0.25 and v == 0.5


def test_mt_back_face_is_hit_two_sided():
    hit, t, u, v = ob.intersect_ray_triangle((0.25, 0.5, -3), (0, 0, 1), V0, V1, V2)
    assert hit and t == 3.0 and u == 0.25 and v == 0.5


def test_mt_edges_and_vertices_inclusive():
    for p in [(0, 0), (1, 0), (0, 1), (0.5, 0.5), (0.5, 0), (0, 0.5)]:
        hit, t, u, v = ob.intersect_ray_triangle((p[0], p[1], 1), (0, 0, -1), V0, V1, V2)
        assert hit and t == 1.0, p
    hit, *_ = ob.intersect_ray_triangle((0.5000001, 0.5000001, 1), (0, 0, -1), V0, V1, V2)
    assert not hit


def test_mt_behind_origin_rejected_and_t_zero_accepted():
    hit, *_ = ob.intersect_ray_triangle((0.25, 0.25, -1), (0, 0, -1), V0, V1, V2)
    assert not hit
    hit, t, _, _ = ob.intersect_ray_triangle((0.25, 0.25, 0), (0, 0, -1), V0, V1, V2)
    assert hit and t == 0.0


def test_mt_parallel_and_epsilon():
    hit, *_ = ob.intersect_ray_triangle((0.25, 0.25, 1), (1, 0, 0), V0, V1, V2)
    assert not hit                      # det == 0
    # |det| = 1e-6: accepted with the CPU epsilon 1e-20, rejected with the GPU flavour's 1e-5
    d = (math.sqrt(1 - 1e-12), 0.0, -1e-6)
    assert ob.intersect_ray_triangle((-1.75, 0.25, 2e-6), d, V0, V1, V2, eps=1e-20)[0]
    assert not ob.intersect_ray_triangle((-1.75, 0.25, 2e-6), d, V0, V1, V2, eps=1e-5)[0]


def test_mt_degenerate_triangle_never_hit():
    assert not ob.intersect_ray_triangle((0, 0, 1), (0, 0, -1), (0, 0, 0), (1, 1, 0), (2, 2, 0))[0]


def test_slab_basic_and_behind():
    # xnacollision.cpp:1228-1297
    hit, dist = ob.intersect_ray_aabb((0, 0, -5), (0, 0, 1), (0, 0, 0), (1, 1, 1))
    assert hit and dist == 4.0
    assert not ob.intersect_ray_aabb((0, 0, 5), (0, 0, 1), (0, 0, 0), (1, 1, 1))[0]
    hit, dist = ob.intersect_ray_aabb((0, 0, 0), (0, 0, 1), (0, 0, 0), (1, 1, 1))
    assert hit and dist == -1.0            # origin inside: t_min is negative, still a hit


def test_slab_parallel_axes():
    # zero direction components: inside the slab -> hit, outside -> miss (:1285-1287)
    assert ob.intersect_ray_aabb((0.5, 0.5, -5), (0, 0, 1), (0, 0, 0), (1, 1, 1))[0]
    assert not ob.intersect_ray_aabb((1.5, 0.5, -5), (0, 0, 1), (0, 0, 0), (1, 1, 1))[0]
    assert ob.intersect_ray_aabb((1.0, 0.5, -5), (0, 0, 1), (0, 0, 0), (1, 1, 1))[0]    # on the face: inclusive
    # flat box (extent 0), as the Cornell floor mesh has
    assert ob.intersect_ray_aabb((0.2, 1, 0.3), (0, -1, 0), (0, 0, 0), (1, 0, 1))[0]


def test_point_in_box_inclusive():
    L = ob.lib()
    assert L.pto_point_in_aabb(ob.f3(1, 1, 1), ob.f3(0, 0, 0), ob.f3(1, 1, 1))
    assert not L.pto_point_in_aabb(ob.f3(1.0000001, 0, 0), ob.f3(0, 0, 0), ob.f3(1, 1, 1))


def test_hemisphere_and_frame():
    out = ob.f3(0, 0, 0)
    ob.lib().pto_uniform_sample_hemisphere(1.0, 0.3, out)
    assert out[1] == 1.0 and abs(out[0]) < 1e-6 and abs(out[2]) < 1e-6     # r1 = cos(theta) = 1 -> the normal
    ob.lib().pto_uniform_sample_hemisphere(0.0, 0.0, out)
    assert out[:] == [1.0, 0.0, 0.0]
    rng = np.random.default_rng(3)
    for _ in range(50):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        nt, nb = ob.f3(0, 0, 0), ob.f3(0, 0, 0)
        ob.lib().pto_create_coordinate_system(ob.f3(*n), nt, nb)
        nt, nb = np.array(nt[:]), np.array(nb[:])
        assert abs(np.dot(nt, n)) < 1e-6 and abs(np.dot(nb, n)) < 1e-6 and abs(np.dot(nt, nb)) < 1e-6
        assert abs(np.linalg.norm(nt) - 1) < 1e-6
        assert np.allclose(np.cross(n, nt), nb, atol=1e-6)


def test_rng_range_and_uniformity():
    xs = np.array([ob.rng_u01(0x5EED, p, s, d, k) for p in range(40) for s in range(10) for d in range(4) for k in range(2)])
    assert xs.min() >= 0.0 and xs.max() < 1.0
    assert abs(xs.mean() - 0.5) < 0.02 and abs(xs.var() - 1 / 12) < 0.01
    assert ob.rng_u01(1, 2, 3, 0, 0) != ob.rng_u01(1, 2, 3, 0, 1) != ob.rng_u01(1, 2, 3, 1, 0)
    assert ob.rng_u01(1, 2, 3, 0, 0) == ob.rng_u01(1, 2, 3, 0, 0)


def test_primary_ray_of_cornell_preset():
    # CPUPathTracer.cpp:489-497 with the Cornell preset (RayTracerView.cpp:231-234): eye (0,1.5,4), looking down -z
    cam, sun = scenes.cornell_camera(div=4)
    o, d = ob.primary_ray(cam, 240, 135)        # image centre: vx = vy = 0
    assert np.allclose(o, [0, 1.5, 4]) and np.allclose(d, [0, 0, -1], atol=1e-5)
    o, d = ob.primary_ray(cam, 0, 0)            # top-left pixel: up, and towards +x because the camera is turned by pi
    assert d[1] > 0 and d[0] > 0 and d[2] < 0
    assert abs(np.linalg.norm(d) - 1) < 1e-6
    assert np.allclose(sun, [0, 20 * math.sin(math.pi * 0.45), 20 * math.cos(math.pi * 0.45)], atol=1e-4)


def test_cornell_analytic_hits(cornell, cornell_oracle):
    # centre pixel: back wall z = -1.04 at distance 5.04 from the eye at z = 4
    cam, _ = scenes.cornell_camera(div=4)
    o, d = ob.primary_ray(cam, 240, 135)
    hit, h = cornell_oracle.trace(o, d)
    assert hit and abs(h.t - 5.04) < 1e-4 and h.tri in (30, 31)
    assert abs(h.hitpoint[2] + 1.04) < 1e-5 and np.allclose(h.normal[:], [0, 0, 1], atol=1e-6)
    # straight down from above the floor centre-right: floor y = 0
    hit, h = cornell_oracle.trace((0.8, 1.0, 0.8), (0, -1, 0))
    assert hit and abs(h.t - 1.0) < 1e-6 and h.tri in (24, 25) and np.allclose(h.normal[:], [0, 1, 0])
    # from the box interior towards +x: right wall x = 1 (green, mesh "rightWall")
    hit, h = cornell_oracle.trace((0.9, 1.0, 0.9), (1, 0, 0))
    assert hit and abs(h.t - 0.1) < 1e-6 and h.tri in (26, 27)
    # upwards: the data set has no ceiling -> miss
    assert not cornell_oracle.trace((0, 1.0, 0), (0, 1, 0))[0]
    # duplicated faces of the file (tallBox lists two faces twice): the lower id wins
    hit, h = cornell_oracle.trace((-0.245, 0.5, 2.0), (0, 0, -1))
    assert hit and h.tri in (0, 1)          # not 10/11, their exact duplicates


def test_cornell_scene_assembly(cornell):
    # MainFrm.cpp:315-492: one mesh per usemtl group, sorted by triangle count descending (stable)
    assert cornell.num_triangles == 32
    assert [m[1] for m in cornell.meshes()] == [12, 12, 2, 2, 2, 2]
    d = cornell.desc.contents
    names_by_count = [d.meshes[i].material for i in range(d.num_meshes)]
    # material indices follow the .mtl order: backWall 0, ceiling 1, floor 2, leftWall 3, rightWall 4, shortBox 5, tallBox 6
    assert names_by_count == [6, 5, 2, 4, 3, 0]
    # diffuse is stored BGR (MainFrm.cpp:247): leftWall Kd = 0.63 0.065 0.05
    assert np.allclose(d.materials[3].diffuse[:], [0.05, 0.065, 0.63])
    # winding flipped (MainFrm.cpp:380-383): first face "f 57//2 58//2 60//2" -> corners 57, 60, 58
    v = cornell.vertices()
    assert np.allclose(v[0, 0, :3], [0.04, 0, -0.09]) and np.allclose(v[0, 1, :3], [-0.53, 0, 0.09])
    assert np.allclose(v[0, 2, :3], [0.04, 1.2, -0.09])


def _on_an_edge(tuv, tol=1e-4):
    u, v = tuv[:, 1], tuv[:, 2]
    return (np.abs(u) < tol) | (np.abs(v) < tol) | (np.abs(u + v - 1) < tol)


def test_octree_matches_brute_force_on_cornell(cornell, cornell_oracle):
    # no mesh of the Cornell box reaches 200 triangles, so the reference's search is the per-mesh root box + list
    assert cornell_oracle.octree_info()[1] == 0
    rays = scenes.random_rays(cornell, 20000, seed=11, aimed=False)
    t0, tuv0, a0 = cornell_oracle.trace_rays(rays, accel=ob.ACCEL_OCTREE)
    t1, tuv1, a1 = cornell_oracle.trace_rays(rays, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(a0, a1) and np.array_equal(t0, t1) and np.array_equal(tuv0, tuv1)
    # rays aimed exactly at vertices and edge midpoints: the reference's box test (xnacollision.cpp:1228-1297) is
    # not conservative with respect to its triangle test, so the two searches may differ there - and only there
    rays = scenes.random_rays(cornell, 20000, seed=12, aimed=True)
    t0, tuv0, a0 = cornell_oracle.trace_rays(rays, accel=ob.ACCEL_OCTREE)
    t1, tuv1, a1 = cornell_oracle.trace_rays(rays, accel=ob.ACCEL_BRUTE)
    diff = (t0 != t1) | (a0 != a1)
    assert diff.mean() < 0.05
    grazing = _on_an_edge(tuv1) | _on_an_edge(tuv0)
    assert grazing[diff].all()


def test_octree_structure_and_agreement_on_atrium(atrium_small, atrium_small_oracle):
    nodes, depth, leaf_tris = atrium_small_oracle.octree_info()
    assert depth >= 1 and nodes > atrium_small.num_meshes        # meshes with >= 200 triangles are subdivided (Mesh.cpp:161)
    assert leaf_tris >= atrium_small.num_triangles * 0.9          # triangles are duplicated into every child holding a vertex
    rays = scenes.random_rays(atrium_small, 6000, seed=5)
    t0, tuv0, a0 = atrium_small_oracle.trace_rays(rays, accel=ob.ACCEL_OCTREE)
    t1, tuv1, a1 = atrium_small_oracle.trace_rays(rays, accel=ob.ACCEL_BRUTE)
    # the octree's box tests may reject grazing rays that the bare triangle test accepts, and ties
    # resolve by traversal order: allow a handful of differences, require the bulk identical
    same = (t0 == t1)
    assert same.mean() > 0.95, f"{(~same).sum()} of {len(same)} rays differ"
    assert np.array_equal(tuv0[same], tuv1[same])
    assert (_on_an_edge(tuv1) | _on_an_edge(tuv0))[~same].all()
    generic = scenes.random_rays(atrium_small, 6000, seed=6, aimed=False)
    g0 = atrium_small_oracle.trace_rays(generic, accel=ob.ACCEL_OCTREE)
    g1 = atrium_small_oracle.trace_rays(generic, accel=ob.ACCEL_BRUTE)
    assert (g0[0] == g1[0]).mean() > 0.999 and (g0[2] == g1[2]).mean() > 0.999


def test_octree_drops_triangles_that_lie_on_a_refitted_box_face():
    """The reference's octree loses triangles (it is the reference's behaviour, restated faithfully).
    CMesh::ComputeSubmeshes gives a triangle to an octant if one of its corners passes the point test
    |p - centre| <= extent (XNA::IntersectPointAxisAlignedBox, xnacollision.cpp:964-972, called from
    AddTriangle, Mesh.cpp:168-186) with the octant's centre = parent centre +- extent/2 and extent =
    parent extent/2 in fp32 (Mesh.cpp:95-132).  Once the parent box has been re-fitted to its triangles
    (RecomputeBB, Mesh.cpp:153-156) its faces are no round numbers any more, and for a corner ON such a
    face the difference p - centre can round to one ulp more than the octant's extent: a triangle with
    all three corners in that face is then in no octant.  The search only tests leaf triangles
    (CMesh::Intersect, CPUPathTracer.cpp:195-249), so it can never hit such a triangle."""
    f = np.float32
    # one box of the 66 k-triangle stand-in (mesh 6, depth 1): re-fitted to z in [-4.0833335, -3.4166667]
    lo, hi = f(-4.0833335), f(-3.4166667)
    c, e = f((lo + hi) / f(2)), f((hi - lo) / f(2))            # set_box_from_minmax / RecomputeBB
    assert (c, e) == (f(-3.75), f(0.33333337))
    oc, oe = f(c - f(e / f(2))), f(e / f(2))                   # the lower octants, Mesh.cpp:95-132
    assert f(oc - oe) == lo                                    # their lower face IS the box face, and yet
    assert abs(f(lo - oc)) == f(0.16666675) and oe == f(0.16666669) and not abs(f(lo - oc)) <= oe
    assert not ob.lib().pto_point_in_aabb(ob.f3(0, 0, lo), ob.f3(0, 0, oc), ob.f3(1, 1, oe))
    scene = host.HostScene.atrium(abi.SCENE_SPONZA, 66000, 0, 1)
    osc = ob.OracleScene(scene.desc)
    assert osc.octree_reachable() == (66012, 61852)
    assert osc.octree_reachable(6) == (3564, 1278)            # the worst mesh: 64 % of its triangles are in no leaf
    # a ray aimed at the middle of a lost triangle: the brute-force search finds it, the octree walk cannot
    mask = osc.octree_reachable_mask()
    lost = np.flatnonzero(mask == 0)
    first, count, _, _ = scene.meshes()[6]
    lost6 = lost[(lost >= first) & (lost < first + count)]
    assert len(lost6) == 3564 - 1278
    tri = scene.vertices()[lost6[:64], :, :3].astype(np.float64)
    centre = tri.mean(axis=1)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    rays = np.concatenate([centre + 0.05 * nrm, -nrm], axis=1).astype(np.float32)
    tb, _, _ = osc.trace_rays(rays, accel=ob.ACCEL_BRUTE)
    to, _, _ = osc.trace_rays(rays, accel=ob.ACCEL_OCTREE)
    assert np.array_equal(tb, lost6[:64]) and not np.isin(to, lost6).any()
    # small scenes lose nothing: no mesh of the Cornell box is subdivided, and the 10 k stand-in keeps all
    small_scene = host.HostScene.atrium(abi.SCENE_CRYTEK, 10000, 0, 1)
    assert ob.OracleScene(small_scene.desc).octree_reachable() == (10008, 10008)


def test_direct_image_properties(cornell, cornell_oracle):
    cam, sun = scenes.cornell_camera(div=8)
    prm = host.default_params(sun_pos=sun, global_illumination=0, num_samples=8)
    img, dword, tri, st = cornell_oracle.render(cam, prm)
    # with GI off the sample count is inert: one primary and at most one shadow ray per pixel
    assert st.closest_rays == 240 * 135 and st.shadow_rays == int((tri >= 0).sum())
    prm2 = host.default_params(sun_pos=sun, global_illumination=0, num_samples=64)
    img2, *_ = cornell_oracle.render(cam, prm2)
    assert np.array_equal(img, img2)
    # radiance is clamped to 1 per level (CPUPathTracer.cpp:442-444) -> at most 255
    assert img.max() <= 255.0 and img.min() >= 0.0
    assert (img[tri < 0] == 0).all()
    # BGR packing of CPUPathTracer.cpp:509
    b, g, r = img[..., 0].astype(np.uint32), img[..., 1].astype(np.uint32), img[..., 2].astype(np.uint32)
    assert np.array_equal(dword, b | (g << 8) | (r << 16) | (255 << 24))
    assert np.array_equal(host.pack_bgra(img).reshape(135, 240), dword)


def test_gi_ray_budget_and_split_estimator(cornell, cornell_oracle):
    cam, sun = scenes.cornell_camera(div=16)
    N = 5
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N)
    img, _, tri, st = cornell_oracle.render(cam, prm)
    pixels = tri.size
    assert st.closest_rays + st.shadow_rays <= pixels * (2 + 6 * N)        # at most 2 + 6N rays per pixel
    acc = cornell_oracle.accumulate(cam, prm)
    assert np.array_equal(cornell_oracle.finalize(cam, prm, acc), img)
    # tile shards add up exactly; sample-range shards add up to rounding
    parts = []
    for r in range(3):
        p = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N, shard_rank=r, shard_count=3, tile_size=8)
        parts.append(cornell_oracle.accumulate(cam, p))
    assert np.array_equal(cornell_oracle.finalize(cam, prm, sum(parts)), img)
    pa = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N, sample_begin=0, sample_end=2)
    pb = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N, sample_begin=2, sample_end=N)
    both = cornell_oracle.accumulate(cam, pa) + cornell_oracle.accumulate(cam, pb)
    assert np.allclose(cornell_oracle.finalize(cam, prm, both), img, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("name", ["cornell_direct_480x270"])
def test_oracle_reproduces_golden(name, cornell, cornell_oracle):
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cam, sun = scenes.cornell_camera(div=4)
    prm = host.default_params(sun_pos=sun, global_illumination=0, num_samples=8)
    img, dword, tri, _ = cornell_oracle.render(cam, prm, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(tri, g["tri"])
    assert np.array_equal(dword, g["dword"])
    assert np.array_equal(img, g["bgr255"])

"""Whole-render parity of the HIP wavefront pipeline against the CPU oracle, through the C ABI.

 - primary-hit triangle ids: identical (ties -> lowest id)
 - direct-light radiance: per-pixel relative error <= 1e-3 (in practice identical)
 - GI: same counter-based random stream on both sides; image mean within 3 sigma of the Monte
   Carlo error, RMSE reported and bounded
"""
import os

import numpy as np
import pytest

import oracle_binding as ob
import scenes
from conftest import GOLDEN
from mipt import abi, host

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3      # direct-light tolerance of the parity contract


def rel_err(a, b):
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)      # images are scaled to 0..255


def test_cornell_direct_golden_480x270(gpu, cornell):
    g = np.load(os.path.join(GOLDEN, "cornell_direct_480x270.npz"))
    cam, sun = scenes.cornell_camera(div=4)
    prm = host.default_params(sun_pos=sun, global_illumination=0, num_samples=8)
    sc = gpu.Scene(cornell.desc)
    img, ids, st = sc.render(cam, prm)
    assert np.array_equal(ids, g["tri"])
    assert rel_err(img, g["bgr255"]).max() <= REL_TOL
    assert np.array_equal(img, g["bgr255"])                 # the arithmetic is the same fp32 sequence
    assert np.array_equal(host.pack_bgra(img).reshape(270, 480), g["dword"])
    assert st.closest_rays == 480 * 270 and st.shadow_rays + st.shadow_rays_culled == int((ids >= 0).sum())


def test_cornell_direct_full_resolution(gpu, cornell, cornell_oracle):
    cam, sun = scenes.cornell_camera(div=1)
    prm = host.default_params(sun_pos=sun, global_illumination=0, num_samples=64)
    sc = gpu.Scene(cornell.desc)
    img, ids, st = sc.render(cam, prm)
    oimg, _, otri, _ = cornell_oracle.render(cam, prm, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(ids, otri)
    assert rel_err(img, oimg).max() <= REL_TOL


@pytest.mark.parametrize("use_textures,use_normals,nnm", [(1, 0, 0), (1, 1, 0), (0, 1, 1), (1, 1, 1)])
def test_atrium_direct_textures_and_normal_maps(gpu, atrium_small, atrium_small_oracle, use_textures, use_normals, nnm):
    cam, sun = scenes.crytek_camera(div=4)
    prm = host.default_params(sun_pos=sun, global_illumination=0, use_textures=use_textures, use_normals=use_normals,
                              normalize_normal_map=nnm)
    sc = gpu.Scene(atrium_small.desc)
    img, ids, _ = sc.render(cam, prm)
    oimg, _, otri, _ = atrium_small_oracle.render(cam, prm, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(ids, otri)
    assert (ids >= 0).mean() > 0.9
    assert rel_err(img, oimg).max() <= REL_TOL
    assert img.max() > 10          # something is lit


@pytest.mark.parametrize("use_normals,nnm", [(0, 0), (1, 0), (1, 1)])
def test_float3_texels_as_the_reference_uploads_them(gpu, atrium_small, atrium_small_oracle, use_normals, nnm):
    """The texel layout CCUDAPathTracer::CalcRays really receives: three floats per texel
    (RayTracerView.cpp:819-842).  Byte-valued float textures are repacked to four bytes per texel by
    the library and must give the image of the DWORD textures bit for bit; textures with other values
    (two materials here hold x.5 everywhere) stay float and are used as PathTracer.cu:311-313,344-346
    uses them, value / 255, which the oracle restates for float texels."""
    cam, sun = scenes.crytek_camera(div=4)
    prm = host.default_params(sun_pos=sun, global_illumination=0, use_textures=1, use_normals=use_normals,
                              normalize_normal_map=nnm)
    # byte values only: identical to the DWORD-texel render
    fdesc, keep = scenes.with_float_textures(atrium_small)
    img, ids, _ = gpu.Scene(fdesc).render(cam, prm)
    oimg, _, otri, _ = atrium_small_oracle.render(cam, prm, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(ids, otri) and np.array_equal(img, oimg)
    # non-byte values in the two largest materials (meshes are ordered by triangle count)
    shifted = tuple(sorted({m for _, _, m, _ in atrium_small.meshes()[:2] if m >= 0}))
    assert shifted
    fdesc2, keep2 = scenes.with_float_textures(atrium_small, offset_materials=shifted)
    osc2 = ob.OracleScene(fdesc2)
    img2, ids2, _ = gpu.Scene(fdesc2).render(cam, prm)
    oimg2, _, otri2, _ = osc2.render(cam, prm, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(ids2, otri2)
    assert np.array_equal(img2.view(np.uint32), oimg2.view(np.uint32))
    assert not np.array_equal(img2, img)            # the half levels are visible in the image


def gi_compare(img, oimg, n_samples):
    diff = img.astype(np.float64) - oimg.astype(np.float64)
    rmse = float(np.sqrt((diff ** 2).mean()))
    # Monte Carlo standard error of the image mean, estimated from the oracle image itself: per-pixel
    # values are independent estimates; the mean of P pixels has sigma = std / sqrt(P)
    sigma = float(oimg.std() / np.sqrt(oimg.size))
    return rmse, abs(float(img.mean() - oimg.mean())), sigma, float((img == oimg).mean())


def test_cornell_gi_against_oracle(gpu, cornell, cornell_oracle):
    g = np.load(os.path.join(GOLDEN, "cornell_gi16_240x135.npz"))
    cam, sun = scenes.cornell_camera(div=8)
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=16)
    sc = gpu.Scene(cornell.desc)
    img, ids, st = sc.render(cam, prm)
    assert np.array_equal(ids, g["tri"])
    rmse, dmean, sigma, same = gi_compare(img, g["bgr255"], 16)
    print(f"cornell GI N=16: rmse={rmse:.4g} |dmean|={dmean:.4g} sigma={sigma:.4g} identical={same:.4f}")
    assert dmean <= 3 * sigma
    assert rmse < 1.0              # same random stream: only sin/cos rounding separates the two sides
    assert st.closest_rays + st.shadow_rays + st.shadow_rays_culled <= ids.size * (2 + 6 * 16)
    assert st.closest_rays > ids.size


def test_atrium_gi_against_oracle(gpu, atrium_small, atrium_small_oracle):
    cam, sun = scenes.crytek_camera(div=8)
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=8, use_normals=1)
    sc = gpu.Scene(atrium_small.desc)
    img, ids, st = sc.render(cam, prm)
    oimg, _, otri, ost = atrium_small_oracle.render(cam, prm, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(ids, otri)
    rmse, dmean, sigma, same = gi_compare(img, oimg, 8)
    print(f"atrium GI N=8: rmse={rmse:.4g} |dmean|={dmean:.4g} sigma={sigma:.4g} identical={same:.4f}")
    assert dmean <= 3 * sigma
    assert rmse < 2.0
    # ray counts agree up to the few paths that bifurcate
    assert abs(int(st.closest_rays) - int(ost.closest_rays)) <= 0.001 * ost.closest_rays


def test_shadow_rays_that_cannot_matter_are_counted_not_traced(gpu, atrium_small, atrium_small_oracle):
    """The reference casts a shadow ray at every hit (CPUPathTracer.cpp:402); where the direct term is +0
    the library skips the traversal.  traced + culled must equal the oracle's shadow-ray count, and with a
    dark sun (intensity 0: every direct term is +0) no shadow ray is traced at all, every block of the
    shadow queue is empty, and the image still equals the oracle's."""
    cam, sun = scenes.crytek_camera(div=8)
    sc = gpu.Scene(atrium_small.desc)
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=4)
    img, ids, st = sc.render(cam, prm)
    oimg, _, otri, ost = atrium_small_oracle.render(cam, prm, accel=ob.ACCEL_BRUTE)
    assert np.array_equal(ids, otri)
    assert st.shadow_rays_culled > 0 and st.shadow_rays > 0
    assert abs(int(st.shadow_rays + st.shadow_rays_culled) - int(ost.shadow_rays)) <= 0.001 * ost.shadow_rays
    dark = host.default_params(sun_pos=sun, global_illumination=1, num_samples=4, sun_intensity=0.0)
    img, ids, st = sc.render(cam, dark)
    oimg, _, otri, ost = atrium_small_oracle.render(cam, dark, accel=ob.ACCEL_BRUTE)
    assert st.shadow_rays == 0 and st.shadow_rays_culled > 0
    assert np.array_equal(ids, otri)
    assert not img.any() and not oimg.any()         # no light, no colour: L_k = (0 + 0 * ...) * A / pi


def test_pass_size_does_not_change_the_image(gpu, cornell):
    cam, sun = scenes.cornell_camera(div=8)
    sc = gpu.Scene(cornell.desc)
    ref = None
    for chunk in (0, 1, 3, 16):
        prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=16, chunk_samples=chunk)
        img, _, st = sc.render(cam, prm)
        if ref is None:
            ref = img
        assert np.array_equal(img, ref), f"chunk_samples={chunk}"


def test_tile_and_sample_shards_add_up(gpu, cornell):
    """What bench.py does across GPUs, run sequentially on one: every shard accumulates into its own
    buffer, the buffers are summed (the RCCL reduce), rank 0 finalizes."""
    cam, sun = scenes.cornell_camera(div=8)
    sc = gpu.Scene(cornell.desc)
    N = 12
    full = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N)
    ref, _, _ = sc.render(cam, full)
    npx = 240 * 135

    def run(params_list):
        total = np.zeros(npx * 3, dtype=np.float32)
        frames = []
        for p in params_list:
            fr = gpu.Frame(sc, cam, p)
            acc = gpu.DeviceBuffer(npx * 12)
            fr.accumulate(acc.ptr)
            total += acc.to_host()
            frames.append(fr)
        acc = gpu.DeviceBuffer(npx * 12)
        out = gpu.DeviceBuffer(npx * 12)
        acc.from_host(total)
        frames[0].finalize(acc.ptr, out.ptr)
        return out.to_host().reshape(135, 240, 3)

    tiles = [host.default_params(sun_pos=sun, global_illumination=1, num_samples=N, shard_rank=r, shard_count=4, tile_size=16)
             for r in range(4)]
    assert np.array_equal(run(tiles), ref)                   # disjoint pixels: x + 0 is exact
    halves = [host.default_params(sun_pos=sun, global_illumination=1, num_samples=N, sample_begin=a, sample_end=b)
              for a, b in ((0, 5), (5, N))]
    assert np.allclose(run(halves), ref, rtol=1e-5, atol=1e-3)   # partial sums: equal up to fp32 summation order


def test_progress_callback_contract(gpu, cornell):
    cam, sun = scenes.cornell_camera(div=8)
    sc = gpu.Scene(cornell.desc)
    calls = []
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=8, chunk_samples=2)
    sc.render(cam, prm, progress=lambda done, total, user: calls.append((done, total)))
    total = 240 * 135
    assert calls[0] == (0, total) and calls[-1] == (total, total)           # PathTracer.cu:721, 726-731
    # four passes of two samples: a call after every depth of every pass (the last depth coincides with the end of the pass)
    assert all(b[0] >= a[0] for a, b in zip(calls, calls[1:])) and len(calls) == 1 + 4 * 3
    assert len(set(calls)) == len(calls), "every call reports more work than the one before"


def test_invalid_arguments_are_rejected(gpu, cornell):
    cam, sun = scenes.cornell_camera(div=8)
    sc = gpu.Scene(cornell.desc)
    bad = host.default_params(sun_pos=sun, shard_rank=3, shard_count=2)
    with pytest.raises(gpu.MiptError):
        sc.render(cam, bad)
    bad = host.default_params(sun_pos=sun, num_samples=4, sample_begin=3, sample_end=9)
    with pytest.raises(gpu.MiptError):
        sc.render(cam, bad)
    cam0 = abi.Camera.from_buffer_copy(bytes(cam)); cam0.div = 0
    with pytest.raises(gpu.MiptError):
        sc.render(cam0, host.default_params(sun_pos=sun))


def test_cancel_and_partial_image(gpu, cornell):
    """Cancel from the progress call-back: the call stops after the pass it is in, the partial sums
    equal those of a render with that many samples, and every call-back sees a usable partial image."""
    cam, sun = scenes.cornell_camera(div=8)
    sc = gpu.Scene(cornell.desc)
    npx = 240 * 135
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=8, chunk_samples=2)
    fr = gpu.Frame(sc, cam, prm)
    acc, out = gpu.DeviceBuffer(npx * 12), gpu.DeviceBuffer(npx * 12)
    previews = []

    def progress(done, total, user):
        # call-backs come after every depth of a pass; the sums (and samples_done) change when a pass ends
        if done > 0 and (not previews or previews[-1][0] != fr.samples_done):
            fr.finalize_partial(acc.ptr, fr.samples_done, out.ptr)
            previews.append((fr.samples_done, out.to_host().reshape(135, 240, 3).copy()))
        if fr.samples_done == 4:
            fr.cancel()

    with pytest.raises(gpu.MiptCancelled) as ei:
        fr.accumulate(acc.ptr, progress=progress)
    assert ei.value.stats.samples_done == 4 and fr.samples_done == 4
    assert [n for n, _ in previews] == [0, 2, 4]           # 0: a call-back inside the first pass sees the direct-light image
    for n, img in previews:
        if n == 0:
            ref, _, _ = sc.render(cam, host.default_params(sun_pos=sun, global_illumination=0, num_samples=1))
            assert np.array_equal(img, ref), "partial image before the first pass ends"
            continue
        ref, _, _ = sc.render(cam, host.default_params(sun_pos=sun, global_illumination=1, num_samples=n))
        assert np.array_equal(img, ref), f"partial image after {n} samples"
    # the flag is consumed: the next call runs to the end
    st = fr.accumulate(acc.ptr)
    assert st.samples_done == 8
    fr.finalize(acc.ptr, out.ptr)
    ref, _, _ = sc.render(cam, prm)
    assert np.array_equal(out.to_host().reshape(135, 240, 3), ref)


def test_size_limits_are_rejected_before_any_work(gpu, cornell):
    """Sizes beyond what the library indexes with 31 bits come back as errors, not as faults."""
    import ctypes as C
    sc = gpu.Scene(cornell.desc)
    # 2^31 rays: refused on the count alone (the array is never touched)
    rays = np.zeros((1, 6), dtype=np.float32)
    rc = gpu.lib().mipt_trace_rays(sc._h, rays.ctypes.data, (1 << 31) - 64, 1e-20, None, None, None)
    assert rc == -4 and b"too many rays" in gpu.lib().mipt_last_error()
    # 65536 x 32768 pixels
    cam, sun = scenes.cornell_camera(div=1, width=65536, height=32768)
    with pytest.raises(gpu.MiptError, match="too large"):
        gpu.Frame(sc, cam, host.default_params(sun_pos=sun))
    # more samples than the reference's dialog accepts are clamped to 10000 (RayTracerView.cpp:785-786), not an error
    cam, sun = scenes.cornell_camera(div=120)
    fr = gpu.Frame(sc, cam, host.default_params(sun_pos=sun, global_illumination=0, num_samples=1 << 30))
    assert fr.num_pixels == 16 * 9


def test_headline_frame_at_full_size(gpu):
    """The bench workload at its real size (262 k triangles, 1920x1080, 128 samples, GI), checked through
    properties that do not need the whole oracle image: the pass size and the tile shards do not change
    a single pixel, every pixel is a colour, and one image row of ids and direct light equals the oracle's
    brute-force search bit for bit."""
    big = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 128, 1)
    cam, sun = scenes.crytek_camera(div=1)
    sc = gpu.Scene(big.desc)
    npx = 1920 * 1080
    full = host.default_params(sun_pos=sun, global_illumination=1, num_samples=128)
    ref, ids, st = sc.render(cam, full)
    assert st.closest_rays + st.shadow_rays + st.shadow_rays_culled > 1.2e9 and st.samples_done == 128
    assert np.isfinite(ref).all() and ref.min() >= 0 and ref.max() <= 255
    assert (ids >= 0).mean() > 0.9
    # smaller passes: same image
    img, _, _ = sc.render(cam, host.default_params(sun_pos=sun, global_illumination=1, num_samples=128, chunk_samples=24))
    assert np.array_equal(img, ref)
    # eight tile shards, summed as the multi-GPU run sums them
    total = np.zeros(npx * 3, dtype=np.float32)
    first = None
    for r in range(8):
        fr = gpu.Frame(sc, cam, host.default_params(sun_pos=sun, global_illumination=1, num_samples=128, shard_rank=r, shard_count=8))
        acc = gpu.DeviceBuffer(npx * 12)
        fr.accumulate(acc.ptr)
        total += acc.to_host()
        if first is None:
            first = fr
    acc, out = gpu.DeviceBuffer(npx * 12), gpu.DeviceBuffer(npx * 12)
    acc.from_host(total)
    first.finalize(acc.ptr, out.ptr)
    assert np.array_equal(out.to_host().reshape(1080, 1920, 3), ref)
    # one row against the oracle's brute-force search (direct light: no random numbers involved)
    osc = ob.OracleScene(big.desc)
    direct = host.default_params(sun_pos=sun, global_illumination=0)
    dimg, dids, _ = sc.render(cam, direct)
    oimg, _, otri, _ = osc.render(cam, direct, accel=ob.ACCEL_BRUTE, row_step=1080, row_phase=540)
    assert np.array_equal(dids[540], otri[540]) and np.array_equal(ids[540], otri[540])
    assert np.array_equal(dimg[540].view(np.uint32), oimg[540].view(np.uint32))


def test_tile_shards_that_finalize_their_own_pixels(gpu, cornell):
    """own_pixels_only: every shard traces depth 0 for its tiles only and finalizes them; the final
    images (black outside the shard) add up to the frame, bit for bit."""
    cam, sun = scenes.cornell_camera(div=8)
    sc = gpu.Scene(cornell.desc)
    N, npx = 12, 240 * 135
    ref, ref_ids, st_full = sc.render(cam, host.default_params(sun_pos=sun, global_illumination=1, num_samples=N))
    total = np.zeros((135, 240, 3), dtype=np.float32)
    ids = np.full((135, 240), -1, dtype=np.int32)
    rays = 0
    for r in range(4):
        prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N, shard_rank=r, shard_count=4, tile_size=16, own_pixels_only=1)
        img, sid, st = sc.render(cam, prm)
        assert ((img != 0).any(axis=2) <= (sid >= 0)).all()          # colour only where this shard traced
        assert not ((total != 0).any(axis=2) & (img != 0).any(axis=2)).any()
        total += img
        ids = np.maximum(ids, sid)
        rays += st.closest_rays + st.shadow_rays + st.shadow_rays_culled
    assert np.array_equal(total, ref) and np.array_equal(ids, ref_ids)
    # no duplicated work, and the stand-in rays of pixels of other shards are not counted
    assert rays == st_full.closest_rays + st_full.shadow_rays + st_full.shadow_rays_culled


def test_box_tests_are_counted_on_request(gpu, atrium_small):
    # mipt_params.count_box_tests: the closest-hit kernel also counts the child boxes that exist in the nodes it visits
    cam, sun = scenes.crytek_camera(div=8)
    sc = gpu.Scene(atrium_small.desc)
    img0, ids0, st0 = sc.render(cam, host.default_params(sun_pos=sun, global_illumination=1, num_samples=2))
    img1, ids1, st1 = sc.render(cam, host.default_params(sun_pos=sun, global_illumination=1, num_samples=2, count_box_tests=1))
    assert st0.closest_box_tests == 0
    assert st1.closest_node_visits == st0.closest_node_visits > 0
    assert 2 * st1.closest_node_visits <= st1.closest_box_tests <= sc.info().node_width * st1.closest_node_visits
    assert np.array_equal(img0, img1) and np.array_equal(ids0, ids1)

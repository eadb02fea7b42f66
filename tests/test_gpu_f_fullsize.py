"""Parity at the full sizes of the benchmark configurations: the whole frame is rendered on the GPU
at 1920x1080 with the real sample count (128 first-bounce samples, in as many passes as the
library chooses by itself), and a window of one image row is rendered by the CPU oracle with its
brute-force search over every triangle (CMesh::Intersect / Radiance, CPUPathTracer.cpp:195-249,
348-446, restated in oracle/pt_oracle.c).  Both sides draw from the same counter-based random
stream, so they agree except where the last bit of sin/cos sends a path to the other side of a
triangle edge.

 - primary-hit ids in the window: identical
 - GI values: mean within 3 sigma of the Monte Carlo error, RMSE printed, >= 90 % identical
 - direct light (no sampling involved): identical bit for bit over a whole row
 - the 10-million-triangle scene of configuration 5 at 3840x2160: a direct-light window bit for bit, and a
   short GI window (8 samples; 512 would keep the brute-force search busy for an hour)
"""
import numpy as np
import pytest

import oracle_binding as ob
from mipt import abi, host

pytestmark = pytest.mark.gpu

WIDTH, HEIGHT = 1920, 1080


def window_compare(img, oimg):
    diff = img.astype(np.float64) - oimg.astype(np.float64)
    rmse = float(np.sqrt((diff ** 2).mean()))
    sigma = float(oimg.std() / np.sqrt(oimg.size))          # standard error of the window's mean, from the oracle's values
    return rmse, abs(float(img.mean() - oimg.mean())), sigma, float((img == oimg).mean())


@pytest.mark.parametrize("kind,tris,row,col0", [
    (abi.SCENE_CRYTEK, 262000, 600, 900),      # configuration 4: the headline frame
    (abi.SCENE_SPONZA, 66000, 700, 400),       # configuration 3
])
def test_gi_128_samples_1080p_window_against_brute_force(gpu, kind, tris, row, col0):
    scene = host.HostScene.atrium(kind, tris, 256, 1)
    cam, sun = host.camera_preset(kind, WIDTH, HEIGHT, 1)
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=128, use_textures=1, use_normals=0)
    sc = gpu.Scene(scene.desc)
    img, ids, st = sc.render(cam, prm)
    assert st.chunk_samples < 128, "the automatic pass size is expected to cut 128 samples at 1080p into several passes"
    osc = ob.OracleScene(scene.desc)
    cols = (col0, col0 + 64)
    oimg, _, otri, ost = osc.render(cam, prm, accel=ob.ACCEL_BRUTE, row_step=HEIGHT, row_phase=row, cols=cols)
    w_img, w_oimg = img[row, cols[0]:cols[1]], oimg[row, cols[0]:cols[1]]
    assert np.array_equal(ids[row, cols[0]:cols[1]], otri[row, cols[0]:cols[1]])
    assert (otri[row, cols[0]:cols[1]] >= 0).mean() > 0.9, "the window should look at geometry"
    rmse, dmean, sigma, same = window_compare(w_img, w_oimg)
    print(f"{tris} triangles, 1080p, N=128, row {row} columns {cols}: rmse={rmse:.4g} |dmean|={dmean:.4g} sigma={sigma:.4g} "
          f"identical={same:.4f} (oracle: {ost.closest_rays + ost.shadow_rays} rays, {ost.triangle_tests:.3g} triangle tests, {ost.seconds:.1f} s)")
    assert dmean <= 3 * sigma
    assert same >= 0.90
    assert rmse < 2.0
    sc.close()


def test_direct_light_full_resolution_row_is_bit_exact(gpu):
    """Configuration 2: the 66 k-triangle scene, direct light only, 1920x1080; one whole image row."""
    scene = host.HostScene.atrium(abi.SCENE_SPONZA, 66000, 256, 1)
    cam, sun = host.camera_preset(abi.SCENE_SPONZA, WIDTH, HEIGHT, 1)
    prm = host.default_params(sun_pos=sun, global_illumination=0, num_samples=64, use_textures=1, use_normals=0)
    sc = gpu.Scene(scene.desc)
    img, ids, st = sc.render(cam, prm)
    osc = ob.OracleScene(scene.desc)
    row = 540
    oimg, odw, otri, _ = osc.render(cam, prm, accel=ob.ACCEL_BRUTE, row_step=HEIGHT, row_phase=row)
    assert np.array_equal(ids[row], otri[row])
    assert np.array_equal(img[row].view(np.uint32), oimg[row].view(np.uint32))
    assert np.array_equal(host.pack_bgra(img)[row * WIDTH:(row + 1) * WIDTH].reshape(-1), odw[row])
    assert st.closest_rays == WIDTH * HEIGHT
    sc.close()


def test_ten_million_triangles_at_2160p_windows_against_brute_force(gpu):
    """Configuration 5: 38 shifted copies of the 262 k-triangle scene as one triangle list, 3840x2160."""
    import time
    t0 = time.time()
    scene = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 256, 1).instanced(38, 4000.0)
    W, H = 3840, 2160
    cam, sun = host.camera_preset(abi.SCENE_CRYTEK, W, H, 1)
    sc = gpu.Scene(scene.desc)
    assert sc.info().num_triangles > 9_900_000
    osc = ob.OracleScene(scene.desc)
    t1 = time.time()
    # direct light: 1024 pixels of one row, bit for bit
    prm = host.default_params(sun_pos=sun, global_illumination=0, num_samples=512, use_textures=1, use_normals=0)
    img, ids, st = sc.render(cam, prm)
    row, cols = 1200, (1408, 2432)
    oimg, odw, otri, ost = osc.render(cam, prm, accel=ob.ACCEL_BRUTE, row_step=H, row_phase=row, cols=cols)
    win = slice(cols[0], cols[1])
    assert np.array_equal(ids[row, win], otri[row, win])
    assert (otri[row, win] >= 0).mean() > 0.9
    assert np.array_equal(img[row, win].view(np.uint32), oimg[row, win].view(np.uint32))
    assert st.closest_rays == W * H
    t2 = time.time()
    # global illumination, 8 samples: 32 pixels of the same row
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=8, use_textures=1, use_normals=0)
    img, ids, st = sc.render(cam, prm)
    cols = (1900, 1932)
    oimg, _, otri, ost = osc.render(cam, prm, accel=ob.ACCEL_BRUTE, row_step=H, row_phase=row, cols=cols)
    win = slice(cols[0], cols[1])
    assert np.array_equal(ids[row, win], otri[row, win])
    rmse, dmean, sigma, same = window_compare(img[row, win], oimg[row, win])
    print(f"9.96 M triangles, 2160p: scenes {t1 - t0:.1f} s, direct window {t2 - t1:.1f} s, GI window {time.time() - t2:.1f} s "
          f"({ost.closest_rays + ost.shadow_rays} rays, {ost.triangle_tests:.3g} triangle tests): rmse={rmse:.4g} |dmean|={dmean:.4g} sigma={sigma:.4g} identical={same:.4f}")
    assert same >= 0.85 and dmean <= 3 * sigma + 1e-6
    sc.close()

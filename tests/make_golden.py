"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference holds no golden vectors for this path and cannot be built here, so these
files pin the ORACLE's output (parity unpinned with respect to the reference): a change in
the restatement or in the harness's scene/camera code shows up as a diff against them, and
the GPU path is compared with the same files.

    python tests/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "cuda-path-tracer_amd"))
import oracle_binding as ob  # noqa: E402
import scenes  # noqa: E402
from mipt import abi, host  # noqa: E402

GOLDEN = os.path.join(HERE, "golden")


def main():
    cornell = host.HostScene.load_obj(os.path.join(GOLDEN, "cornell", "CornellBox-Original.mesh"), True)
    osc = ob.OracleScene(cornell.desc)
    # config 1 of the baseline: Cornell box, direct light only, 8 samples (inert), resolution / 4
    cam, sun = scenes.cornell_camera(div=4)
    prm = host.default_params(sun_pos=sun, global_illumination=0, num_samples=8)
    img, dword, tri, _ = osc.render(cam, prm, accel=ob.ACCEL_BRUTE)
    np.savez_compressed(os.path.join(GOLDEN, "cornell_direct_480x270.npz"), tri=tri, dword=dword, bgr255=img)
    # GI, 16 samples, resolution / 8: per-pixel image plus mean / variance summary
    cam8, _ = scenes.cornell_camera(div=8)
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=16)
    img, dword, tri, _ = osc.render(cam8, prm, accel=ob.ACCEL_BRUTE)
    np.savez_compressed(os.path.join(GOLDEN, "cornell_gi16_240x135.npz"), tri=tri, bgr255=img,
                        mean=img.mean(axis=(0, 1)), var=img.var(axis=(0, 1)))
    # seeded ray batch on the Cornell box: closest hit and any-hit answers
    rays = scenes.random_rays(cornell, 4096, seed=2024)
    t, tuv, anyhit = osc.trace_rays(rays, accel=ob.ACCEL_BRUTE)
    np.savez_compressed(os.path.join(GOLDEN, "cornell_rays_4096.npz"), rays=rays, tri=t, tuv=tuv, anyhit=anyhit)
    print("golden fixtures written to", GOLDEN)


if __name__ == "__main__":
    main()

"""How much would ordering the first-bounce rays buy?  Builds the bench scene, takes the
primary hits of the preset camera, draws `spp` uniform hemisphere rays per hit (face normals)
and traces the same ray set in several orders.  Run under
`rocprofv3 --kernel-trace --output-format csv`; the k_trace<0> dispatches appear in the order
printed here, one per ordering.

    rocprofv3 --kernel-trace --output-format csv -d out -o probe -- python3 profiles/coherence_probe.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
from mipt import abi, device, host  # noqa: E402


def main():
    spp = int(os.environ.get("PROBE_SPP", "8"))
    device.init(0)
    scene = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 64, 1)
    cam, _ = host.camera_preset(abi.SCENE_CRYTEK, 1920, 1080, 1)
    dev = device.Scene(scene.desc)
    W, H = cam.image_size
    m = np.array(cam.to_local[:], dtype=np.float32).reshape(4, 4)
    j, h = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    vx = (2.0 * j / W - 1.0) / cam.p0
    vy = (-2.0 * h / H + 1.0) / cam.p1
    d = np.stack([vx, vy, np.ones_like(vx)], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d @ m[:3, :3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.broadcast_to(m[3, :3], d.shape)
    tri, tuv, _ = dev.trace_rays(np.concatenate([o, d], 1), anyhit=False)
    ok = tri >= 0
    P = (o + tuv[:, :1] * d)[ok]
    # face normal of the hit triangle, turned towards the viewer
    v = scene.vertices()[:, :, :3][tri[ok]]   # global triangle id = position in the flattened list (no light meshes here)
    n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    n = np.where((n * d[ok]).sum(1, keepdims=True) > 0, -n, n).astype(np.float32)
    nAct = P.shape[0]
    rng = np.random.default_rng(7)
    rays = np.empty((spp, nAct, 6), dtype=np.float32)
    for s in range(spp):
        r1 = rng.random(nAct, dtype=np.float32)
        phi = 2 * np.pi * rng.random(nAct, dtype=np.float32)
        st = np.sqrt(1 - r1 * r1)
        a = np.where(np.abs(n[:, :1]) > 0.5, [[0, 1, 0]], [[1, 0, 0]]).astype(np.float32)
        t = np.cross(n, a); t /= np.linalg.norm(t, axis=1, keepdims=True)
        b = np.cross(n, t)
        w = st[:, None] * np.cos(phi)[:, None] * b + r1[:, None] * n + st[:, None] * np.sin(phi)[:, None] * t
        rays[s, :, :3] = P + 0.01 * w
        rays[s, :, 3:] = w
    flat = rays.reshape(-1, 6)
    N = flat.shape[0]
    octant = ((flat[:, 3] < 0).astype(np.int64) | ((flat[:, 4] < 0).astype(np.int64) << 1) | ((flat[:, 5] < 0).astype(np.int64) << 2))
    slot = np.arange(N)
    lo, hi = flat[:, :3].min(0), flat[:, :3].max(0)
    cell = np.minimum(((flat[:, :3] - lo) / (hi - lo) * 64).astype(np.int64), 63)

    def morton(c):
        k = np.zeros(c.shape[0], dtype=np.int64)
        for bit in range(6):
            for ax in range(3):
                k |= ((c[:, ax] >> bit) & 1) << (3 * bit + ax)
        return k
    mort = morton(cell)
    # direction quantised to a 8x8 grid per cube face (finer than the octant)
    dq = np.minimum(((flat[:, 3:] * 0.5 + 0.5) * 8).astype(np.int64), 7)
    dkey = dq[:, 0] | (dq[:, 1] << 3) | (dq[:, 2] << 6)
    orders = {
        "sample-major (as rendered)": slot,
        "octant bins, stable": np.argsort(octant, kind="stable"),
        "octant bins within blocks of 1M slots": np.lexsort((octant, slot >> 20)),
        "octant bins within blocks of 64k slots": np.lexsort((octant, slot >> 16)),
        "octant bins within blocks of 1024 slots": np.lexsort((octant, slot >> 10)),
        "dir 8^3 cell within blocks of 1024 slots": np.lexsort((dkey, slot >> 10)),
        "dir 8^3 cell within blocks of 16k slots": np.lexsort((dkey, slot >> 14)),
        "octant, then origin morton": np.lexsort((mort, octant)),
        "dir 8^3 cell, then origin morton": np.lexsort((mort, dkey)),
        "origin morton, then octant": np.lexsort((octant, mort)),
        "random shuffle": rng.permutation(N),
    }
    ref = None
    for name, perm in orders.items():
        t_, _, _ = dev.trace_rays(flat[perm], anyhit=False)
        inv = np.empty(N, dtype=np.int64); inv[perm] = np.arange(N)
        t_ = t_[inv]
        if ref is None:
            ref = t_
        print(f"{name}: rays {N}, identical ids {bool((t_ == ref).all())}", flush=True)


if __name__ == "__main__":
    main()

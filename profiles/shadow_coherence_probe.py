"""How much would ordering the shadow rays of the bounce levels buy?  (variant of coherence_probe.py)
How much would ordering the first-bounce rays buy?  Builds the bench scene, takes the
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
    cam, SUN = host.camera_preset(abi.SCENE_CRYTEK, 1920, 1080, 1)
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
    # hit points of the first-bounce rays: where the shadow rays of level 1 start
    tri1, tuv1, _ = dev.trace_rays(flat, anyhit=False)
    ok1 = tri1 >= 0
    hp = flat[ok1, :3] + tuv1[ok1, :1] * flat[ok1, 3:]
    v1 = scene.vertices()[:, :, :3][tri1[ok1]]
    n1 = np.cross(v1[:, 1] - v1[:, 0], v1[:, 2] - v1[:, 0])
    n1 /= np.maximum(np.linalg.norm(n1, axis=1, keepdims=True), 1e-30)
    n1 = np.where((n1 * flat[ok1, 3:]).sum(1, keepdims=True) > 0, -n1, n1).astype(np.float32)
    sun = np.array(SUN, dtype=np.float32)
    sd = sun - hp
    sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    keep = (n1 * sd).sum(1) > 0                      # the others are not traced at all
    sh = np.concatenate([hp + 0.01 * n1, sd], 1)[keep].astype(np.float32)
    N = sh.shape[0]
    lo, hi = sh[:, :3].min(0), sh[:, :3].max(0)

    def morton(c, bits):
        k = np.zeros(c.shape[0], dtype=np.int64)
        for bit in range(bits):
            for ax in range(3):
                k |= ((c[:, ax] >> bit) & 1) << (3 * bit + ax)
        return k
    orders = {"as generated": np.arange(N)}
    for bits in (3, 4, 6):
        cell = np.minimum(((sh[:, :3] - lo) / (hi - lo) * (1 << bits)).astype(np.int64), (1 << bits) - 1)
        orders[f"origin morton {bits} bits per axis, stable"] = np.argsort(morton(cell, bits), kind="stable")
    orders["random shuffle"] = rng.permutation(N)
    ref = None
    for name, perm in orders.items():
        _, _, ah = dev.trace_rays(sh[perm], closest=False, anyhit=True)
        inv = np.empty(N, dtype=np.int64); inv[perm] = np.arange(N)
        ah = ah[inv]
        if ref is None:
            ref = ah
        print(f"{name}: shadow rays {N}, occluded {float(ah.mean()):.3f}, identical flags {bool((ah == ref).all())}", flush=True)


if __name__ == "__main__":
    main()

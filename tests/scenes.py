"""Shared inputs of the parity tests: seeded rays, cameras, parameter sets."""
import numpy as np

from mipt import abi, host


def cornell_camera(div=4, width=1920, height=1080):
    return host.camera_preset(abi.SCENE_CORNELL, width, height, div)


def crytek_camera(div=4, width=1920, height=1080):
    return host.camera_preset(abi.SCENE_CRYTEK, width, height, div)


def random_rays(scene, n, seed, inside_fraction=0.7, aimed=True):
    """Seeded rays: origins inside the scene's bounds (and some outside), random directions,
    plus rays aimed exactly at vertices and edge midpoints (edge-inclusive ties)."""
    rng = np.random.default_rng(seed)
    v = scene.vertices()[:, :, :3].reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    ext = hi - lo
    o = lo + rng.random((n, 3)) * ext
    outside = rng.random(n) > inside_fraction
    o[outside] = (lo - 0.5 * ext) + rng.random((int(outside.sum()), 3)) * (2 * ext)
    d = rng.normal(size=(n, 3))
    # a third of the rays aim at a vertex or an edge midpoint of a random triangle
    k = n // 3 if aimed else 0
    if k:
        tri = scene.vertices()[rng.integers(0, scene.num_triangles, k), :, :3]
        pick = rng.integers(0, 3, k)
        target = tri[np.arange(k), pick]
        mid = 0.5 * (tri[np.arange(k), pick] + tri[np.arange(k), (pick + 1) % 3])
        use_mid = rng.random(k) < 0.5
        target = np.where(use_mid[:, None], mid, target)
        d[:k] = target - o[:k]
    # some axis-parallel directions (zero components)
    m = max(1, n // 50)
    axis = rng.integers(0, 3, m)
    dd = np.zeros((m, 3)); dd[np.arange(m), axis] = rng.choice([-1.0, 1.0], m)
    d[k:k + m] = dd
    d = d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30)
    return np.concatenate([o, d], axis=1).astype(np.float32)


def scene_from_triangles(tris):
    """mipt_scene_desc over raw triangles (n, 3, 3), one mesh, no material.  Returns (desc, keepalive)."""
    import ctypes as C
    tris = np.asarray(tris, dtype=np.float32).reshape(-1, 3, 3)
    n = tris.shape[0]
    v = np.zeros((n, 3, 8), dtype=np.float32)
    v[:, :, :3] = tris
    v[:, :, 4] = 1.0
    verts = (abi.Vertex * (3 * n)).from_buffer_copy(v.tobytes()) if n else None
    mesh = (abi.Mesh * 1)(abi.Mesh(0, n, -1, 0))
    desc = abi.SceneDesc()
    desc.vertices = C.cast(verts, C.POINTER(abi.Vertex)) if n else None
    desc.num_triangles = n
    desc.meshes = C.cast(mesh, C.POINTER(abi.Mesh))
    desc.num_meshes = 1
    desc.num_materials = 0
    return desc, (verts, mesh)

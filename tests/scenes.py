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


def with_float_textures(scene, offset_materials=(), offset=0.5):
    """The scene's descriptor with every texture and normal map given as three floats per texel
    (MIPT_TEX_F32X3), the layout the reference's GPU path is handed (CUDAMaterial::pTexData,
    PathTracer.h:21, filled at RayTracerView.cpp:819-842: the bytes of the DWORD texel as floats,
    .x = blue).  Textures of the materials in offset_materials get `offset` added to every value, so
    that they hold numbers no byte can hold.  Returns (pointer to desc, keepalive)."""
    import ctypes as C
    src = scene.desc.contents
    mats = (abi.Material * src.num_materials)()
    keep = [mats]

    def to_float(ptr, w, h, shift):
        dw = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(w * h,))
        fl = np.stack([dw & 255, (dw >> 8) & 255, (dw >> 16) & 255], axis=1).astype(np.float32)
        if shift:
            fl += np.float32(shift)
        keep.append(fl)
        return fl.ctypes.data

    for i in range(src.num_materials):
        m = src.materials[i]
        o = mats[i]
        o.diffuse = m.diffuse
        shift = offset if i in offset_materials else 0.0
        if m.tex_data and m.tex_format == abi.MIPT_TEX_BGRA8:
            o.tex_data, o.tex_width, o.tex_height, o.tex_format = to_float(m.tex_data, m.tex_width, m.tex_height, shift), m.tex_width, m.tex_height, abi.MIPT_TEX_F32X3
        if m.nml_data and m.nml_format == abi.MIPT_TEX_BGRA8:
            o.nml_data, o.nml_width, o.nml_height, o.nml_format = to_float(m.nml_data, m.nml_width, m.nml_height, shift), m.nml_width, m.nml_height, abi.MIPT_TEX_F32X3
    desc = abi.SceneDesc()
    desc.vertices, desc.num_triangles = src.vertices, src.num_triangles
    desc.meshes, desc.num_meshes = src.meshes, src.num_meshes
    desc.materials, desc.num_materials = C.cast(mats, C.POINTER(abi.Material)), src.num_materials
    keep.append(scene)
    return C.pointer(desc), keep

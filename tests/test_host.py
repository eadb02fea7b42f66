"""Host logic of the headless harness: OBJ/MTL reading, scene assembly, camera marshalling,
the procedural atrium, PNG decoding, the 8-bit output stage."""
import math
import os
import struct
import zlib

import numpy as np
import pytest

import scenes
from mipt import abi, host, sharding


def test_camera_preset_matches_reference_numbers():
    # RayTracerView.cpp:211: fovY = pi/4 -> P(1,1) = cot(pi/8), P(0,0) = P(1,1) / aspect
    cam, sun = host.camera_preset(abi.SCENE_CRYTEK, 1920, 1080, 1)
    assert abs(cam.p1 - 1 / math.tan(math.pi / 8)) < 1e-6 and abs(cam.p0 - cam.p1 * 1080 / 1920) < 1e-6
    m = np.array(cam.to_local[:]).reshape(4, 4)
    assert np.allclose(m[3], [-1280, 138, -50, 1])                      # eye position, :224
    assert np.allclose(m[2, :3], [1, 0, 0], atol=1e-5)                  # look = +x after RotateY(3.1416/2), :225
    assert np.allclose(sun, [0, 2000 * math.sin(0.45 * math.pi), 2000 * math.cos(0.45 * math.pi)], rtol=1e-6)
    cam, sun = host.camera_preset(abi.SCENE_SPONZA, 1920, 1080, 2)
    m = np.array(cam.to_local[:]).reshape(4, 4)
    assert np.allclose(m[3, :3], [10, 5, 0]) and np.allclose(m[2, :3], [-1, 0, 0], atol=1e-5) and cam.image_size == (960, 540)
    assert abs(sun[1] - 80 * math.sin(0.45 * math.pi)) < 1e-4
    with pytest.raises(ValueError):
        host.camera_preset(7, 1920, 1080, 1)
    # to_local is the inverse of a rigid view matrix: orthonormal rotation part
    r = m[:3, :3]
    assert np.allclose(r @ r.T, np.eye(3), atol=1e-6)


def test_default_params_follow_the_ui_defaults():
    p = host.default_params()
    assert (p.num_samples, p.global_illumination, p.use_textures, p.use_normals) == (16, 1, 1, 0)   # SideView.cpp:81-119
    assert p.sun_intensity == 3.0 and abs(p.epsilon - 1e-20) < 1e-26 and p.seed == 0x5EED


def test_obj_loader_features(tmp_path):
    (tmp_path / "m.mtl").write_text("newmtl red\nKd 1 0 0\nnewmtl grey\nKd 0.5\nmap_Kd missing.png\nnewmtl tex\nKd 0.1 0.2 0.3\nmap_Kd t.png\nmap_Bump n.png\n")
    # 2x2 RGB and 1x1 grey PNGs written by hand (filter type 0)
    def png(path, w, h, ctype, rows):
        raw = b"".join(b"\x00" + bytes(r) for r in rows)
        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
        path.write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                         chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    png(tmp_path / "t.png", 2, 2, 2, [[255, 0, 0, 0, 255, 0], [0, 0, 255, 10, 20, 30]])
    png(tmp_path / "n.png", 1, 1, 0, [[128]])
    (tmp_path / "m.obj").write_text(
        "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvt 1 1\nvn 0 0 1\n"
        "f 1/1/1 2/2/1 3/3/1\n"                      # before any usemtl: belongs to no mesh (MainFrm.cpp:315)
        "usemtl red\nf 1/1/1 2/2/1 3/3/1 4/4/1\n"    # a quad: only its first three corners are used (:364-400)
        "usemtl tex\nf 1/1/1 2/2/1 3/3/1\nf -3/-3/-1 -2/-2/-1 -1/-1/-1\nf 2/2/1 4/4/1 3/3/1\n"
        "usemtl grey\nf 1/1/1 3/3/1 2/2/1\n")
    s = host.HostScene.load_obj(str(tmp_path / "m.obj"))
    d = s.desc.contents
    # meshes sorted by triangle count, descending and stable: tex(3), red(1), grey(1)
    assert [m[1] for m in s.meshes()] == [3, 1, 1]
    # material "grey" names a diffuse map that does not load -> it is dropped (MainFrm.cpp:249-297): 2 materials remain
    assert d.num_materials == 2
    mats = [m[2] for m in s.meshes()]
    assert mats == [1, 0, -1]
    assert np.allclose(d.materials[0].diffuse[:], [0, 0, 1])            # Kd 1 0 0 stored BGR
    t = d.materials[1]
    assert (t.tex_width, t.tex_height, t.tex_format, t.nml_width, t.nml_height) == (2, 2, abi.MIPT_TEX_BGRA8, 1, 1)
    tex = np.ctypeslib.as_array(C_cast_u32(t.tex_data), shape=(4,))
    assert [hex(x) for x in tex] == ["0xffff0000", "0xff00ff00", "0xff0000ff", "0xff0a141e"]   # byte0 = blue
    v = s.vertices()
    assert np.allclose(v[3, :, :3], [[0, 0, 0], [1, 0, 0], [0, 1, 0]])  # the quad's first three corners
    flipped = host.HostScene.load_obj(str(tmp_path / "m.obj"), flip_winding=True)
    assert np.allclose(flipped.vertices()[3, :, :3], [[0, 0, 0], [0, 1, 0], [1, 0, 0]])
    with pytest.raises(RuntimeError):
        host.HostScene.load_obj(str(tmp_path / "absent.obj"))


def C_cast_u32(ptr):
    import ctypes as C
    return C.cast(ptr, C.POINTER(C.c_uint32))


def test_png_decoder_against_pil(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    for mode, shape in (("RGB", (37, 53, 3)), ("RGBA", (16, 16, 4)), ("L", (31, 7))):
        arr = rng.integers(0, 256, size=shape, dtype=np.uint8)
        path = tmp_path / f"{mode}.png"
        Image.fromarray(arr, mode).save(path)
        got = host.load_png(str(path))
        rgb = np.array(Image.open(path).convert("RGB")).astype(np.uint32)
        want = rgb[..., 2] | (rgb[..., 1] << 8) | (rgb[..., 0] << 16) | (255 << 24)
        assert np.array_equal(got, want), mode
    pal = Image.fromarray(rng.integers(0, 256, size=(9, 9, 3), dtype=np.uint8), "RGB").convert("P", palette=Image.ADAPTIVE, colors=16)
    pal.save(tmp_path / "p.png")
    rgb = np.array(pal.convert("RGB")).astype(np.uint32)
    assert np.array_equal(host.load_png(str(tmp_path / "p.png")), rgb[..., 2] | (rgb[..., 1] << 8) | (rgb[..., 0] << 16) | (255 << 24))
    with pytest.raises(IOError):
        host.load_png(str(tmp_path / "nope.png"))


def test_atrium_is_deterministic_and_hits_its_budget():
    a = host.HostScene.atrium(abi.SCENE_CRYTEK, 66000, 32, 1)
    b = host.HostScene.atrium(abi.SCENE_CRYTEK, 66000, 32, 1)
    assert abs(a.num_triangles - 66000) < 0.03 * 66000
    assert a.num_triangles == b.num_triangles and np.array_equal(a.vertices(), b.vertices())
    counts = [m[1] for m in a.meshes()]
    assert counts == sorted(counts, reverse=True) and 20 <= a.num_meshes <= 25      # one mesh per material, sorted
    v = a.vertices()
    nrm = np.linalg.norm(v[:, :, 3:6], axis=2)
    assert np.allclose(nrm, 1, atol=1e-3)
    assert v[:, :, 6:].min() < 0 and v[:, :, 6:].max() > 1                          # UVs exercise the wrap loops
    # the camera preset sits inside the court; the default-Sponza variant is 1/120 the size
    lo, hi = v[:, :, :3].reshape(-1, 3).min(0), v[:, :, :3].reshape(-1, 3).max(0)
    assert (lo <= [-1280, 0, -50]).all() and (hi > [-1280, 138, -50]).all()
    s = host.HostScene.atrium(abi.SCENE_SPONZA, 5000, 0, 1)
    sv = s.vertices()[:, :, :3].reshape(-1, 3)
    assert sv.max() < 20 and sv.min() > -20
    d = a.desc.contents
    textured = [i for i in range(d.num_materials) if d.materials[i].tex_width == 32]
    assert len(textured) >= 20 and all(d.materials[i].nml_width == 32 for i in textured)


def test_instancing_multiplies_triangles():
    base = host.HostScene.atrium(abi.SCENE_CRYTEK, 3000, 0, 1)
    inst = base.instanced(4, 5000.0)
    assert inst.num_triangles == 4 * base.num_triangles and inst.num_meshes == base.num_meshes
    assert inst.vertices()[:, :, 0].max() > base.vertices()[:, :, 0].max() + 4000


def test_output_stage_truncates_like_the_reference(tmp_path):
    img = np.array([[0.0, 0.99, 1.0], [127.5, 254.999, 255.0]], dtype=np.float32)
    packed = host.pack_bgra(img)
    assert packed[0] == (0 | (0 << 8) | (1 << 16) | (255 << 24))           # CPUPathTracer.cpp:509: b | g<<8 | r<<16
    assert packed[1] == (127 | (254 << 8) | (255 << 16) | (255 << 24))
    path = tmp_path / "o.ppm"
    host.write_ppm(str(path), img.reshape(1, 2, 3), 2, 1)
    data = path.read_bytes()
    assert data.startswith(b"P6\n2 1\n255\n") and data[-6:] == bytes([1, 0, 0, 255, 254, 127])   # RGB order in the file


def test_shard_fields_and_tile_owner():
    assert sharding.shard_fields(0, 1, 128) == dict(shard_rank=0, shard_count=1, tile_size=32, sample_begin=0, sample_end=0)
    f = [sharding.shard_fields(r, 4, 128, "tiles") for r in range(4)]
    assert [x["shard_rank"] for x in f] == [0, 1, 2, 3] and all(x["shard_count"] == 4 for x in f)
    g = [sharding.shard_fields(r, 3, 10, "samples") for r in range(3)]
    assert [(x["sample_begin"], x["sample_end"]) for x in g] == [(0, 4), (4, 8), (8, 10)]
    g = [sharding.shard_fields(r, 8, 4, "samples") for r in range(8)]
    covered = sorted(n for x in g for n in range(x["sample_begin"], x["sample_end"]))
    assert covered == [0, 1, 2, 3] and g[7]["sample_begin"] == g[7]["sample_end"] == 4
    owners = {sharding.tile_owner(x, y, 100, 4, 32) for x in range(100) for y in range(70)}
    assert owners == {0, 1, 2, 3}
    assert sharding.tile_owner(31, 31, 100, 4, 32) == 0 and sharding.tile_owner(32, 0, 100, 4, 32) == 1
    assert sharding.tile_owner(0, 32, 100, 4, 32) == 0                  # 4 tiles per row -> row 1 starts at tile 4


def test_jpeg_decoder_against_libjpeg(tmp_path):
    """Baseline JPEG files written by Pillow (libjpeg) decode to the same picture as Pillow's own
    decoder gives, up to the rounding of a different inverse DCT; with subsampled chroma Pillow
    interpolates where this decoder replicates, so only the luminance-dominated error is bounded.
    The reference decodes through GDI+ (CImage, MainFrm.cpp:258): parity with it is unpinned."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:77, 0:101]
    img = np.stack([128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 40 + 2 * xx, 255 - 3 * yy], axis=2)
    img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
    cases = [("444.jpg", dict(quality=92, subsampling=0), 2, 0.6), ("420.jpg", dict(quality=85, subsampling=2), 40, 3.0),
             ("422_restart.jpg", dict(quality=75, subsampling=1, restart_marker_blocks=3), 40, 3.0),
             ("optimized.jpg", dict(quality=95, subsampling=0, optimize=True), 2, 0.6)]
    for name, opts, max_err, mean_err in cases:
        path = str(tmp_path / name)
        try:
            PIL.fromarray(img, "RGB").save(path, "JPEG", **opts)
        except TypeError:
            opts.pop("restart_marker_blocks", None)
            PIL.fromarray(img, "RGB").save(path, "JPEG", **opts)
        want = np.asarray(PIL.open(path).convert("RGB")).astype(np.int32)
        tex = host.load_image(path)
        assert tex.shape == (77, 101)
        got = np.stack([(tex >> 16) & 255, (tex >> 8) & 255, tex & 255], axis=2).astype(np.int32)     # byte0 = blue
        assert (tex >> 24 == 255).all()
        err = np.abs(got - want)
        assert err.max() <= max_err and err.mean() <= mean_err, (name, err.max(), err.mean())
    # greyscale
    path = str(tmp_path / "grey.jpg")
    PIL.fromarray(img[:, :, 0], "L").save(path, "JPEG", quality=90)
    want = np.asarray(PIL.open(path)).astype(np.int32)
    tex = host.load_image(path)
    assert np.abs((tex & 255).astype(np.int32) - want).max() <= 2 and ((tex & 255) == ((tex >> 8) & 255)).all()
    # progressive files are refused, not misread
    path = str(tmp_path / "prog.jpg")
    PIL.fromarray(img, "RGB").save(path, "JPEG", quality=90, progressive=True)
    with pytest.raises(IOError):
        host.load_image(path)
    # load_image also takes PNG
    path = str(tmp_path / "a.png")
    PIL.fromarray(img, "RGB").save(path, "PNG")
    tex = host.load_image(path)
    assert np.array_equal(np.stack([(tex >> 16) & 255, (tex >> 8) & 255, tex & 255], axis=2).astype(np.uint8), img)


def test_obj_material_with_jpeg_texture(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    PIL.fromarray(np.full((4, 8, 3), (200, 100, 50), dtype=np.uint8), "RGB").save(str(tmp_path / "wall.jpg"), "JPEG", quality=95, subsampling=0)
    (tmp_path / "m.mtl").write_text("newmtl wall\nKd 1 1 1\nmap_Kd wall.jpg\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nusemtl wall\nf 1/1/1 2/2/1 3/3/1\n")
    s = host.HostScene.load_obj(str(tmp_path / "m.obj"))
    d = s.desc.contents
    assert d.num_materials == 1
    m = d.materials[0]
    assert (m.tex_width, m.tex_height, m.tex_format) == (8, 4, abi.MIPT_TEX_BGRA8)
    import ctypes as C
    tex = np.ctypeslib.as_array(C.cast(m.tex_data, C.POINTER(C.c_uint32)), shape=(4, 8))
    r, g, b = (tex >> 16) & 255, (tex >> 8) & 255, tex & 255
    assert abs(int(r[0, 0]) - 200) <= 2 and abs(int(g[0, 0]) - 100) <= 2 and abs(int(b[0, 0]) - 50) <= 2


def test_png_writer_and_image_compare(tmp_path):
    """The PNG written for an image decodes (by Pillow and by the library's own reader) to the truncated
    bytes of the output stage; the PPM reads back the same; mipt_host_compare_images and pt_imgdiff
    report the figures numpy computes."""
    import subprocess
    from PIL import Image
    rng = np.random.default_rng(3)
    h, w = 23, 41
    img = (rng.random((h, w, 3)) * 255.999).astype(np.float32)          # BGR, 0..255
    png, ppm = tmp_path / "o.png", tmp_path / "o.ppm"
    assert host.write_png(str(png), img, w, h) == 0
    assert host.write_ppm(str(ppm), img, w, h) == 0
    want_rgb = np.floor(img[..., ::-1]).astype(np.uint8)
    assert np.array_equal(np.array(Image.open(png).convert("RGB")), want_rgb)
    back = host.read_image(str(png))
    assert back.shape == (h, w, 3) and np.array_equal(back, np.floor(img))
    assert np.array_equal(host.read_image(str(ppm)), np.floor(img))
    with pytest.raises(OSError):
        host.read_image(str(tmp_path / "missing.png"))

    other = np.clip(img + rng.normal(0, 2.0, img.shape).astype(np.float32), 0, 255)
    d = host.compare_images(other, img)
    diff = other.astype(np.float64) - img.astype(np.float64)
    assert math.isclose(d.rmse, math.sqrt((diff ** 2).mean()), rel_tol=1e-9)
    assert math.isclose(d.max_abs, np.abs(diff).max(), rel_tol=1e-9)
    assert math.isclose(d.mean_diff, abs(other.astype(np.float64).mean() - img.astype(np.float64).mean()), rel_tol=1e-6, abs_tol=1e-9)
    assert math.isclose(d.sigma, img.astype(np.float64).std() / math.sqrt(img.size), rel_tol=1e-6)
    assert d.identical < 0.2 and host.compare_images(img, img).identical == 1.0

    tool = os.path.join(os.path.dirname(host.HOST_LIB_PATH), "pt_imgdiff")
    same = subprocess.run([tool, str(png), str(ppm)], capture_output=True, text=True)
    assert same.returncode == 0 and "identical 1.0000" in same.stdout
    host.write_png(str(tmp_path / "bright.png"), np.clip(img + 40, 0, 255), w, h)
    differ = subprocess.run([tool, str(png), str(tmp_path / "bright.png")], capture_output=True, text=True)
    assert differ.returncode == 1 and "differ" in differ.stdout
    host.write_png(str(tmp_path / "small.png"), img[:5], w, 5)
    assert subprocess.run([tool, str(png), str(tmp_path / "small.png")], capture_output=True).returncode == 1


def test_decoders_refuse_malformed_headers(tmp_path):
    """Files that lie about themselves are refused, not decoded with missing tables or sizes: a JPEG whose
    frame header names one component id twice (the scan then leaves a component without Huffman tables), a
    JPEG with two frame headers, a PNG whose IHDR chunk is shorter than its 13 bytes, a PNG with absurd
    dimensions."""
    import struct
    import zlib
    PIL = pytest.importorskip("PIL.Image")
    img = (np.arange(16 * 16 * 3, dtype=np.uint32).reshape(16, 16, 3) * 7 % 256).astype(np.uint8)
    good = str(tmp_path / "good.jpg")
    PIL.fromarray(img, "RGB").save(good, "JPEG", quality=90, subsampling=0)
    assert host.load_image(good).shape == (16, 16)
    data = bytearray(open(good, "rb").read())
    sof = data.index(b"\xff\xc0")
    assert data[sof + 9] == 3                            # three components; ids at +10, +13, +16
    dup = bytearray(data)
    dup[sof + 13] = dup[sof + 10]                        # second component takes the id of the first
    bad = tmp_path / "dup_ids.jpg"
    bad.write_bytes(bytes(dup))
    with pytest.raises(IOError):
        host.load_image(str(bad))
    seg_len = struct.unpack(">H", data[sof + 2:sof + 4])[0]
    twice = data[:sof + 2 + seg_len] + data[sof:sof + 2 + seg_len] + data[sof + 2 + seg_len:]
    bad = tmp_path / "two_sof.jpg"
    bad.write_bytes(bytes(twice))
    with pytest.raises(IOError):
        host.load_image(str(bad))

    def chunk(kind, payload):
        return struct.pack(">I", len(payload)) + kind + payload + struct.pack(">I", zlib.crc32(kind + payload))

    raw = b"".join(b"\x00" + bytes(row) for row in img.reshape(16, 48))
    sig = b"\x89PNG\r\n\x1a\n"
    ihdr = struct.pack(">IIBBBBB", 16, 16, 8, 2, 0, 0, 0)
    ok = tmp_path / "ok.png"
    ok.write_bytes(sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    assert host.load_png(str(ok)).shape == (16, 16)
    short = tmp_path / "short_ihdr.png"
    short.write_bytes(sig + chunk(b"IHDR", ihdr[:12]) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    with pytest.raises(IOError):
        host.load_png(str(short))
    huge = tmp_path / "huge.png"
    huge.write_bytes(sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 1 << 30, 1 << 30, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    with pytest.raises(IOError):
        host.load_png(str(huge))

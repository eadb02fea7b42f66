"""Binding of the headless harness library (host/libmipt_host.so): scenes, cameras, output stage."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "..", "host", "libmipt_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise RuntimeError(f"{HOST_LIB_PATH} is missing: run `make -C cuda-path-tracer_amd` first")
        L = C.CDLL(HOST_LIB_PATH)
        L.mipt_host_scene_load_obj.restype = C.c_void_p
        L.mipt_host_scene_load_obj.argtypes = [C.c_char_p, C.c_int32]
        L.mipt_host_scene_atrium.restype = C.c_void_p
        L.mipt_host_scene_atrium.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_uint32]
        L.mipt_host_scene_instanced.restype = C.c_void_p
        L.mipt_host_scene_instanced.argtypes = [C.c_void_p, C.c_int32, C.c_float]
        L.mipt_host_scene_free.argtypes = [C.c_void_p]
        L.mipt_host_scene_desc.restype = C.POINTER(abi.SceneDesc)
        L.mipt_host_scene_desc.argtypes = [C.c_void_p]
        L.mipt_host_scene_label.restype = C.c_char_p
        L.mipt_host_scene_label.argtypes = [C.c_void_p]
        L.mipt_host_camera_preset.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                              C.POINTER(abi.Camera), C.POINTER(C.c_float)]
        L.mipt_host_camera_make.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_int32,
                                            C.c_int32, C.c_int32, C.POINTER(abi.Camera)]
        L.mipt_host_default_params.argtypes = [C.POINTER(abi.Params)]
        L.mipt_host_pack_bgra.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.mipt_host_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.mipt_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.mipt_host_compare_images.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(abi.ImageDiff)]
        L.mipt_host_read_image_bgr255.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int32),
                                                  C.POINTER(C.c_int32)]
        L.mipt_host_load_png.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mipt_host_load_image.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.mipt_host_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class HostScene:
    """Owns a harness scene; .desc is the mipt_scene_desc handed to the render library."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("scene construction failed")
        self._h = C.c_void_p(handle)
        self.desc = lib().mipt_host_scene_desc(self._h)
        self.label = lib().mipt_host_scene_label(self._h).decode()

    @classmethod
    def load_obj(cls, path, flip_winding=False):
        return cls(lib().mipt_host_scene_load_obj(os.fsencode(path), int(flip_winding)))

    @classmethod
    def atrium(cls, kind, target_triangles, texture_size=0, seed=1):
        return cls(lib().mipt_host_scene_atrium(kind, target_triangles, texture_size, seed))

    def instanced(self, copies, spacing):
        return HostScene(lib().mipt_host_scene_instanced(self._h, copies, spacing))

    @property
    def num_triangles(self):
        return self.desc.contents.num_triangles

    @property
    def num_meshes(self):
        return self.desc.contents.num_meshes

    def vertices(self):
        """(num_triangles, 3, 8) float32 view: pos3, normal3, uv2 per corner."""
        n = self.num_triangles
        buf = C.cast(self.desc.contents.vertices, C.POINTER(C.c_float * (n * 24))).contents
        return np.frombuffer(buf, dtype=np.float32).reshape(n, 3, 8)

    def meshes(self):
        d = self.desc.contents
        return [(d.meshes[i].first_triangle, d.meshes[i].num_triangles, d.meshes[i].material, d.meshes[i].is_light)
                for i in range(d.num_meshes)]

    def close(self):
        if self._h:
            lib().mipt_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_preset(kind, client_width, client_height, div=1):
    cam = abi.Camera()
    sun = (C.c_float * 3)()
    rc = lib().mipt_host_camera_preset(kind, client_width, client_height, div, C.byref(cam), sun)
    if rc != 0:
        raise ValueError("bad camera preset arguments")
    return cam, [sun[0], sun[1], sun[2]]


def camera_make(pos, yaw, pitch, client_width, client_height, div=1, fov_y=0.25 * 3.1415926535):
    cam = abi.Camera()
    p = (C.c_float * 3)(*pos)
    rc = lib().mipt_host_camera_make(p, yaw, pitch, fov_y, client_width, client_height, div, C.byref(cam))
    if rc != 0:
        raise ValueError("bad camera arguments")
    return cam


def default_params(**overrides):
    p = abi.Params()
    lib().mipt_host_default_params(C.byref(p))
    for k, v in overrides.items():
        if k == "sun_pos":
            p.sun_pos[0], p.sun_pos[1], p.sun_pos[2] = v
        else:
            setattr(p, k, v)
    return p


def pack_bgra(bgr255):
    a = np.ascontiguousarray(bgr255, dtype=np.float32).reshape(-1, 3)
    out = np.empty(a.shape[0], dtype=np.uint32)
    lib().mipt_host_pack_bgra(a.ctypes.data, a.shape[0], out.ctypes.data)
    return out


def write_ppm(path, bgr255, width, height):
    a = np.ascontiguousarray(bgr255, dtype=np.float32)
    return lib().mipt_host_write_ppm(os.fsencode(path), a.ctypes.data, width, height)


def write_png(path, bgr255, width, height):
    a = np.ascontiguousarray(bgr255, dtype=np.float32)
    return lib().mipt_host_write_png(os.fsencode(path), a.ctypes.data, width, height)


def compare_images(a, b):
    """rmse, max_abs, mean_diff, sigma, identical of two float 255*rgb images (mipt_host_compare_images)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    if a.size != b.size:
        raise ValueError("images differ in size")
    d = abi.ImageDiff()
    rc = lib().mipt_host_compare_images(a.ctypes.data, b.ctypes.data, C.c_int64(a.size), C.byref(d))
    if rc != 0:
        raise ValueError("mipt_host_compare_images failed")
    return d


def read_image(path):
    """Image file (PPM of this library, PNG, baseline JPEG) -> float32 array (height, width, 3), BGR, 0..255."""
    out = C.POINTER(C.c_float)()
    w, h = C.c_int32(), C.c_int32()
    rc = lib().mipt_host_read_image_bgr255(os.fsencode(path), C.byref(out), C.byref(w), C.byref(h))
    if rc != 0:
        raise OSError(f"cannot read image {path}")
    try:
        return np.ctypeslib.as_array(out, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().mipt_host_free(out)


def load_png(path):
    return _load(lib().mipt_host_load_png, path)


def load_image(path):
    """PNG or baseline JPEG -> (h, w) uint32 BGRA8 texels."""
    return _load(lib().mipt_host_load_image, path)


def _load(fn, path):
    ptr = C.c_void_p()
    w, h = C.c_int32(), C.c_int32()
    rc = fn(os.fsencode(path), C.byref(ptr), C.byref(w), C.byref(h))
    if rc != 0:
        raise IOError(f"cannot decode {path}")
    arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(h.value, w.value)).copy()
    lib().mipt_host_free(ptr)
    return arr

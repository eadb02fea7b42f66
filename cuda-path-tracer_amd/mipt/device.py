"""Binding of the HIP library (csrc/libmipt.so) through its C ABI (include/mipt.h).

There is no CPU fallback: if the library is missing or no GPU is present every call
raises."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "..", "csrc", "libmipt.so")
_lib = None


class MiptError(RuntimeError):
    pass


class MiptCancelled(MiptError):
    """mipt_frame_accumulate was ended by mipt_frame_cancel; .stats holds the partial call's statistics."""
    stats = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiptError(f"{LIB_PATH} is missing: run `make -C cuda-path-tracer_amd` (hipcc, gfx950) first")
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        L.mipt_init.argtypes = [i32]
        L.mipt_last_error.restype = C.c_char_p
        L.mipt_device_name.argtypes = [C.c_char_p, i32]
        L.mipt_scene_create.argtypes = [C.POINTER(abi.SceneDesc), C.POINTER(abi.BuildOptions), C.POINTER(vp)]
        L.mipt_scene_destroy.argtypes = [vp]
        L.mipt_scene_destroy.restype = None
        L.mipt_scene_get_info.argtypes = [vp, C.POINTER(abi.SceneInfo)]
        L.mipt_scene_bvh_sizes.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_uint32)]
        L.mipt_scene_bvh_grid.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.mipt_scene_download_bvh.argtypes = [vp, vp, vp, vp, vp]
        L.mipt_trace_rays.argtypes = [vp, vp, i64, f32, vp, vp, vp]
        L.mipt_trace_rays_device.argtypes = [vp, vp, i64, f32, vp, vp, vp]
        L.mipt_frame_create.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.Params), C.POINTER(vp)]
        L.mipt_frame_destroy.argtypes = [vp]
        L.mipt_frame_destroy.restype = None
        L.mipt_frame_set_params.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.Params)]
        L.mipt_frame_accumulate.argtypes = [vp, vp, C.POINTER(abi.Stats), abi.PROGRESS_FN, vp]
        L.mipt_frame_finalize.argtypes = [vp, vp, vp]
        L.mipt_frame_finalize_partial.argtypes = [vp, vp, i32, vp]
        L.mipt_frame_cancel.argtypes = [vp]
        L.mipt_frame_samples_done.argtypes = [vp]
        L.mipt_frame_samples_done.restype = i32
        L.mipt_frame_primary_ids.argtypes = [vp, vp]
        L.mipt_frame_num_pixels.argtypes = [vp]
        L.mipt_frame_num_pixels.restype = i64
        L.mipt_render.argtypes = [vp, C.POINTER(abi.Camera), C.POINTER(abi.Params), vp, vp, C.POINTER(abi.Stats),
                                  abi.PROGRESS_FN, vp]
        L.mipt_device_alloc.argtypes = [C.POINTER(vp), i64]
        L.mipt_device_free.argtypes = [vp]
        L.mipt_copy_to_host.argtypes = [vp, vp, i64]
        L.mipt_copy_to_device.argtypes = [vp, vp, i64]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise MiptError(f"mipt error {rc}: {lib().mipt_last_error().decode()}")


def init(device=0):
    _check(lib().mipt_init(device))


def device_name():
    buf = C.create_string_buffer(256)
    _check(lib().mipt_device_name(buf, 256))
    return buf.value.decode()


_NO_PROGRESS = C.cast(None, abi.PROGRESS_FN)


class Scene:
    """Device-resident scene: geometry, textures and the GPU-built BVH."""

    def __init__(self, desc_ptr, sah_bins=0, max_leaf_triangles=0, traversal_cost=0.0):
        opt = abi.BuildOptions(sah_bins, max_leaf_triangles, traversal_cost, 0)
        h = C.c_void_p()
        _check(lib().mipt_scene_create(desc_ptr, C.byref(opt), C.byref(h)))
        self._h = h

    def info(self):
        inf = abi.SceneInfo()
        _check(lib().mipt_scene_get_info(self._h, C.byref(inf)))
        return inf

    def download_bvh(self):
        ns, ls, root = C.c_int64(), C.c_int64(), C.c_uint32()
        _check(lib().mipt_scene_bvh_sizes(self._h, C.byref(ns), C.byref(ls), C.byref(root)))
        nodes = np.zeros((ns.value, 4 * self.info().node_width), dtype=np.uint32)
        v0 = np.zeros((ls.value, 4), dtype=np.float32)
        e1 = np.zeros((ls.value, 4), dtype=np.float32)
        e2 = np.zeros((ls.value, 4), dtype=np.float32)
        _check(lib().mipt_scene_download_bvh(self._h, nodes.ctypes.data, v0.ctypes.data, e1.ctypes.data, e2.ctypes.data))
        qmin, qcell = (C.c_float * 3)(), (C.c_float * 3)()
        _check(lib().mipt_scene_bvh_grid(self._h, qmin, qcell))
        grid = (np.array(qmin[:], dtype=np.float32), np.array(qcell[:], dtype=np.float32))
        return nodes, v0, e1, e2, root.value, grid

    def trace_rays(self, rays, eps=1e-20, closest=True, anyhit=True):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
        n = rays.shape[0]
        tri = np.empty(n, dtype=np.int32) if closest else None
        tuv = np.empty((n, 3), dtype=np.float32) if closest else None
        ah = np.empty(n, dtype=np.uint8) if anyhit else None
        _check(lib().mipt_trace_rays(self._h, rays.ctypes.data, n, eps,
                                     tri.ctypes.data if closest else None, tuv.ctypes.data if closest else None,
                                     ah.ctypes.data if anyhit else None))
        return tri, tuv, ah

    def trace_rays_device(self, d_rays_ptr, n, eps=1e-20, d_tri_ptr=None, d_tuv_ptr=None, d_any_ptr=None):
        """mipt_trace_rays_device: rays and results are DEVICE pointers (ints); nothing is allocated."""
        _check(lib().mipt_trace_rays_device(self._h, C.c_void_p(d_rays_ptr), n, eps, C.c_void_p(d_tri_ptr),
                                            C.c_void_p(d_tuv_ptr), C.c_void_p(d_any_ptr)))

    def render(self, cam, params, want_ids=True, progress=None):
        """mipt_render: whole image into host arrays -> (bgr255 (H,W,3), ids (H,W) or None, Stats)."""
        w, h = cam.image_size
        img = np.zeros((h, w, 3), dtype=np.float32)
        ids = np.zeros((h, w), dtype=np.int32) if want_ids else None
        st = abi.Stats()
        cb = abi.PROGRESS_FN(progress) if progress else _NO_PROGRESS
        _check(lib().mipt_render(self._h, C.byref(cam), C.byref(params), img.ctypes.data,
                                 ids.ctypes.data if want_ids else None, C.byref(st), cb, None))
        return img, ids, st

    def close(self):
        if self._h:
            lib().mipt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Frame:
    """Work buffers of one image size; accumulate/finalize take DEVICE pointers (ints)."""

    def __init__(self, scene, cam, params):
        h = C.c_void_p()
        _check(lib().mipt_frame_create(scene._h, C.byref(cam), C.byref(params), C.byref(h)))
        self._h = h
        self.scene = scene
        self.num_pixels = lib().mipt_frame_num_pixels(h)
        self.size = cam.image_size

    def set_params(self, cam, params):
        _check(lib().mipt_frame_set_params(self._h, C.byref(cam), C.byref(params)))

    def accumulate(self, d_accum_ptr, progress=None):
        """mipt_frame_accumulate.  Raises MiptCancelled (stats attached) if cancel() ended it early."""
        st = abi.Stats()
        cb = abi.PROGRESS_FN(progress) if progress else _NO_PROGRESS
        rc = lib().mipt_frame_accumulate(self._h, C.c_void_p(d_accum_ptr), C.byref(st), cb, None)
        if rc == abi.E_CANCELLED:
            e = MiptCancelled(lib().mipt_last_error().decode())
            e.stats = st
            raise e
        _check(rc)
        return st

    def cancel(self):
        _check(lib().mipt_frame_cancel(self._h))

    @property
    def samples_done(self):
        return lib().mipt_frame_samples_done(self._h)

    def finalize(self, d_accum_ptr, d_out_ptr):
        _check(lib().mipt_frame_finalize(self._h, C.c_void_p(d_accum_ptr), C.c_void_p(d_out_ptr)))

    def finalize_partial(self, d_accum_ptr, samples, d_out_ptr):
        _check(lib().mipt_frame_finalize_partial(self._h, C.c_void_p(d_accum_ptr), int(samples), C.c_void_p(d_out_ptr)))

    def primary_ids(self):
        w, h = self.size
        ids = np.zeros((h, w), dtype=np.int32)
        _check(lib().mipt_frame_primary_ids(self._h, ids.ctypes.data))
        return ids

    def close(self):
        if self._h:
            lib().mipt_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """Device memory through the library's own allocator (for hosts without torch)."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(lib().mipt_device_alloc(C.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes

    def to_host(self, dtype=np.float32):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(lib().mipt_copy_to_host(out.ctypes.data, C.c_void_p(self.ptr), self.nbytes))
        return out

    def from_host(self, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().mipt_copy_to_device(C.c_void_p(self.ptr), arr.ctypes.data, arr.nbytes))

    def free(self):
        if self.ptr:
            lib().mipt_device_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

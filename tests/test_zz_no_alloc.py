"""mipt_trace_rays_device, the entry point that allocates nothing (its ray queue, hit records and
counters live in a fixed array inside the library), against mipt_trace_rays on the same rays and
against the CPU oracle's brute-force search: triangle ids, (t, ua, va) bit patterns and any-hit
flags must be identical.  Two calls in a row share the array; the second is smaller, so stale
entries of the first lie behind its end.  The device's free memory is read before and after a call."""
import numpy as np
import pytest

import oracle_binding as ob
import scenes
from mipt import abi

pytestmark = pytest.mark.gpu


def _run(gpu, sc, rays):
    n = rays.shape[0]
    d_rays, d_tri, d_tuv, d_any = (gpu.DeviceBuffer(24 * n), gpu.DeviceBuffer(4 * n), gpu.DeviceBuffer(12 * n),
                                   gpu.DeviceBuffer(n))
    d_rays.from_host(rays)
    # poison the outputs: every entry must be written by the call
    d_tri.from_host(np.full(n, 0x5A5A5A5A, dtype=np.int32))
    d_tuv.from_host(np.full(3 * n, np.nan, dtype=np.float32))
    d_any.from_host(np.full(n, 0xEE, dtype=np.uint8))
    sc.trace_rays_device(d_rays.ptr, n, 1e-20, d_tri.ptr, d_tuv.ptr, d_any.ptr)
    return d_tri.to_host(np.int32), d_tuv.to_host(np.float32).reshape(n, 3), d_any.to_host(np.uint8)


def test_trace_rays_device_matches_trace_rays_and_the_oracle_on_two_calls(gpu, atrium_small, atrium_small_oracle):
    sc = gpu.Scene(atrium_small.desc)
    for n, seed in ((60000, 5), (7001, 6)):
        rays = scenes.random_rays(atrium_small, n, seed=seed)
        tri, tuv, anyhit = sc.trace_rays(rays)
        dtri, dtuv, dany = _run(gpu, sc, rays)
        assert (tri >= 0).mean() > 0.5
        assert np.array_equal(dtri, tri), f"ids differ at n={n}"
        assert np.array_equal(dtuv.view(np.uint32), tuv.view(np.uint32)), f"(t, ua, va) differ at n={n}"
        assert np.array_equal(dany, anyhit), f"any-hit flags differ at n={n}"
        # and directly against the oracle (the two entry points share their kernels: a fault common to both would pass the lines above)
        otri, otuv, oany = atrium_small_oracle.trace_rays(rays, accel=ob.ACCEL_BRUTE)
        assert np.array_equal(dtri, otri) and np.array_equal(dany, oany)
        assert np.array_equal(dtuv.view(np.uint32), otuv.view(np.uint32))


def _free_device_bytes():
    """hipMemGetInfo of the HIP runtime the library itself runs on (the copy that is mapped into this process)."""
    import ctypes
    with open("/proc/self/maps") as maps:
        path = next(line.split()[-1] for line in maps if "libamdhip64" in line)
    hip = ctypes.CDLL(path)
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_trace_rays_device_leaves_the_free_memory_as_it_was(gpu, atrium_small):
    sc = gpu.Scene(atrium_small.desc)
    n = 50000
    rays = scenes.random_rays(atrium_small, n, seed=12)
    d_rays, d_tri, d_tuv, d_any = (gpu.DeviceBuffer(24 * n), gpu.DeviceBuffer(4 * n), gpu.DeviceBuffer(12 * n), gpu.DeviceBuffer(n))
    d_rays.from_host(rays)
    sc.trace_rays_device(d_rays.ptr, n, 1e-20, d_tri.ptr, d_tuv.ptr, d_any.ptr)      # first call: kernels are loaded, attributes set
    before = _free_device_bytes()
    for _ in range(3):
        sc.trace_rays_device(d_rays.ptr, n, 1e-20, d_tri.ptr, d_tuv.ptr, d_any.ptr)
    after = _free_device_bytes()
    assert after == before, f"free device memory changed by {before - after} bytes over three calls"


def test_trace_rays_device_outputs_are_optional_and_the_limit_is_an_error(gpu, atrium_small):
    sc = gpu.Scene(atrium_small.desc)
    n = 3000
    rays = scenes.random_rays(atrium_small, n, seed=9)
    tri, _, anyhit = sc.trace_rays(rays)
    d_rays, d_tri, d_any = gpu.DeviceBuffer(24 * n), gpu.DeviceBuffer(4 * n), gpu.DeviceBuffer(n)
    d_rays.from_host(rays)
    sc.trace_rays_device(d_rays.ptr, n, 1e-20, d_tri.ptr, None, None)
    assert np.array_equal(d_tri.to_host(np.int32), tri)
    sc.trace_rays_device(d_rays.ptr, n, 1e-20, None, None, d_any.ptr)
    assert np.array_equal(d_any.to_host(np.uint8), anyhit)
    with pytest.raises(gpu.MiptError, match="exceed"):
        sc.trace_rays_device(d_rays.ptr, abi.TRACE_DEVICE_MAX_RAYS + 1, 1e-20, d_tri.ptr, None, None)
    # the largest call the array takes
    nmax = abi.TRACE_DEVICE_MAX_RAYS
    rays = scenes.random_rays(atrium_small, nmax, seed=10, aimed=False)
    tri, tuv, anyhit = sc.trace_rays(rays)
    dtri, dtuv, dany = _run(gpu, sc, rays)
    assert np.array_equal(dtri, tri) and np.array_equal(dtuv.view(np.uint32), tuv.view(np.uint32)) and np.array_equal(dany, anyhit)

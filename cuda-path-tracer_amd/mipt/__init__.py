"""Python side of the MI355X path-tracing core: ctypes bindings of the C ABI.

  mipt.abi     structures of include/mipt.h / include/mipt_host.h
  mipt.host    headless harness (scenes, cameras, output stage) - host only
  mipt.device  the HIP library (scene upload + BVH build, render) - needs a GPU
"""
from . import abi, host  # noqa: F401

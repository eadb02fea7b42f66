"""The C-ABI libraries load without a GPU and export every symbol their headers declare; the
ctypes mirrors have the sizes the C compiler gives the structures."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT
from mipt import abi, device, host

INCLUDE = os.path.join(ROOT, "include")


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mipt_[a-z0-9_]+)\s*\(", text)) - {"mipt_progress_fn"})


def test_libmipt_exports_every_declared_symbol():
    names = declared_functions("mipt.h")
    assert sorted(abi.MIPT_SYMBOLS) == names
    L = device.lib()
    assert [n for n in names if not hasattr(L, n)] == []


def test_libmipt_host_exports_every_declared_symbol():
    names = declared_functions("mipt_host.h")
    assert sorted(abi.MIPT_HOST_SYMBOLS) == names
    L = host.lib()
    assert [n for n in names if not hasattr(L, n)] == []


def test_struct_sizes_match_the_c_compiler(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "mipt.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(mipt_vertex),sizeof(mipt_material),sizeof(mipt_mesh),sizeof(mipt_scene_desc),sizeof(mipt_camera),"
                   "sizeof(mipt_params),sizeof(mipt_build_options),sizeof(mipt_scene_info),sizeof(mipt_stats));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", INCLUDE, str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(t) for t in (abi.Vertex, abi.Material, abi.Mesh, abi.SceneDesc, abi.Camera, abi.Params,
                                  abi.BuildOptions, abi.SceneInfo, abi.Stats)]
    assert got == want
    assert C.sizeof(abi.Vertex) == 32          # CCUDAVertex, PathTracer.h:12-17


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(device.MiptError):
        device.init(0)
    with pytest.raises(device.MiptError):
        device.Scene(host.HostScene.atrium(abi.SCENE_CRYTEK, 2000, 0, 1).desc)


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "cuda-path-tracer_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                if re.search(r"oracle|pto_", open(os.path.join(base, f), errors="ignore").read()):
                    offenders.append(f)
    assert offenders == []

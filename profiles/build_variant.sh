#!/bin/bash
# Builds a variant of the HIP library with extra compiler flags into profiles/ab/<name>.so (for ab_libs.sh).
# usage: bash profiles/build_variant.sh <name> "<flags>"     e.g.  abl_cvt "-DMIPT_ABLATE_CVT"
set -eu
cd "$(dirname "$0")/../cuda-path-tracer_amd"
mkdir -p ../profiles/ab
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function $2 -shared -o "../profiles/ab/$1.so" csrc/mipt_api.hip csrc/bvh_build.hip csrc/pt_kernels.hip

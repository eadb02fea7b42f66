#!/bin/bash
# A/B of library builds on the GPU box: runs the bench once per library given on the command line
# (each is copied over csrc/libmipt.so of the box's scratch copy of the tree).
# usage: bash profiles/ab_libs.sh <outdir> lib1.so lib2.so ...
set -u
OUT=$1; shift
mkdir -p "$OUT"
cp cuda-path-tracer_amd/csrc/libmipt.so "$OUT/base.so"
for L in base "$@"; do
  if [ "$L" != base ]; then cp "$L" cuda-path-tracer_amd/csrc/libmipt.so; else cp "$OUT/base.so" cuda-path-tracer_amd/csrc/libmipt.so; fi
  name=$(basename "$L" .so)
  timeout -k 10 200 python -u bench.py --steps ${AB_STEPS:-5} --warmup 1 --no-cpu-baseline > "$OUT/$name.log" 2>&1 || { echo "$name failed"; break; }
  python3 -c "
import json,sys; d=json.loads(open('$OUT/$name.log').read().strip().splitlines()[-1]); k=d['roofline']['kernel_time_share']; r=d['roofline']; print('$name', 'frame', round(d['ms_per_step'],1), 'closest/frame min', min(r['closest_ms_per_step']), 'mean', round(sum(r['closest_ms_per_step'])/len(r['closest_ms_per_step']),1), 'shadow min', min(r['shadow_ms_per_step']), {n: round(v*d['ms_per_step'],1) for n,v in k.items()}, d['config']['image_sha1'][:8], 'visits/ray', round(r['node_visits_per_s']/r['closest_rays_per_s'],2), 'tests/ray', round(r['triangle_tests_per_s']/r['closest_rays_per_s'],2))"
done
cp "$OUT/base.so" cuda-path-tracer_amd/csrc/libmipt.so      # the tree's own build is back in place
rm -f "$OUT/base.so"

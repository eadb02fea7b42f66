#!/bin/bash
# Effective shader clock per kernel: GRBM_GUI_ACTIVE (cycles, summed over XCDs) against the kernel's duration.
# usage: bash profiles/clock_probe.sh <outdir> lib.so ...
OUT=$1; shift
mkdir -p "$OUT"
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for L in "$@"; do
  name=$(basename "$L" .so)
  cp "$L" "$ROOT/cuda-path-tracer_amd/csrc/libmipt.so"
  (cd /tmp && TMPDIR=/tmp timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT" -o "$name" -- python3 "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1) || { echo "$name failed"; break; }
done

#!/bin/bash
# Phase thresholds of the build with cooperative leaf tests (profiles/ab/coop.so), with the tree's own build at its defaults as the first line
DROPS="28" VOTES="16" STAYS="3" bash profiles/sweep_thresholds.sh | sed 's/^/base: /'
cp cuda-path-tracer_amd/csrc/libmipt.so /tmp/base.so; cp profiles/ab/coop.so cuda-path-tracer_amd/csrc/libmipt.so
bash profiles/sweep_thresholds.sh | sed 's/^/coop: /'
cp /tmp/base.so cuda-path-tracer_amd/csrc/libmipt.so

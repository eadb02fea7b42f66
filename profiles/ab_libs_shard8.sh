#!/bin/bash
# The A/B of ab_libs.sh on one eighth of the frame (the tiles of rank 0 of 8), the share a rank renders in an 8-GPU run.
# usage: bash profiles/ab_libs_shard8.sh <outdir> lib1.so lib2.so ...
MIPT_BENCH_AS_SHARD=0/8 bash "$(dirname "$0")/ab_libs.sh" "$@"

#!/bin/bash
# Samples clocks and power while the bench runs: is the frame time limited by the power cap?
# usage (on the GPU box): bash profiles/power_probe.sh <outfile>
OUT=$1
( for i in $(seq 1 60); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo; sleep 0.2; done ) > "$OUT" 2>&1 &
POLL=$!
timeout -k 10 200 python -u bench.py --steps 30 --warmup 2 --no-cpu-baseline > "$OUT.bench" 2>&1
kill $POLL 2>/dev/null
wait $POLL 2>/dev/null
tail -c 300 "$OUT.bench"

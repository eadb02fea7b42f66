#!/bin/bash
# The measurements a set of files profiles/<prefix>_* is made of, on the GPU box:
#   bash profiles/measure_round.sh <name> <prefix>        e.g.  r3s r03_b
# 1. the counter passes (pmc_passes.sh) and, from them, profiles/traffic.json for the tree's kernel sources
# 2. the bench under rocprofv3 --kernel-trace --stats, and the plain default bench line (which then quotes that traffic)
# 3. the step profile of the closest-hit kernel (profiles/ab/prof.so: -DMIPT_TRACE_PROFILE -DMIPT_TRACE_CLOCK -DMIPT_TRACE_TAIL)
# 4. the other workloads and the one-GPU rehearsal of 2, 4 and 8 tile shards
# Everything lands in gpurun_out/<name>/; the files to commit are copied to gpurun_out/<name>/publish/<prefix>_*.
set -u
NAME=$1; PREFIX=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$NAME; PUB=$OUT/publish; mkdir -p "$OUT" "$PUB"
cd "$ROOT"
bash profiles/pmc_passes.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1; tail -1 "$OUT/pmc.log"
python3 profiles/pmc_summary.py "$OUT/pmc" "$PUB/${PREFIX}_pmc_per_kernel.csv" > /dev/null &&
python3 profiles/pmc_report.py "$PUB/${PREFIX}_pmc_per_kernel.csv" profiles/traffic.json crytek_gi128_1080p "$(cat "$OUT/pmc/source_sha1.txt")" > "$PUB/${PREFIX}_pmc_report_traversal.txt" || echo "counter summary failed"
cp profiles/traffic.json "$PUB/traffic.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o b -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err" || echo "rocprof bench failed"
cd "$ROOT"
tail -1 "$OUT/bench_under_rocprof.json" > "$PUB/${PREFIX}_bench_line_under_rocprof.json"
python3 profiles/summarize.py "$(ls "$OUT"/prof/*/b_kernel_stats.csv "$OUT"/prof/b_kernel_stats.csv 2>/dev/null | head -1)" "$PUB/${PREFIX}_bench_kernel_stats.csv" || echo "kernel statistics failed"
timeout -k 10 400 python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err" || echo "bench failed"
tail -1 "$OUT/bench_line.json" > "$PUB/${PREFIX}_bench_line.json"; tail -c 400 "$PUB/${PREFIX}_bench_line.json"; echo
AB_STEPS=2 bash profiles/ab_libs.sh "$OUT/ab" profiles/ab/prof.so > "$OUT/ab.txt" 2>&1; grep "trace profile\|trace clock\|trace tail" "$OUT/ab/prof.log" | tail -3 > "$PUB/${PREFIX}_step_profile.txt"; cat "$PUB/${PREFIX}_step_profile.txt"
for W in sponza_direct_1080p sponza_gi128_1080p crytek38x_gi512_2160p crytek_gi16_270p; do
  timeout -k 10 300 python -u bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline 2> "$OUT/$W.err" | tail -1 >> "$OUT/workloads.jsonl" || echo "$W failed"
done
for N in 2 4 8; do
  MIPT_BENCH_AS_SHARD=0/$N timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline 2> "$OUT/shard$N.err" | tail -1 >> "$OUT/shards.jsonl" || echo "shard 0/$N failed"
done
python3 - "$OUT" "$PUB/${PREFIX}_workloads_and_shards.txt" <<'EOF'
import json, sys
out = open(sys.argv[2], "w")
for name in ("workloads.jsonl", "shards.jsonl"):
    for line in open(sys.argv[1] + "/" + name):
        d = json.loads(line); c = d["config"]; r = d["roofline"]
        print(name[:-6], c["workload"], c.get("parallelism"), "triangles", c["triangles"], "image", c["image"], "ms_per_frame %.2f" % d["ms_per_step"], "Mrays/s %.0f" % d["value"],
              "reference rays per frame %.4g" % c["reference_rays_per_frame"], "not traversed %.4g" % c["shadow_rays_not_traversed_per_frame"],
              "closest ms", min(r["closest_ms_per_step"]), "shadow ms", min(r["shadow_ms_per_step"]), "bvh build ms %.1f" % c["bvh_build_ms"], c["image_sha1"][:8], file=out)
out.close()
print(open(sys.argv[2]).read())
EOF

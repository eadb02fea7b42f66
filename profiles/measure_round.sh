set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3r; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err || echo "rocprof bench failed"
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err || echo "bench failed"
tail -c 600 $OUT/bench_line.json
bash profiles/pmc_passes.sh $OUT/pmc > $OUT/pmc.log 2>&1; tail -2 $OUT/pmc.log
AB_STEPS=2 bash profiles/ab_libs.sh $OUT/ab profiles/ab/prof.so; grep "trace profile\|trace clock" $OUT/ab/prof.log | tail -2

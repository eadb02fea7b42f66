set -u
mkdir -p gpurun_out/r2i
MIPT_BVH_REINSERT=2 timeout -k 10 300 python -u -m pytest tests/test_gpu_a_bvh.py tests/test_gpu_b_trace.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -3
for P in 0 1 2 4; do
MIPT_BVH_REINSERT=$P MIPT_BUILD_VERBOSE=1 timeout -k 10 400 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2i/bench$P.txt 2>&1 || { tail -5 gpurun_out/r2i/bench$P.txt; exit 1; }
grep "reinsertion" gpurun_out/r2i/bench$P.txt | head -1
python3 -c "
import json; d=json.loads(open('gpurun_out/r2i/bench$P.txt').read().strip().splitlines()[-1]); r=d['roofline']; c=min(r['closest_ms_per_step'])
print('passes $P frame', round(d['ms_per_step'],1), 'closest', c, 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8], 'visits/ray %.2f tests/ray %.2f depth %d' % (r['node_visits_per_s']/r['closest_rays_per_s'], r['triangle_tests_per_s']/r['closest_rays_per_s'], d['config']['bvh_depth']))"
done

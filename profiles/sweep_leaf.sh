#!/bin/bash
# usage: bash profiles/sweep_leaf.sh -> one bench run per (largest leaf, SAH traversal cost) of the BVH build
for lf in ${LEAVES:-2 3 4 6 8}; do for tc in ${TCOSTS:-0.7 1 1.5 2.5}; do
  env MIPT_BENCH_LEAF=$lf MIPT_BENCH_TCOST=$tc timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/s.log 2>&1 || { echo failed; tail -3 /tmp/s.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); r=d['roofline']; print('leaf=$lf tcost=$tc frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), 'visits/ray %.2f tests/ray %.2f' % (r['node_visits_per_s']/r['closest_rays_per_s'], r['triangle_tests_per_s']/r['closest_rays_per_s']), d['config']['image_sha1'][:8])"
done; done

#!/bin/bash
# usage: bash profiles/sweep_env2.sh "A=1 B=2" "A=2 B=3" ...   -> one bench run per environment setting
for S in "$@"; do
  env $S timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/sweep_env.log 2>&1 || { echo "$S failed"; tail -3 /tmp/sweep_env.log; break; }
  python3 -c "
import json; d=json.loads(open('/tmp/sweep_env.log').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('$S', round(d['ms_per_step'],1), 'closest', round(r['avg_launch_ms'],2), 'nodes/ray', round(r['node_visits_per_s']/r['closest_rays_per_s'],1), 'tris/ray', round(r['triangle_tests_per_s']/r['closest_rays_per_s'],1), 'sah', round(c['bvh_sah_cost'],1), 'depth', c['bvh_depth'])"
done

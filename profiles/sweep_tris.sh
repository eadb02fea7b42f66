#!/bin/bash
# bench at several triangle counts: how the traversal rate changes with the size of the BVH
for T in "$@"; do
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --tris $T > /tmp/sweep_tris.log 2>&1 || { echo "tris=$T failed"; tail -3 /tmp/sweep_tris.log; break; }
  python3 -c "
import json; d=json.loads(open('/tmp/sweep_tris.log').read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
print('tris', c['triangles'], 'ms', round(d['ms_per_step'],1), 'Mrays/s', round(d['value']), 'closest Grays/s', round(r['closest_rays_per_s']/1e9,2), 'Gnodes/s', round(r['node_visits_per_s']/1e9,1), 'Gtris/s', round(r['triangle_tests_per_s']/1e9,1), 'nodes/ray', round(r['node_visits_per_s']/r['closest_rays_per_s'],1), 'build ms', round(c['bvh_build_ms'],1))"
done

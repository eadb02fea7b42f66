#!/bin/bash
# closest-hit and any-hit time against the number of node slots in the LDS copy (same launch shape otherwise)
for c in ${SLOTS:-128 256 384 512}; do
  env MIPT_TRACE_CACHE=$c timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/s.log 2>&1 || { echo failed; tail -3 /tmp/s.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); r=d['roofline']; print('cache slots $c frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8])"
done

#!/bin/bash
# Traversal time against resident waves per compute unit: extra (unused) LDS per workgroup lowers the number of
# workgroups a compute unit holds.  usage: bash profiles/sweep_occupancy.sh "W:EXTRA" ...
for WE in "$@"; do
  IFS=: read W E <<< "$WE"
  env MIPT_TRACE_WAVES=$W MIPT_TRACE_LDS_EXTRA=$E timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/occ.log 2>&1 || { echo "waves=$W extra=$E failed"; tail -3 /tmp/occ.log; break; }
  python3 -c "
import json; d=json.loads(open('/tmp/occ.log').read().strip().splitlines()[-1]); r=d['roofline']
print('waves/wg=$W extra_lds=$E frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8])"
done

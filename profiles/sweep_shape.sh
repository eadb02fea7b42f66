#!/bin/bash
# usage: bash profiles/sweep_shape.sh "W:C[:S]" ...  -> one bench run per (waves per workgroup, node slots cached in LDS,
# stack rows kept in LDS above the two bottom rows; the scene is built under MIPT_STACK_LDS, which sets that number)
for WCS in "$@"; do
  IFS=: read W C S <<< "$WCS"
  env MIPT_TRACE_WAVES=$W MIPT_TRACE_CACHE=$C ${S:+MIPT_STACK_LDS=$S} timeout -k 10 200 python -u bench.py --steps ${SWEEP_STEPS:-3} --warmup 1 --no-cpu-baseline > /tmp/sweep_shape.log 2>&1 || { echo "waves=$W cache=$C failed"; tail -3 /tmp/sweep_shape.log; break; }
  python3 -c "
import json; d=json.loads(open('/tmp/sweep_shape.log').read().strip().splitlines()[-1]); r=d['roofline']; k=r['kernel_time_share']
print('waves=$W cache=$C stack=${S:-default} frame', round(d['ms_per_step'],1), 'closest/frame min', min(r['closest_ms_per_step']), 'shadow min', min(r['shadow_ms_per_step']), {n: round(v*d['ms_per_step'],1) for n,v in k.items()}, d['config']['image_sha1'][:8])"
done

#!/bin/bash
# usage: bash profiles/sweep_env.sh VAR v1 v2 ...  -> one bench run per value of one environment switch of the library
VAR=$1; shift
for V in "$@"; do
  env $VAR=$V timeout -k 10 200 python -u bench.py --steps ${SWEEP_STEPS:-3} --warmup 1 --no-cpu-baseline > /tmp/sweep_env.log 2>&1 || { echo "$VAR=$V failed"; tail -3 /tmp/sweep_env.log; break; }
  python3 -c "
import json; d=json.loads(open('/tmp/sweep_env.log').read().strip().splitlines()[-1]); r=d['roofline']
print('$VAR=$V frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8])"
done

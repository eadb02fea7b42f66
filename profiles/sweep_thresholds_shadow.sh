#!/bin/bash
# usage: bash profiles/sweep_thresholds_shadow.sh  -> one bench run per (node drop, triangle vote, triangle stay) of the any-hit kernel's phase thresholds
for nd in ${DROPS:-20 28 36}; do for tv in ${VOTES:-8 16 24}; do for ts in ${STAYS:-2 3 5}; do
  env MIPT_TRACE_NODE_DROP_SHADOW=$nd MIPT_TRACE_TRI_VOTE_SHADOW=$tv MIPT_TRACE_TRI_STAY_SHADOW=$ts timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/s.log 2>&1 || { echo failed; tail -3 /tmp/s.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); r=d['roofline']; print('shadow: drop=$nd vote=$tv stay=$ts frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8])"
done; done; done

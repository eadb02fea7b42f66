run() { env "$@" timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/s.log 2>&1 || { echo failed; tail -3 /tmp/s.log; return; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); r=d['roofline']; print('$*', 'frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8])"; }
run A=1
run MIPT_STACK_LDS=14 MIPT_TRACE_CACHE=608
run MIPT_STACK_LDS=12 MIPT_TRACE_CACHE=704
run MIPT_STACK_LDS=10 MIPT_TRACE_CACHE=800
run MIPT_STACK_LDS=8 MIPT_TRACE_CACHE=896
run A=2

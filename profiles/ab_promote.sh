timeout -k 10 300 python -u -m pytest tests/test_gpu_a_bvh.py tests/test_gpu_b_trace.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3
timeout -k 10 120 python -u profiles/cache_choice_probe.py 262000 | head -3
run() { env "$@" timeout -k 10 200 python -u bench.py --steps 4 --warmup 1 --no-cpu-baseline > /tmp/s.log 2>&1 || { echo failed; tail -3 /tmp/s.log; return; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); r=d['roofline']; print('$*', 'frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8], 'build ms', round(d['config']['bvh_build_ms'],2))"; }
run A=1
run MIPT_BUILD_NO_PROMOTE=1
run A=2
run MIPT_BUILD_NO_PROMOTE=1

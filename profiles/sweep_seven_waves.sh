run() { lib=$1; shift; cp $lib cuda-path-tracer_amd/csrc/libmipt.so; env "$@" timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/s.log 2>&1 || { echo failed; tail -3 /tmp/s.log; return; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); r=d['roofline']; print('$lib $*', 'frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8])"; }
cp cuda-path-tracer_amd/csrc/libmipt.so /tmp/base.so
run /tmp/base.so A=1
run profiles/ab/eu7.so A=1
run profiles/ab/eu7.so MIPT_TRACE_WAVES=14 MIPT_TRACE_CACHE=384
run profiles/ab/eu7.so MIPT_TRACE_WAVES=14 MIPT_TRACE_CACHE=256
run /tmp/base.so MIPT_TRACE_WAVES=14 MIPT_TRACE_CACHE=384
run /tmp/base.so A=2
cp /tmp/base.so cuda-path-tracer_amd/csrc/libmipt.so

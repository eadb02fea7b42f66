for cfg in "256 512" "128 512" "128 256" "192 512" "64 256"; do set -- $cfg
  MIPT_BENCH_AS_SHARD=0/8 MIPT_TRACE_CHUNK=$1 MIPT_TRACE_CHUNK_SHADOW=$2 timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/s8.log 2>&1 || { echo failed; tail -3 /tmp/s8.log; }
  python3 -c "
import json; d=json.loads(open('/tmp/s8.log').read().strip().splitlines()[-1]); r=d['roofline']; print('shard 0/8 chunk $1 shadow chunk $2 frame', round(d['ms_per_step'],2), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8])"
done

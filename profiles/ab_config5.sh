for L in tree ${C5_LIBS:-}; do
  if [ "$L" != tree ]; then cp cuda-path-tracer_amd/csrc/libmipt.so /tmp/tree.so 2>/dev/null; cp "$L" cuda-path-tracer_amd/csrc/libmipt.so; fi
  timeout -k 10 200 python -u bench.py --workload crytek38x_gi512_2160p --steps 2 --warmup 1 --no-cpu-baseline > /tmp/c5.log 2>&1 || { echo failed; tail -3 /tmp/c5.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/c5.log').read().strip().splitlines()[-1]); r=d['roofline']; print('$L', 'frame', round(d['ms_per_step'],1), 'closest', min(r['closest_ms_per_step']), 'shadow', min(r['shadow_ms_per_step']), d['config']['image_sha1'][:8], 'visits/ray', round(r['node_visits_per_s']/r['closest_rays_per_s'],2), 'tests/ray', round(r['triangle_tests_per_s']/r['closest_rays_per_s'],2))"
done

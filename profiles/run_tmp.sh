set -u
mkdir -p gpurun_out/r2e
for T in 1000 4000 16000 66000 262000 1000000; do
  timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu-baseline --tris $T > /tmp/s.log 2>&1 || { echo failed; tail -3 /tmp/s.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); r=d['roofline']; c=min(r['closest_ms_per_step'])
nv=r['node_visits_per_s']*sum(r['closest_ms_per_step'])/2e3; tt=r['triangle_tests_per_s']*sum(r['closest_ms_per_step'])/2e3; cr=r['closest_rays_per_s']*sum(r['closest_ms_per_step'])/2e3
print('tris', d['config']['triangles'], 'frame', round(d['ms_per_step'],1), 'closest', c, 'shadow', min(r['shadow_ms_per_step']), 'rays %.0fM visits/ray %.2f tests/ray %.2f  ps/visit %.2f' % (cr/1e6, nv/cr, tt/cr, c*1e9/nv))"
done 2>&1 | tee gpurun_out/r2e/size.txt

set -u
mkdir -p gpurun_out/r02a
for W in sponza_direct_1080p sponza_gi128_1080p crytek_gi128_1080p crytek_gi16_270p crytek38x_gi512_2160p; do
  S=3; [ $W = crytek38x_gi512_2160p ] && S=1
  timeout -k 10 500 python -u bench.py --steps $S --warmup 1 --no-cpu-baseline --workload $W > /tmp/s.log 2>&1 || { echo "$W failed"; tail -3 /tmp/s.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); c=d['config']
print('$W', c['triangles'], c['image'], 'ref rays %.4g culled %.4g' % (c['reference_rays_per_frame'], c['shadow_rays_not_traversed_per_frame']), 'ms %.2f' % d['ms_per_step'], 'Mrays/s %.0f' % d['value'], 'ref Mrays/s %.0f' % c['reference_mrays_per_s'], 'build ms %.1f' % c['bvh_build_ms'])" | tee -a gpurun_out/r02a/workloads.txt
done
for N in 2 4 8; do
  MIPT_BENCH_AS_SHARD=0/$N timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/s.log 2>&1 || { tail -3 /tmp/s.log; exit 1; }
  python3 -c "
import json; d=json.loads(open('/tmp/s.log').read().strip().splitlines()[-1]); print('shard 0/$N ms %.2f' % d['ms_per_step'])" | tee -a gpurun_out/r02a/workloads.txt
done

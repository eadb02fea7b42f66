"""Runs bench.py once per value of a command-line option.

    python profiles/sweep_args.py --chunk 4 8 16 32 -- --steps 2 --warmup 1
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
opt, values = args[0], args[1:]
for v in values:
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", opt, v] + extra,
                       capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        print(opt, v, "FAILED", p.stderr[-400:])
        break
    d = json.loads(p.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    print(f"{opt} {v}: {d['value']:.0f} Mrays/s, {d['ms_per_step']:.1f} ms/frame, closest {r['avg_launch_ms']:.3f} ms/launch x{r['launches']}, "
          f"share closest {r['kernel_time_share']['closest']:.3f} shadow {r['kernel_time_share']['shadow']:.3f}", flush=True)

"""Grid sweep of two environment variables over bench.py.

    python profiles/sweep2.py VAR_A a1,a2 VAR_B b1,b2 -- --steps 2 --warmup 1
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
va, la, vb, lb = args[0], args[1].split(","), args[2], args[3].split(",")
for a in la:
    row = []
    for b in lb:
        env = dict(os.environ, **{va: a, vb: b})
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + extra, env=env,
                           capture_output=True, text=True, timeout=90)
        if p.returncode != 0:
            row.append("FAIL")
            continue
        d = json.loads(p.stdout.strip().splitlines()[-1])
        row.append(f"{d['ms_per_step']:.1f}")
    print(f"{va}={a}: " + "  ".join(f"{vb}={b}:{r}" for b, r in zip(lb, row)), flush=True)

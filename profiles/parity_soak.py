"""A larger run of the traversal's parity test than the test suite affords: random, aimed and axis-parallel rays on the
headline scene (262 k triangles) against the oracle's brute-force search, ids, flags and (t, u, v) bit for bit.

    python profiles/parity_soak.py [rays per batch] [batches]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
import scenes  # noqa: E402
from mipt import abi, device, host  # noqa: E402

per, batches = (int(sys.argv[1]) if len(sys.argv) > 1 else 50000), (int(sys.argv[2]) if len(sys.argv) > 2 else 6)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0        # number of the first batch (its seed is 1000 + number)
device.init(0)
big = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 0, 1)
osc = ob.OracleScene(big.desc)
sc = device.Scene(big.desc)
total = bad = 0
t0 = time.time()
V = big.vertices()[:, :, :3].astype(np.float64)
for b in range(first, first + batches):
    rays = scenes.random_rays(big, per, seed=1000 + b, inside_fraction=0.8)
    tri, tuv, anyhit = sc.trace_rays(rays)
    otri, otuv, oany = osc.trace_rays(rays, eps=1e-20, accel=ob.ACCEL_BRUTE)
    diff = (tri != otri) | (anyhit != oany) | (tuv.view(np.uint32) != otuv.view(np.uint32)).any(axis=1)
    total += per; bad += int(diff.sum())
    for i in np.nonzero(diff)[0]:
        o, d = rays[i, :3].astype(np.float64), rays[i, 3:].astype(np.float64)
        print("  ray", i, repr(rays[i].tolist()), "gpu", tri[i], tuv[i].tolist(), int(anyhit[i]), "oracle", otri[i], otuv[i].tolist(), int(oany[i]))
        for k in {int(tri[i]), int(otri[i])} - {-1}:
            T = V[k]; e1, e2 = T[1] - T[0], T[2] - T[0]; nrm = np.cross(e1, e2); nn = nrm / np.linalg.norm(nrm)
            cos = abs(nn @ d) / np.linalg.norm(d); tt = ((T[0] - o) @ nn) / (d @ nn); P = o + tt * d
            M = np.array([[e1 @ e1, e1 @ e2], [e1 @ e2, e2 @ e2]]); ub, vb = np.linalg.solve(M, np.array([(P - T[0]) @ e1, (P - T[0]) @ e2]))
            print("    triangle", k, "cos %.6f" % cos, "t %.4f" % tt, "barycentric %.7f %.7f" % (ub, vb), "edges %.2f %.2f" % (np.linalg.norm(e1), np.linalg.norm(e2)), "corners", T.tolist())
    print(f"batch {b}: {per} rays, hits {(tri >= 0).mean():.3f}, differing {int(diff.sum())}, {time.time() - t0:.0f} s", flush=True)
print(f"total {total} rays on {big.num_triangles} triangles: {bad} differ from the brute-force search")
sys.exit(1 if bad else 0)

import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "cuda-path-tracer_amd")); sys.path.insert(0, "tests")
import numpy as np
import oracle_binding as ob
from mipt import abi, device, host
device.init(0)
import scenes
sc_h = host.HostScene.atrium(abi.SCENE_CRYTEK, 10000, 64, 1)
osc = ob.OracleScene(sc_h.desc)
sc = device.Scene(sc_h.desc)
rng = np.random.default_rng(41)
v = sc_h.vertices()[:, :, :3]
lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
size = float(np.linalg.norm(hi - lo))
for dist in (3.0, 30.0, 300.0, 3000.0):
    n = 6000
    tri = v[rng.integers(0, v.shape[0], n)]
    w = rng.random((n, 3)); w /= w.sum(axis=1, keepdims=True)
    kind = rng.integers(0, 3, n)
    target = np.einsum("nk,nkc->nc", w, tri)
    target = np.where((kind == 1)[:, None], tri[:, 0], target)
    target = np.where((kind == 2)[:, None], 0.5 * (tri[:, 0] + tri[:, 1]), target)
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o = (target - dirs * dist * size).astype(np.float32)
    d = target - o.astype(np.float64); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d.astype(np.float32)], axis=1).astype(np.float32)
    t1, tuv1, a1 = sc.trace_rays(rays)
    t2, tuv2, a2 = osc.trace_rays(rays, eps=1e-20, accel=ob.ACCEL_BRUTE)
    bad = np.nonzero(t1 != t2)[0]
    print("dist", dist, "bad closest", bad.size, "bad anyhit", int((a1 != a2).sum()))
    for i in bad[:8]:
        # geometry of the oracle's triangle in float64
        k = t2[i] if t2[i] >= 0 else t1[i]
        T = v[k].astype(np.float64); oo = rays[i, :3].astype(np.float64); dd = rays[i, 3:].astype(np.float64)
        nrm = np.cross(T[1] - T[0], T[2] - T[0]); nn = nrm / np.linalg.norm(nrm)
        cos = abs(nn @ dd) / np.linalg.norm(dd)
        tt = ((T[0] - oo) @ nn) / (dd @ nn); P = oo + tt * dd
        # barycentrics
        e1, e2 = T[1] - T[0], T[2] - T[0]; M = np.array([[e1 @ e1, e1 @ e2], [e1 @ e2, e2 @ e2]]); rhs = np.array([(P - T[0]) @ e1, (P - T[0]) @ e2])
        ub, vb = np.linalg.solve(M, rhs)
        print("  ray", i, "gpu", t1[i], "oracle", t2[i], "cos %.4f" % cos, "bary %.5f %.5f" % (ub, vb), "edge %.1f" % np.linalg.norm(e1), "t %.1f" % tt, "tuv gpu", tuv1[i], "oracle", tuv2[i])

"""Which node slots should the traversal's LDS copy hold?  Compares the share of the surface area (the share of
node visits of random rays) of the first 512 slots, which the copy holds, with that of the 512 slots of largest area."""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "cuda-path-tracer_amd"))
import numpy as np
from mipt import abi, device, host
device.init(0)
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 262000
sc = host.HostScene.atrium(abi.SCENE_CRYTEK, tris, 64, 1)
dev = device.Scene(sc.desc)
nodes, v0, e1, e2, root, grid = dev.download_bvh()
codes = nodes[:, 12:16]
n = nodes.shape[0]
w = nodes[:, :12].reshape(-1, 4, 3).astype(np.int64)
valid = codes != 0xFFFFFFFF
lo = np.where(valid[..., None], w & 0xFFFF, 1 << 20).min(axis=1).astype(np.float64) * grid[1]
hi = np.where(valid[..., None], w >> 16, -1).max(axis=1).astype(np.float64) * grid[1]
d = np.maximum(hi - lo, 0)
area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
area[~valid.any(axis=1)] = 0
tot = area.sum()
order = np.argsort(-area)
for k in (256, 512, 768, 1024):
    print("slots", k, "first-k share %.3f" % (area[:k].sum() / tot), "best-k share %.3f" % (area[order[:k]].sum() / tot),
          "best-k slots beyond k: %d" % int((order[:k] >= k).sum()))

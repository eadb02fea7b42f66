"""Fill of the four-wide BVH nodes of the headline scene: children per node, triangles per leaf."""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "cuda-path-tracer_amd"))
import numpy as np
from mipt import abi, device, host
device.init(0)
sc = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 64, 1)
dev = device.Scene(sc.desc)
inf = dev.info()
nodes, v0, e1, e2, root, grid = dev.download_bvh()
codes = nodes[:, 12:16]
# walk from the root: slots that no node points to are padding of the line pairs
reach = np.zeros(nodes.shape[0], dtype=bool)
front = np.array([root], dtype=np.int64)
hist = np.zeros(5, dtype=np.int64); leaves = 0; depth = 0
while front.size:
    reach[front] = True
    c = codes[front]
    valid = c != 0xFFFFFFFF
    hist += np.bincount(valid.sum(axis=1), minlength=5)
    leaf = valid & (c >= nodes.shape[0])      # codes from the number of node slots on are leaf slots
    leaves += int(leaf.sum())
    front = c[valid & ~leaf].astype(np.int64)
    depth += 1
print("node slots", nodes.shape[0], "reached", int(reach.sum()), "levels", depth, "leaf children", leaves, "tris", inf.num_triangles,
      "tris per leaf %.2f" % (inf.num_triangles / max(leaves, 1)))
print("children per node histogram (0..4):", hist, "mean %.2f" % ((hist * np.arange(5)).sum() / hist.sum()))

# the same histogram weighted by the surface area of the node (union of its child boxes): the share of
# node visits of random rays that each kind of node receives
w = nodes[reach][:, :12].reshape(-1, 4, 3).astype(np.int64)       # word = min | max << 16 per child and axis
c = codes[reach]
valid = c != 0xFFFFFFFF
lo = np.where(valid[..., None], w & 0xFFFF, 1 << 20).min(axis=1).astype(np.float64) * grid[1]
hi = np.where(valid[..., None], w >> 16, -1).max(axis=1).astype(np.float64) * grid[1]
d = np.maximum(hi - lo, 0)
area = d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]
cnt = valid.sum(axis=1)
wh = np.bincount(cnt, weights=area, minlength=5)
print("area-weighted share by children per node (0..4):", np.round(wh / wh.sum(), 3), "mean %.2f" % ((wh * np.arange(5)).sum() / wh.sum()))

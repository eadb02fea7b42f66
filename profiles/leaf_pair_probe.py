"""How many of the triangle pairs that an owner lane and its helper test together share an edge?
Such a pair has four distinct corners (48 bytes) where two records hold six (and cost six per-lane accesses)."""
import sys, os
sys.path.insert(0, os.path.join(os.getcwd(), "cuda-path-tracer_amd"))
import numpy as np
from mipt import abi, device, host
device.init(0)
sc = host.HostScene.atrium(abi.SCENE_CRYTEK, 262000, 64, 1)
dev = device.Scene(sc.desc)
nodes, v0, e1, e2, root, grid = dev.download_bvh()
ids = v0[:, 3].copy().view(np.uint32).astype(np.int64)
last = e1[:, 3].copy().view(np.uint32)
verts = sc.vertices()[:, :, :3][ids]                      # corners in BVH order
n = verts.shape[0]
first = np.concatenate([[True], last[:-1] == 1])          # first triangle of a leaf
pos_in_leaf = np.arange(n) - np.maximum.accumulate(np.where(first, np.arange(n), 0))
pair_start = (pos_in_leaf % 2 == 0) & (last == 0)         # triangles 0 and 2 of a leaf that have a successor in it
a = verts[pair_start]; b = verts[np.nonzero(pair_start)[0] + 1]
shared = np.zeros(a.shape[0], dtype=np.int64)
for i in range(3):
    for j in range(3):
        shared += (a[:, i] == b[:, j]).all(axis=1)
leaves = int(first.sum())
print("triangles", n, "leaves", leaves, "triangles per leaf %.2f" % (n / leaves))
print("pairs tested together", a.shape[0], "of them sharing an edge (two corners): %.3f" % (shared >= 2).mean(), "one corner: %.3f" % (shared == 1).mean())
single = int((pos_in_leaf % 2 == 0).sum() - a.shape[0])
print("triangles tested alone (last of an odd leaf)", single)
acc_now = 3 * n
acc_shared = 3 * n - 2 * int((shared >= 2).sum())
print("per-lane accesses for one test of every triangle: %d now, %d with four-corner records for such pairs (%.1f %% fewer)" % (acc_now, acc_shared, 100 * (1 - acc_shared / acc_now)))

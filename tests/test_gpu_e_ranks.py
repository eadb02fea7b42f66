"""The N-rank flow of bench.py as real processes under a process group: every rank is a fresh
interpreter that creates the scene, traces its shard with mipt_frame_accumulate, and the shares meet
in one torch.distributed reduce on rank 0, which finalizes.  One GPU is at hand here, so the ranks
share it and the sums travel through host memory (gloo); on a node with one GPU per rank the only
difference is the backend (RCCL)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

BENCH = os.path.join(ROOT, "bench.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(ranks, shard, dump, extra_env=None):
    args = ["--gpus", str(ranks), "--steps", "1", "--warmup", "0", "--workload", "crytek_gi16_270p", "--no-cpu-baseline",
            "--shard", shard, "--dump-image", str(dump)]
    if ranks == 1:
        cmd = [sys.executable, BENCH] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH] + args
    env = dict(os.environ, MIPT_BENCH_BACKEND="gloo", OMP_NUM_THREADS="4")
    env.update(extra_env or {})
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # rank 0 alone prints the result
    return json.loads(lines[0]), np.load(dump)


def test_two_ranks_render_the_one_rank_image(tmp_path):
    one, img1 = _bench(1, "tiles", tmp_path / "one.npy")
    assert one["n_gpus"] == 1 and img1.shape == (270, 480, 3) and img1.max() > 10
    tiles, img_t = _bench(2, "tiles", tmp_path / "tiles.npy")
    assert tiles["n_gpus"] == 2 and tiles["config"]["parallelism"] == "tilesx2"
    # every pixel is traced, summed and clamped by exactly one rank: the frames are identical
    assert tiles["config"]["image_sha1"] == one["config"]["image_sha1"] and np.array_equal(img_t, img1)
    assert tiles["config"]["rays_per_frame"] == one["config"]["rays_per_frame"]            # no ray traced twice
    samples, img_s = _bench(2, "samples", tmp_path / "samples.npy")
    assert samples["n_gpus"] == 2 and samples["config"]["parallelism"] == "samplesx2"
    # a pixel's samples come from two ranks: equal up to the fp32 order of the sum
    assert np.allclose(img_s, img1, rtol=1e-5, atol=1e-3)
    assert (img_s == img1).mean() > 0.5


def test_one_rank_reduces_through_rccl(tmp_path):
    """The collective path of a multi-GPU run on the one GPU at hand: a one-rank `nccl` (= RCCL) process group is
    created after the device has been chosen, the accumulator (sample shards) or the final image (tile shards) goes
    through torch.distributed.reduce on the GPU, and the library's own stream then reads what the collective's
    stream wrote.  The image must be the one of the plain run."""
    plain, img0 = _bench(1, "tiles", tmp_path / "plain.npy")
    for shard in ("tiles", "samples"):
        env = {"MIPT_BENCH_BACKEND": "nccl", "MIPT_BENCH_FORCE_PG": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
               "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}
        res, img = _bench(1, shard, tmp_path / f"rccl_{shard}.npy", env)
        assert res["config"]["process_group"] == "ncclx1"
        assert res["config"]["image_sha1"] == plain["config"]["image_sha1"] and np.array_equal(img, img0)

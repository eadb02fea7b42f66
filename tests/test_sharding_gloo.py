"""The multi-GPU data path with two ranks on CPU: each rank produces its share of the per-pixel
first-bounce sums (here with the CPU oracle standing in for the HIP renderer), the shares are
reduced to rank 0 with torch.distributed (gloo here, RCCL on GPUs), rank 0 finalizes.  Tile
shards must reproduce the single-rank image exactly, sample-range shards to rounding."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, mode, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "..", "cuda-path-tracer_amd"))
    import oracle_binding as ob
    import scenes
    from conftest import CORNELL
    from mipt import host, sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = host.HostScene.load_obj(CORNELL, True)
    osc = ob.OracleScene(scene.desc)
    cam, sun = scenes.cornell_camera(div=16)
    N = 6
    fields = sharding.shard_fields(rank, world, N, mode, tile_size=8)
    prm = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N, **fields)
    accum = torch.from_numpy(osc.accumulate(cam, prm, threads=2))
    owned = int((accum.abs().sum(dim=2) > 0).sum())
    sharding.reduce_to_rank0(accum, dist, world)
    counts = torch.tensor([owned], dtype=torch.int64)
    dist.all_reduce(counts)
    if rank == 0:
        full = host.default_params(sun_pos=sun, global_illumination=1, num_samples=N)
        img = osc.finalize(cam, full, accum.numpy(), threads=2)
        ref, _, _, _ = osc.render(cam, full, threads=2)
        np.savez(os.path.join(outdir, f"{mode}.npz"), img=img, ref=ref, owned=owned, total=int(counts.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tiles", "samples"])
def test_two_rank_reduce(tmp_path, mode):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    r = np.load(tmp_path / f"{mode}.npz")
    if mode == "tiles":
        assert np.array_equal(r["img"], r["ref"])
        assert 0 < r["owned"] < r["total"]              # both ranks contributed pixels
    else:
        assert np.allclose(r["img"], r["ref"], rtol=1e-5, atol=1e-3)

"""Partition of one frame over the GPUs of a node and the reduce that puts it back together.

Pixels are dealt in square tiles round-robin (tile t belongs to rank t % world), or the
first-bounce sample loop is cut into contiguous ranges.  Every rank produces the unclamped
per-pixel sum of r1 * L1 / pdf (PathTracer.cu:449-451) for its share and zeros elsewhere; the
shares are added on rank 0, which alone applies the final division, clamp and *255
(PathTracer.cu:453-462, 550-552).  The clamp is not linear, so it must come after the sum."""


def shard_fields(rank, world, num_samples, mode="tiles", tile_size=32, own_pixels_only=False):
    """mipt_params fields for this rank.  own_pixels_only (tile mode): every rank traces and finalizes
    its own tiles only and the FINAL images are added, instead of the unclamped sums."""
    if world <= 1:
        return dict(shard_rank=0, shard_count=1, tile_size=tile_size, sample_begin=0, sample_end=0)
    if mode == "tiles":
        f = dict(shard_rank=rank, shard_count=world, tile_size=tile_size, sample_begin=0, sample_end=0)
        if own_pixels_only:
            f["own_pixels_only"] = 1
        return f
    if mode == "samples":
        per = (num_samples + world - 1) // world
        begin = min(num_samples, rank * per)
        end = min(num_samples, (rank + 1) * per)
        if begin == 0 and end == 0:      # 0,0 means "all samples" at the C ABI: an empty share is [n, n)
            begin = end = num_samples
        return dict(shard_rank=0, shard_count=1, tile_size=tile_size, sample_begin=begin, sample_end=end)
    raise ValueError(f"unknown shard mode {mode!r}")


def tile_owner(x, y, width, world, tile_size=32):
    """Rank that traces pixel (x, y); mirrors pixel_owned() of csrc/pt_kernels.hip."""
    if world <= 1:
        return 0
    tiles_x = (width + tile_size - 1) // tile_size
    return ((y // tile_size) * tiles_x + (x // tile_size)) % world


def reduce_to_rank0(accum, dist=None, world=1, force=False):
    """Sums the per-rank accumulators (a torch tensor, on the GPU under RCCL or on the CPU under
    gloo) into rank 0's.  One collective per frame.  force: run the collective with one rank too."""
    if world > 1 or (force and dist is not None):
        dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
        if getattr(accum, "is_cuda", False):
            # the collective runs on the communicator's stream and the render library on its own:
            # the host waits here so that the finalize kernel cannot start before the sum is complete
            import torch
            torch.cuda.current_stream().synchronize()
    return accum

"""Batch sharding of the LceBconv2d path across the GPUs of one node.

Every output pixel depends only on its own image (core/bconv2d/reference.h:84 iterates the
batch in the outermost loop; indirect_bgemm/kernel.h:46-47 folds it into the pixel count),
so the batch splits into contiguous NHWC slabs with NO data-path collective: weights and
folded parameters are replicated, rank r computes images [start, start+count).  The
process group (RCCL over xGMI on the GPU node, gloo in the CPU tests) is used only for
  * the barrier around timed regions and the MAX-over-ranks of the elapsed time, and
  * an optional all-gather of outputs when a consumer needs the whole batch on every rank
    (never part of the layer timing: a float L0 shard is 822 MB, i.e. >= 5 ms on a
    ~153 GB/s xGMI link versus < 1 ms of compute).
"""
from __future__ import annotations


def shard_range(global_batch: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous split; the first (global_batch % world_size) ranks take one extra image."""
    if world_size < 1 or not 0 <= rank < world_size or global_batch < 0:
        raise ValueError((global_batch, world_size, rank))
    base, extra = divmod(global_batch, world_size)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def max_over_ranks(seconds: float, dist=None, device=None) -> float:
    """MAX-reduce a rank-local elapsed time (the contract of bench.py)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(seconds: float, dist=None, device=None) -> list:
    """Every rank's elapsed time, in rank order (bench.py reports them next to the MAX)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [seconds]
    import torch
    mine = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    return [float(p.item()) for p in parts]


def all_gather_batch(local, global_batch: int, dist):
    """Reassemble the full-batch tensor on every rank from contiguous shards (shards may
    differ by one image, so they are padded to the largest shard for the collective)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_range(global_batch, world, r)[1] for r in range(world)]
    biggest = max(counts)
    pad = torch.zeros((biggest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)

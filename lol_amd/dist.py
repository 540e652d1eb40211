"""lol_amd/dist.py — batch sharding across the GPUs of one node.

Every polynomial of a batch is independent (no operation of the Tensor interface couples
two batch items), so the data path shards by contiguous batch ranges with
NO collective.  The only collective is optional: an all-gather of result shards when a
caller wants the whole batch on every rank (RCCL over xGMI when the backend is "nccl";
the same code runs on "gloo" for CPU tests).  The reference has no distributed code at all.
"""
from __future__ import annotations


def shard_range(batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of a batch owned by `rank`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world) or batch < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(batch: int, world: int) -> list[int]:
    return [shard_range(batch, r, world)[1] - shard_range(batch, r, world)[0] for r in range(world)]


def allgather_batch(local, batch: int, group=None):
    """Gather ragged shards [b_r, n, T] (shard_range order) into the full [batch, n, T] tensor
    on every rank.  One collective per call; shards are padded to the largest shard so a
    single fixed-size all_gather (RCCL's fast path) suffices."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = shard_sizes(batch, world)
    if local.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local shard has the wrong batch size")
    mx = max(sizes) if sizes else 0
    if sizes and min(sizes) == mx:          # equal shards (the usual case): no padding copy, no concatenation
        out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * mx: r * mx + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)

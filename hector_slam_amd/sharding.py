"""Multi-GPU data parallelism for batched matching: one process per GPU, the pyramid replicated,
the batch of independent (pose hypothesis, scan) pairs split contiguously across ranks, and ONE
collective per batched match -- an all-gather of the [B/G, 3] fp32 poses (optionally the
[B/G, 9] Hessians).  With backend "nccl" this is RCCL over xGMI; the payload is tens of KiB per
rank, i.e. latency-bound, so a single un-bucketed all-gather is the right shape.
The reference has no distributed path at all; nothing here translates reference code.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) of rank's share; the first total % world ranks get one extra."""
    base, rem = divmod(total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def max_shard(total: int, world: int) -> int:
    return (total + world - 1) // world


def all_gather_rows(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """Gather row-sharded results (shard_bounds layout) from every rank into [total, C] on each rank.

    Ranks may hold uneven shards: rows are padded to max_shard so a single
    all_gather_into_tensor (one RCCL call) suffices, then the padding is dropped.
    """
    if not (dist.is_available() and dist.is_initialized()):
        assert local.shape[0] == total
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    b, e = shard_bounds(total, rank, world)
    assert local.shape[0] == e - b, "local shard does not match shard_bounds"
    m = max_shard(total, world)
    cols = local.shape[1]
    if local.shape[0] == m:
        send = local.contiguous()
    else:
        send = torch.zeros((m, cols), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    out = torch.empty((world * m, cols), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if total == world * m:
        return out
    parts = []
    for r in range(world):
        rb, re_ = shard_bounds(total, r, world)
        parts.append(out[r * m: r * m + (re_ - rb)])
    return torch.cat(parts, 0)

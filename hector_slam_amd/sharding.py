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


class AsyncRowGather:
    """Overlapped all-gather of equally sized per-rank result rows (the [B/G, 3] poses of a batched match).

    The matcher of batch k+1 must not wait for the collective of batch k: results are written into one of
    ``depth`` local buffers, the all-gather of that buffer is enqueued asynchronously (RCCL runs it on its own
    stream, ordered after the compute stream's work at enqueue time), and the compute stream only waits for a
    collective when its buffer comes up for reuse ``depth`` batches later.  With no process group initialised
    it degenerates to handing the local buffer back.
    """

    def __init__(self, rows_per_rank: int, cols: int, device, dtype=torch.float32, depth: int = 2, group=None):
        self.group = group
        self.dist = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist else 1
        self.depth = depth
        self.local = [torch.zeros((rows_per_rank, cols), dtype=dtype, device=device) for _ in range(depth)]
        self.out = [torch.empty((self.world * rows_per_rank, cols), dtype=dtype, device=device) if self.dist else None
                    for _ in range(depth)]
        self.work = [None] * depth
        self.k = 0

    def next_local(self) -> torch.Tensor:
        """buffer the next batch's results go into (waits, on the stream, for the collective that last used it)"""
        slot = self.k % self.depth
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        return self.local[slot]

    def launch(self) -> int:
        """enqueue the all-gather of the buffer handed out by the last next_local(); returns its slot"""
        slot = self.k % self.depth
        if self.dist:
            self.work[slot] = dist.all_gather_into_tensor(self.out[slot], self.local[slot], group=self.group,
                                                          async_op=True)
        self.k += 1
        return slot

    def result(self, slot: int) -> torch.Tensor:
        """gathered [world * rows, cols] tensor of ``slot`` (waits for its collective)"""
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        return self.out[slot] if self.dist else self.local[slot]

    def wait_all(self) -> None:
        for s in range(self.depth):
            if self.work[s] is not None:
                self.work[s].wait()
                self.work[s] = None

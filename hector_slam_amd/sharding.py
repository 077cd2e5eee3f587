"""Multi-GPU data parallelism for batched matching: one process per GPU, the pyramid replicated,
the batch of independent (pose hypothesis, scan) pairs split contiguously across ranks, and ONE exchange step --
every rank ends up with every rank's [B/G, 3] fp32 poses (optionally the [B/G, 9] Hessians) -- per batched match.

Transports of that step:
  DirectRowGather    (round 6, what bench.py --gpus N times) the library's device-side exchange (hsm_exchange_*,
                     csrc/pose_exchange.h): every rank stores its rows straight into every rank's IPC-mapped mailbox over
                     xGMI, one small kernel per match on the matcher's stream; torch.distributed only carries the 64-byte
                     IPC handles once, at set-up.
  AsyncRowGather     one torch.distributed all-gather per match (backend "nccl" = RCCL); 45 us of host time per enqueue
                     and an RCCL kernel beside the matcher launch (103 us per step against 58.5: profiles/r05/README.md 7)
  BucketedRowGather  the same collective for a bucket of consecutive matches: amortises the above, but is not a gather
                     per match; kept as the labelled comparison figure.
The reference has no distributed path at all; nothing here translates reference code.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [begin, end) of rank's share; the first total % world ranks get one extra.  The same closed form as the
    native library's hsm_shard_bounds (tests/test_sharding_cpu.py::test_shard_bounds_is_the_native_rule holds the two
    equal), so the single-process group (hsm_group_match_batch) and this process-per-GPU path split a batch identically and
    results gathered by either transport line up row for row -- stated here in Python so that an index computation (and
    the CPU / gloo path built on it) needs neither a HIP toolchain nor a built library."""
    if total < 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError(f"shard_bounds: bad argument (total={total}, rank={rank}, world={world})")
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


class DirectRowGather:
    """ONE gather per batched match without a collective: the device-side exchange of the native library.

    Set-up (once): every rank creates its mailbox (capi.PoseExchange), the 64-byte IPC handles travel over the process group
    (all_gather_object -- any backend), every rank maps the others' mailboxes.  Per match, on the matcher's stream:
    ``launch()`` = ONE kernel that posts this rank's rows of batch k to every rank and -- with ``lag`` = 1 -- unpacks batch
    k-1's gathered rows, which have been travelling behind a whole matcher launch; ``drain()`` waits for what is still in
    flight.  Rows of rank r sit at ``shard_bounds(total_rows, r, world)`` of the gathered [total_rows, cols] array.
    The mailbox holds 2 + 2 lag buffers: no acknowledgements are needed (csrc/pose_exchange.h).
    Without a process group it is a one-rank exchange (same kernels, same protocol)."""

    def __init__(self, total_rows: int, cols: int, device, lag: int = 1, group=None, dtype=torch.float32):
        from . import capi
        assert dtype == torch.float32, "the exchange carries fp32 rows"
        self.group = group
        self.dist = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rank = dist.get_rank(group) if self.dist else 0
        self.total_rows, self.cols, self.lag = total_rows, cols, lag
        self.first_row, end = shard_bounds(total_rows, self.rank, self.world)
        self.rows = end - self.first_row
        dev = torch.device(device)
        # Set-up is collective: every rank takes part in BOTH object gathers whatever happened to it locally -- a rank whose mailbox
        # could not be created or connected says so instead of leaving the others inside a collective it never joins -- and a
        # failure anywhere is raised on every rank (make_row_gather then falls back on all of them together).
        self.x, err = None, None
        try:
            self.x = capi.PoseExchange(self.rank, self.world, total_rows, cols, depth=2 + 2 * lag, device=dev.index if dev.index is not None else -1)
        except Exception as exc:
            err = f"{type(exc).__name__}: {str(exc)[:160]}"
        if self.world > 1:
            handles = [None] * self.world
            dist.all_gather_object(handles, (self.x.handle() if self.x is not None else None, err), group=group)
            bad = [(r, e) for r, (hd, e) in enumerate(handles) if hd is None]
            if not bad:
                try:
                    self.x.connect([hd for hd, _ in handles])
                except Exception as exc:
                    err = f"{type(exc).__name__}: {str(exc)[:160]}"
            oks = [None] * self.world
            dist.all_gather_object(oks, err if not bad else (err or "a peer has no mailbox"), group=group)
            bad = bad or [(r, e) for r, e in enumerate(oks) if e]
            if bad:
                if self.x is not None:
                    self.x.close()
                    self.x = None
                raise RuntimeError("device-side exchange set-up failed: " + "; ".join(f"rank {r}: {e}" for r, e in bad[:3]))
        elif err:
            raise RuntimeError("device-side exchange set-up failed: " + err)
        self.local = torch.zeros((max(self.rows, 1), cols), dtype=dtype, device=dev)[: self.rows]
        self.out = [torch.zeros((total_rows, cols), dtype=dtype, device=dev) for _ in range(2)]
        self.launched = 0   # epochs posted
        self.landed = 0     # epochs unpacked (stream order)
        self.collectives = 0  # torch.distributed collectives on the data path: stays 0

    def next_local(self) -> torch.Tensor:
        """[rows, cols] view the next batch's results go into (one buffer: the post that reads it precedes the next matcher
        launch in stream order)"""
        return self.local

    def launch(self, stream=None) -> None:
        """the batch written into next_local() has been queued on ``stream``: post it, unpack the batch ``lag`` matches back"""
        s = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        e = self.launched + 1
        w = e - self.lag
        lands = w >= 1 and w > self.landed  # (not after a drain: that batch has been unpacked already)
        self.x.post_wait(self.local.data_ptr(), self.first_row, self.rows, self.lag, self.out[w % 2].data_ptr() if lands else 0, s)
        self.launched = e
        if lands:
            self.landed = w

    def match_and_launch(self, matcher, batch, d_begin, d_pts, d_offsets, shared_n, d_out_cov=0, stream=None) -> None:
        """the matcher launch of this batch AND its exchange step in one call (hsm_match_batch_device_gather): the poses go into
        next_local(); where the matcher form can, the launch posts them itself and unpacks the batch `lag` back in its tail"""
        s = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        e = self.launched + 1
        w = e - self.lag
        lands = w >= 1 and w > self.landed
        matcher.match_batch_device_gather(batch, d_begin, d_pts, d_offsets, shared_n, self.local.data_ptr(), d_out_cov, self.x, self.first_row,
                                          self.lag, self.out[w % 2].data_ptr() if lands else 0, s)
        self.launched = e
        if lands:
            self.landed = w

    def drain(self, stream=None) -> None:
        """wait (on the stream) for every posted batch"""
        s = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        while self.landed < self.launched:
            self.landed += 1
            self.x.wait(self.out[self.landed % 2].data_ptr(), s)

    flush = drain

    def wait_all(self) -> None:
        pass  # nothing outside the stream: synchronising the stream is the wait

    def last_result(self) -> torch.Tensor:
        """[total_rows, cols]: every rank's rows of the most recent batch (drains; valid once the stream has been synchronised)"""
        if self.launched == 0:
            raise RuntimeError("DirectRowGather.last_result: no batch yet")
        self.drain()
        return self.out[self.landed % 2]

    def check(self) -> None:
        """raises if a wait timed out (after synchronising the stream)"""
        self.x.check()

    def close(self) -> None:
        if self.x is not None:
            self.x.close()


def make_row_gather(total_rows: int, rows_per_rank: int, cols: int, device, lag: int = 1, fallback_bucket: int = 1, group=None):
    """The gather a sharded run should use on THIS machine: the device-side exchange (DirectRowGather) if every rank can set it up
    AND a self-test epoch of known rows arrives intact on every rank; else a torch.distributed all-gather (BucketedRowGather) --
    decided collectively, so that all ranks take the same transport.  Returns (gatherer, "direct" | "rccl", note).  The self-test is
    what stands between a topology this code has not seen (IPC mapping of uncached memory refused, no peer path between two
    devices, stores that do not become visible to a polling kernel) and a run that hangs or gathers garbage: such a machine gets the
    collective, and the note says why."""
    ok, note, g = True, "", None
    try:
        g = DirectRowGather(total_rows, cols, device, lag=lag, group=group)
        probe = torch.arange(g.rows * cols, dtype=torch.float32, device=g.local.device).reshape(g.rows, cols) + 1000.0 * (g.rank + 1)
        g.next_local().copy_(probe)
        g.launch()
        g.drain()
        torch.cuda.synchronize()
        g.check()
        got = g.out[g.landed % 2]
        for r in range(g.world):
            b, e = shard_bounds(total_rows, r, g.world)
            want = torch.arange((e - b) * cols, dtype=torch.float32, device=got.device).reshape(e - b, cols) + 1000.0 * (r + 1)
            if not torch.equal(got[b:e], want):
                ok, note = False, f"self-test: the rows of rank {r} did not arrive intact on rank {g.rank}"
                break
    except Exception as exc:  # set-up or self-test failed on this rank
        ok, note = False, f"{type(exc).__name__}: {str(exc)[:200]}"
    if dist.is_available() and dist.is_initialized():
        flags = [None] * dist.get_world_size(group)
        dist.all_gather_object(flags, (ok, note), group=group)
        bad = [(r, n) for r, (o, n) in enumerate(flags) if not o]
        if bad:
            ok, note = False, "; ".join(f"rank {r}: {n}" for r, n in bad[:3])
    if ok:
        return g, "direct", ""
    if g is not None:
        try:
            g.close()
        except Exception:
            pass
    return BucketedRowGather(rows_per_rank, cols, device, bucket=fallback_bucket, group=group), "rccl", "device-side exchange unavailable (" + note + ")"


class BucketedRowGather:
    """The same gather, BUCKETED: the rows of ``bucket`` consecutive batches travel in ONE all-gather.

    Measured on MI355X with the real "nccl" backend (bench.py, HSM_BENCH_FORCE_DIST=1, profiles/r05/README.md 7): enqueueing one
    asynchronous torch.distributed all-gather costs the host ~45 us -- as much as the 58 us matcher launch it is supposed to hide
    behind -- and the RCCL kernel, running beside a matcher launch, takes CUs from that launch's single generation of workgroups
    (+45 us for it).  A loop that gathers after EVERY batched match runs at 103 us per step instead of 58.5 (238 M instead of 418 M GN
    it/s per GPU); buckets of 8: 65.5 us; one collective behind the last match of a stream of batches: 62.0 us (59.6 with no gather at
    all).  The payload is tiny (48 KiB per rank and batch), so nothing is lost by sending ``bucket`` batches at once: the poses of
    batch k go into slot k % bucket of a [bucket, rows, cols] block, a full block is all-gathered asynchronously (RCCL's own stream,
    ordered behind the compute stream at enqueue time) while the next block fills; ``depth`` blocks rotate.  ``flush()`` sends a
    partially filled block (end of a stream of batches).  Every rank ends up with every rank's rows of every batch.
    With no process group initialised it degenerates to handing local buffers back.
    """

    def __init__(self, rows_per_rank: int, cols: int, device, dtype=torch.float32, bucket: int = 8, depth: int = 2, group=None):
        self.group = group
        self.dist = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.dist else 1
        self.rows, self.cols, self.bucket, self.depth = rows_per_rank, cols, max(1, bucket), max(2, depth)
        self.local = [torch.zeros((self.bucket, rows_per_rank, cols), dtype=dtype, device=device) for _ in range(self.depth)]
        self.out = [torch.empty((self.world, self.bucket, rows_per_rank, cols), dtype=dtype, device=device) if self.dist else None
                    for _ in range(self.depth)]
        self.work = [None] * self.depth
        self.block = 0      # block being filled
        self.fill = 0       # batches already in it
        self.collectives = 0
        self.last = None    # (block, slot) of the most recent batch

    def next_local(self) -> torch.Tensor:
        """[rows, cols] view the next batch's results go into (entering a block waits, on the stream, for the collective that last
        used it)"""
        if self.fill == 0 and self.work[self.block] is not None:
            self.work[self.block].wait()
            self.work[self.block] = None
        return self.local[self.block][self.fill]

    def launch(self) -> None:
        """the batch handed out by the last next_local() has been queued on the compute stream; a full block is sent"""
        self.last = (self.block, self.fill)
        self.fill += 1
        if self.fill == self.bucket:
            self.flush()

    def flush(self) -> None:
        """send the block being filled (all ``bucket`` slots travel; the unfilled ones carry stale rows nobody reads)"""
        if self.fill == 0:
            return
        if self.dist:
            self.work[self.block] = dist.all_gather_into_tensor(self.out[self.block].view(-1, self.cols),
                                                                self.local[self.block].view(-1, self.cols), group=self.group, async_op=True)
            self.collectives += 1
        self.block = (self.block + 1) % self.depth
        self.fill = 0

    def wait_all(self) -> None:
        for b in range(self.depth):
            if self.work[b] is not None:
                self.work[b].wait()
                self.work[b] = None

    def last_result(self) -> torch.Tensor:
        """[world * rows, cols]: every rank's rows of the most recent batch (flushes and waits for its block)"""
        if self.last is None:
            raise RuntimeError("BucketedRowGather.last_result: no batch yet")
        b, slot = self.last
        if b == self.block and self.fill > 0:
            self.flush()
        if self.work[b] is not None:
            self.work[b].wait()
            self.work[b] = None
        if not self.dist:
            return self.local[b][slot]
        return self.out[b][:, slot].reshape(self.world * self.rows, self.cols)


class ReplicaSync:
    """Replicated-map protocol of BASELINE configs[4] (one dense scan at a time does not shard): every rank holds a
    replica of the pyramid; rank ``src`` runs matchData, then ONE broadcast carries the matched pose and the scan
    ([x, y, theta, n] + n endpoints, fp32) to every rank, and every rank replays the deterministic updateByScan on its
    replica -- identical inputs, bit-identical maps.  ``digests_equal`` compares per-level map digests across ranks.
    Works with any backend (nccl = RCCL on GPUs, gloo in the CPU tests); without a process group it is a no-op.
    """

    def __init__(self, max_beams: int, device, group=None):
        self.group = group
        self.dist = dist.is_available() and dist.is_initialized()
        self.device = device
        self.buf = torch.zeros(4 + 2 * max_beams, dtype=torch.float32, device=device)
        self.max_beams = max_beams

    def broadcast(self, pose, pts, src: int = 0):
        """rank ``src`` passes (pose[3], pts[n, 2]) numpy arrays, the others pass (None, None); returns both on every rank"""
        import numpy as np
        rank = dist.get_rank(self.group) if self.dist else 0
        if not self.dist:
            return np.asarray(pose, np.float32), np.asarray(pts, np.float32).reshape(-1, 2)
        if rank == src:
            p = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
            n = p.shape[0]
            # an oversize scan must not leave the other ranks blocked in the collective: it travels as the count -1 and every
            # rank raises the same error after the broadcast
            bad = n > self.max_beams
            if bad:
                p, n = p[:0], 0
            host = np.empty(4 + 2 * n, np.float32)
            host[:3] = np.asarray(pose, np.float32)
            host[3:4] = np.array([-1 if bad else n], np.int32).view(np.float32)  # the count travels as raw bits
            host[4:] = p.reshape(-1)
            self.buf[: 4 + 2 * n].copy_(torch.from_numpy(host))
        dist.broadcast(self.buf, src=src, group=self.group)  # fixed-size payload: one collective, no size exchange
        head = self.buf[:4].cpu().numpy()
        n = int(head[3:4].view(np.int32)[0])
        if not 0 <= n <= self.max_beams:
            raise ValueError(f"ReplicaSync.broadcast: scan of the source rank does not fit the {self.max_beams}-beam buffer (count {n})")
        body = self.buf[4: 4 + 2 * n].cpu().numpy().reshape(n, 2).copy()
        return head[:3].copy(), body

    def digests_equal(self, digest) -> bool:
        """digest: 1-D int64 tensor/array (e.g. one checksum per level); True iff all ranks hold the same values"""
        import numpy as np
        d = torch.as_tensor(np.asarray(digest, np.int64)).to(self.device)
        if not self.dist:
            return True
        world = dist.get_world_size(self.group)
        out = torch.empty((world, d.numel()), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(out, d.reshape(1, -1).contiguous(), group=self.group)
        return bool((out == out[0:1]).all().item())


def map_digest(logodds, update_index) -> int:
    """order-sensitive 63-bit digest of one level (log-odds bits and update stamps), host arrays"""
    import hashlib

    import numpy as np
    h = hashlib.blake2b(digest_size=8)
    h.update(np.ascontiguousarray(logodds, np.float32).tobytes())
    h.update(np.ascontiguousarray(update_index, np.int32).tobytes())
    return int.from_bytes(h.digest(), "little") >> 1

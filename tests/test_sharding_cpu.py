"""N>1 path on CPU: world_size-2 gloo processes shard a batch with shard_bounds and reassemble
it with the single all-gather the GPU path uses (backend nccl = RCCL there)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    from hector_slam_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(total * 3, dtype=torch.float32).reshape(total, 3) * 0.5 - 7.0
        b, e = sharding.shard_bounds(total, rank, world)
        local = full[b:e].clone() + 0.0  # stand-in for the poses this rank's GPU matched
        got = sharding.all_gather_rows(local, total)
        ok = bool(torch.equal(got, full))
        q.put((rank, ok, b, e))
    finally:
        dist.destroy_process_group()


def _worker_async(rank, world, port, rows, q):
    sys.path.insert(0, ROOT)
    from hector_slam_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.AsyncRowGather(rows, 3, "cpu", depth=2)
        ok = True
        slots = []
        for k in range(7):  # more batches than buffers: every slot is reused, results must never mix
            buf = g.next_local()
            buf.copy_(torch.full((rows, 3), float(100 * k + rank)))  # "the matcher wrote batch k"
            slots.append((k, g.launch()))
            if k >= 1:  # consume the previous batch while this one is in flight
                kk, sl = slots[k - 1]
                out = g.result(sl)
                exp = torch.cat([torch.full((rows, 3), float(100 * kk + r)) for r in range(world)])
                ok = ok and bool(torch.equal(out, exp))
        g.wait_all()
        out = g.result(slots[-1][1])
        exp = torch.cat([torch.full((rows, 3), float(100 * 6 + r)) for r in range(world)])
        q.put((rank, ok and bool(torch.equal(out, exp))))
    finally:
        dist.destroy_process_group()


def test_two_rank_async_gather_double_buffered():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, 4096, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)


def _worker_bucketed(rank, world, port, rows, q):
    sys.path.insert(0, ROOT)
    from hector_slam_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.BucketedRowGather(rows, 3, "cpu", bucket=4, depth=2)
        ok = True
        seen = {}
        for k in range(23):  # 5 full buckets + a partial one: every block is reused, batches must never mix
            buf = g.next_local()
            buf.copy_(torch.full((rows, 3), float(100 * k + rank)))  # "the matcher wrote batch k"
            g.launch()
            if k % 5 == 2:  # look at the most recent batch now and then (flushes a partial bucket early)
                out = g.last_result()
                exp = torch.cat([torch.full((rows, 3), float(100 * k + r)) for r in range(world)])
                ok = ok and bool(torch.equal(out, exp))
                seen[k] = True
        g.flush()
        g.wait_all()
        out = g.last_result()
        exp = torch.cat([torch.full((rows, 3), float(100 * 22 + r)) for r in range(world)])
        ok = ok and bool(torch.equal(out, exp))
        q.put((rank, ok, g.collectives))
    finally:
        dist.destroy_process_group()


def test_two_rank_bucketed_gather():
    """one all-gather per bucket of batches (what bench.py's N > 1 loop uses): blocks rotate, partial buckets flush, the most
    recent batch is always retrievable, and fewer collectives run than batches"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bucketed, args=(r, 2, port, 257, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(5 <= n < 23 for _, _, n in res), res


def test_bucketed_gather_without_process_group_is_identity():
    from hector_slam_amd import sharding
    g = sharding.BucketedRowGather(5, 3, "cpu", bucket=3)
    for k in range(5):
        g.next_local().fill_(float(k))
        g.launch()
    assert torch.equal(g.last_result(), torch.full((5, 3), 4.0)) and g.collectives == 0


def test_async_gather_without_process_group_is_identity():
    from hector_slam_amd import sharding
    g = sharding.AsyncRowGather(5, 3, "cpu")
    b = g.next_local()
    b.fill_(2.0)
    s = g.launch()
    assert torch.equal(g.result(s), torch.full((5, 3), 2.0))


@pytest.mark.parametrize("total", [8, 9, 4097])
def test_two_rank_gather_reassembles_batch(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    spans = sorted((b, e) for _, _, b, e in res)
    assert spans[0][0] == 0 and spans[0][1] == spans[1][0] and spans[1][1] == total


def test_shard_bounds_partition():
    from hector_slam_amd import sharding
    for total in (0, 1, 7, 4096, 32768, 32771):
        for world in (1, 2, 4, 8):
            spans = [sharding.shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == sharding.max_shard(total, world)


def test_shard_bounds_is_the_native_rule():
    """one partitioning for both transports: sharding.shard_bounds IS hsm_shard_bounds (the rule hsm_group_match_batch splits
    by), equal to the closed form, and a bad rank / world is refused"""
    import pytest
    from hector_slam_amd import capi, sharding
    for total in (0, 1, 5, 10, 4095, 4096, 32771):
        for world in (1, 2, 3, 4, 7, 8):
            for r in range(world):
                base, rem = divmod(total, world)
                b = r * base + min(r, rem)
                assert sharding.shard_bounds(total, r, world) == capi.shard_bounds(total, r, world) == (b, b + base + (1 if r < rem else 0))
    for bad in ((10, 4, 4), (10, -1, 4), (10, 0, 0), (-1, 0, 2)):
        with pytest.raises(capi.HsmError):
            capi.shard_bounds(*bad)


def _worker_replica(rank, world, port, q):
    """configs[4] protocol on CPU: the CPU oracle stands in for the GPU replica (same deterministic updateByScan)"""
    sys.path.insert(0, ROOT)
    import numpy as np
    from hector_slam_amd import sharding, synth
    from oracle import pyoracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = synth.make_scene(n_beams=181, map_size=128, levels=2, resolution=0.1, n_build=10, n_query=1,
                              room=(10.0, 8.0), seed=5)
        rep = pyoracle.Oracle("ho", sc.resolution, sc.map_size, sc.map_size, sc.levels)
        sync = sharding.ReplicaSync(max_beams=256, device="cpu")
        pose = sc.build_poses[0].copy()
        for t in range(8):
            if rank == 0:  # the scan arrives at rank 0 only, which matches it (a no-op on the still empty map)
                scan = sc.build_scans[t]
                hint = pose + (sc.build_poses[t] - sc.build_poses[max(t - 1, 0)])
                pose = rep.match(hint, scan)[0]
                pose_b, scan_b = sync.broadcast(pose, scan)
            else:
                pose_b, scan_b = sync.broadcast(None, None)
                rep.match(pose_b, scan_b)  # retains the coarse-level containers like rank 0's matchData did
            rep.update_by_scan(pose_b, scan_b)
            rep.on_map_updated()
        dig = [sharding.map_digest(*rep.download_level(lvl)) for lvl in range(sc.levels)]
        same = sync.digests_equal(dig)
        bad = sync.digests_equal([d + rank for d in dig])  # negative control: rank-dependent digests must differ
        q.put((rank, same, bad, dig))
    finally:
        dist.destroy_process_group()


def test_two_rank_replica_replay_keeps_maps_identical():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_replica, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(same for _, same, _, _ in res) and not any(bad for _, _, bad, _ in res)
    assert res[0][3] == res[1][3] and all(d > 0 for d in res[0][3])


def _worker_oversize(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import numpy as np
    from hector_slam_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sync = sharding.ReplicaSync(max_beams=16, device="cpu")
        # a scan that fits, then one that does not: EVERY rank must raise (nobody may stay blocked in the collective), and the
        # protocol must still work afterwards
        ok1 = sync.broadcast(np.zeros(3, np.float32), np.ones((16, 2), np.float32))[1].shape == (16, 2) if rank == 0 else \
            sync.broadcast(None, None)[1].shape == (16, 2)
        raised = False
        try:
            sync.broadcast(np.zeros(3, np.float32), np.ones((17, 2), np.float32)) if rank == 0 else sync.broadcast(None, None)
        except ValueError:
            raised = True
        after = sync.broadcast(np.full(3, 2.0, np.float32), np.full((3, 2), 5.0, np.float32)) if rank == 0 else sync.broadcast(None, None)
        q.put((rank, bool(ok1), raised, after[0].tolist(), after[1].shape))
    finally:
        dist.destroy_process_group()


def test_two_rank_replica_broadcast_rejects_an_oversize_scan_on_every_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_oversize, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok1, raised, pose, shape in res:
        assert ok1 and raised and pose == [2.0, 2.0, 2.0] and shape == (3, 2), (rank, ok1, raised, pose, shape)


def _worker_group_child(rank, world, port, q):
    """bench.py's hand-off around the `--group N` child of rank 0: the other ranks wait on the rendezvous store (NOT on a
    device barrier) until rank 0's child has finished; nothing may hang, whatever the child does"""
    sys.path.insert(0, ROOT)
    import time
    import types
    from hsm_bench import group as bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def fake_child(extra_args, timeout_s=300, env=None):
            calls.append((list(extra_args), dict(env or {})))
            time.sleep(1.0)
            return {"value": 1.0, "args": list(extra_args)}

        bench.run_child = fake_child
        os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
        args = types.SimpleNamespace(steps=20, batch=4096)
        t0 = time.time()
        rec = bench.group_child_from_rank0(args, world, dist)
        dt = time.time() - t0
        ok = (rec is not None and rec["value"] == 1.0 and "--group" in rec["args"] and str(world) in rec["args"]) if rank == 0 else rec is None
        if rank == 0:  # the child must not inherit the torchrun identity of rank 0
            ok = ok and all(k not in calls[0][1] for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"))
        q.put((rank, bool(ok), dt))
    finally:
        dist.destroy_process_group()


def test_group_child_handoff_does_not_hang():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_group_child, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [True, True], res
    assert all(0.5 < r[2] < 30 for r in res), res  # rank 1 waited for the child, nobody waited for a timeout


def _worker_fallback(rank, world, port, q):
    """make_row_gather where the device-side exchange cannot be set up (no HIP device here): every rank must settle on the
    collective -- collectively -- say why, and still gather correctly"""
    sys.path.insert(0, ROOT)
    from hector_slam_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = 5
        g, kind, note = sharding.make_row_gather(world * rows, rows, 3, "cpu", lag=1, fallback_bucket=2)
        ok = kind == "rccl" and isinstance(g, sharding.BucketedRowGather) and "unavailable" in note and "rank 0" in note
        for k in range(3):
            g.next_local().copy_(torch.full((rows, 3), float(10 * k + rank)))
            g.launch()
        out = g.last_result()
        exp = torch.cat([torch.full((rows, 3), float(20 + r)) for r in range(world)])
        q.put((rank, bool(ok and torch.equal(out, exp)), note[:120]))
    finally:
        dist.destroy_process_group()


def _worker_fallback_one_rank(rank, world, port, q):
    """... and when only ONE rank cannot set its mailbox up (rank 0 gets a stand-in that 'works', rank 1 has no device): the set-up
    is collective, so rank 0 must not wait inside a handle exchange rank 1 never joins -- both settle on the collective"""
    sys.path.insert(0, ROOT)
    from hector_slam_amd import capi, sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if rank == 0:
            class Stub:
                def __init__(self, *a, **k):
                    self.closed = False

                def handle(self):
                    return b"x" * 64

                def connect(self, handles):
                    raise AssertionError("must not connect: a peer has no mailbox")

                def close(self):
                    self.closed = True
            capi.PoseExchange = Stub
        rows = 4
        g, kind, note = sharding.make_row_gather(world * rows, rows, 3, "cpu", lag=1, fallback_bucket=1)
        ok = kind == "rccl" and isinstance(g, sharding.BucketedRowGather) and "rank 1" in note
        g.next_local().copy_(torch.full((rows, 3), float(7 + rank)))
        g.launch()
        out = g.last_result()
        exp = torch.cat([torch.full((rows, 3), float(7 + r)) for r in range(world)])
        q.put((rank, bool(ok and torch.equal(out, exp)), note[:160]))
    finally:
        dist.destroy_process_group()


def test_gather_selection_falls_back_together_when_one_rank_has_no_mailbox():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: the exchange sets up (tests/test_gpu_exchange.py covers it)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fallback_one_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [True, True], res


def test_gather_selection_falls_back_to_the_collective_on_every_rank():
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present: the exchange sets up (tests/test_gpu_exchange.py covers it)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_fallback, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert [r[1] for r in res] == [True, True], res

"""The device-side gather of sharded results (hsm_exchange_*, csrc/pose_exchange.hip) on the GPU: the kernels, the peer-access
form inside one process, the IPC form between processes that share the box's device, lagged waits, ragged shards, the bounded
wait, the runtime's refusals -- and the same transport under sharding.DirectRowGather with the real matcher in front of it.
(The protocol's flow control is also run on CPU between processes: tests/test_exchange_protocol.py.)"""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rows(epoch, first, n, cols):
    """what a rank posts in `epoch` for its rows [first, first + n): distinct per epoch, row, column; a few special values"""
    r = np.arange(first, first + n, dtype=np.float64)[:, None]
    c = np.arange(cols, dtype=np.float64)[None, :]
    a = (np.sin(r * 0.37 + c * 1.3 + epoch * 0.11) * 1e3 + epoch).astype(np.float32)
    if n:
        a[0, 0] = np.float32(-0.0)
        a[-1, cols - 1] = np.float32(np.inf) if epoch % 3 == 0 else np.float32(1e-42)  # (a denormal travels as its bits)
    return a


@pytest.mark.parametrize("world,lag,total", [(2, 0, 4096), (2, 1, 8191), (3, 1, 100), (4, 2, 4099)])
def test_exchange_inside_one_process(world, lag, total):
    """`world` ranks on device 0, one stream each, peer access: every epoch's gathered array on every rank equals what the
    ranks posted, bit for bit -- lag 0 / 1 / 2, equal and ragged shards, 40 epochs (the mailbox buffers are reused many times)"""
    import torch
    from hector_slam_amd import capi, sharding
    cols, epochs = 3, 40
    xs = [capi.PoseExchange(r, world, total, cols, depth=2 + 2 * lag, device=0) for r in range(world)]
    assert xs[0].memory_kind() in ("uncached", "fine-grained")
    for x in xs:
        x.connect_local(xs)
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
    bounds = [sharding.shard_bounds(total, r, world) for r in range(world)]
    outs = [[torch.zeros((total, cols), dtype=torch.float32, device=dev) for _ in range(epochs)] for _ in range(world)]
    local = [torch.zeros((bounds[r][1] - bounds[r][0], cols), dtype=torch.float32, device=dev) for r in range(world)]
    want = []
    for e in range(1, epochs + 1):
        want.append(np.concatenate([_rows(e, b, en - b, cols) for b, en in bounds]))
        for r in range(world):
            b, en = bounds[r]
            with torch.cuda.stream(streams[r]):
                local[r].copy_(torch.from_numpy(want[-1][b:en]), non_blocking=False)  # "the matcher wrote batch e" (stream ordered)
                w = e - lag
                xs[r].post_wait(local[r].data_ptr(), b, en - b, lag, outs[r][w - 1].data_ptr() if w >= 1 else 0, streams[r].cuda_stream)
    for r in range(world):  # drain
        for w in range(epochs - lag + 1, epochs + 1):
            xs[r].wait(outs[r][w - 1].data_ptr(), streams[r].cuda_stream)
    torch.cuda.synchronize()
    for r in range(world):
        xs[r].check()
        assert xs[r].epochs() == (epochs, epochs)
        for e in range(epochs):
            assert np.array_equal(bits(outs[r][e].cpu().numpy()), bits(want[e])), (r, e)
    for x in xs:
        x.close()


def test_exchange_refuses_what_the_protocol_cannot_carry():
    import torch
    from hector_slam_amd import capi
    with pytest.raises(capi.HsmError):
        capi.PoseExchange(0, 17, 16, 3)  # more ranks than HSM_EXCHANGE_MAX_WORLD
    with pytest.raises(capi.HsmError):
        capi.PoseExchange(0, 1, 16, 3, depth=1)
    x = capi.PoseExchange(0, 2, 16, 3, depth=2, device=0)
    d = torch.zeros((8, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(capi.HsmError, match="not connected"):
        x.post(d.data_ptr(), 0, 8)
    y = capi.PoseExchange(1, 2, 16, 3, depth=2, device=0)
    x.connect_local([x, y])
    y.connect_local([x, y])
    with pytest.raises(capi.HsmError, match="depth"):
        x.post_wait(d.data_ptr(), 0, 8, 1, 0)  # lag 1 needs depth 4
    with pytest.raises(capi.HsmError, match="outside"):
        x.post(d.data_ptr(), 12, 8)
    with pytest.raises(capi.HsmError):
        x.wait(0)  # nothing posted
    x.post(d.data_ptr(), 0, 8)
    with pytest.raises(capi.HsmError, match="ahead"):
        x.post(d.data_ptr(), 0, 8)  # depth 2: the second post needs the first wait
    y.post(d.data_ptr(), 8, 8)
    x.wait(0)
    y.wait(0)
    torch.cuda.synchronize()
    x.check()
    x.close()
    y.close()


def test_a_rank_that_never_posts_costs_one_timeout(monkeypatch):
    """the wait is bounded: the rows of a silent peer read NaN, the status call fails and names the epoch, the device is fine"""
    import torch
    from hector_slam_amd import capi
    monkeypatch.setenv("HSM_EXCHANGE_TIMEOUT_MS", "30")
    x = capi.PoseExchange(0, 2, 64, 3, depth=2, device=0)
    y = capi.PoseExchange(1, 2, 64, 3, depth=2, device=0)
    x.connect_local([x, y])
    y.connect_local([x, y])
    rows = torch.full((32, 3), 7.0, dtype=torch.float32, device="cuda:0")
    out = torch.zeros((64, 3), dtype=torch.float32, device="cuda:0")
    x.post_wait(rows.data_ptr(), 0, 32, 0, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert (got[:32] == 7.0).all() and np.isnan(got[32:]).all()
    with pytest.raises(capi.HsmError, match="did not arrive"):
        x.check()
    assert float(torch.ones(4, device="cuda:0").sum().item()) == 4.0
    x.close()
    y.close()


def _ipc_rank(rank, world, port, total, cols, lag, epochs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from hector_slam_amd import sharding
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = sharding.DirectRowGather(total, cols, "cuda:0", lag=lag)
        ok, first_bad = True, None
        results = []
        for e in range(1, epochs + 1):
            g.next_local().copy_(torch.from_numpy(_rows(e, g.first_row, g.rows, cols)))
            g.launch()
            if g.landed >= 1:
                results.append((g.landed, g.out[g.landed % 2].clone()))  # stream ordered: a copy of what has just been unpacked
            if rank == 1 and e % 7 == 0:
                torch.cuda.synchronize()  # the ranks drift: one of them stalls now and then
        g.drain()
        torch.cuda.synchronize()
        results.append((g.landed, g.out[g.landed % 2].clone()))
        g.check()
        for e, t in results:
            bounds = [sharding.shard_bounds(total, r, world) for r in range(world)]
            want = np.concatenate([_rows(e, b, en - b, cols) for b, en in bounds])
            if not np.array_equal(t.cpu().numpy().view(np.uint32), want.view(np.uint32)):
                ok, first_bad = False, e
                break
        dist.barrier()
        q.put((rank, ok, first_bad, len(results), g.x.memory_kind(), g.collectives))
        g.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lag,total", [(1, 8192), (0, 4097)])
def test_exchange_between_processes_sharing_the_device(lag, total):
    """the IPC form: two PROCESSES (gloo carries the 64-byte handles once), both on the box's one device, 120 epochs through
    sharding.DirectRowGather; every epoch's gathered rows are bit-identical to what the two ranks posted, on both ranks; no
    torch.distributed collective on the data path"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_ipc_rank, args=(r, 2, port, total, 3, lag, 120, q)) for r in range(2)]
    [p.start() for p in ps]
    recs = sorted(q.get(timeout=300) for _ in range(2))
    [p.join(60) for p in ps]
    for rank, ok, first_bad, n, kind, collectives in recs:
        assert ok, (rank, first_bad)
        assert n >= 120 - lag and collectives == 0 and kind in ("uncached", "fine-grained")


def test_direct_gather_behind_the_matcher(pyramid_scene):
    """the shape bench.py --gpus N times, on one rank: match a batch, launch() the exchange on the same stream, lag 1 -- the
    gathered poses of every batch are the matcher's poses, bit for bit, and nothing but the exchange kernel ran in between"""
    import torch
    from hector_slam_amd import capi, sharding, synth
    sc = pyramid_scene
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, device=0)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    pts, offs = synth.pack_scans(sc.query_scans)
    B = len(sc.query_scans)
    want, _ = g.match_batch(sc.query_init, pts, offs)
    dev = torch.device("cuda", 0)
    d_pts, d_offs = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev)
    gat = sharding.DirectRowGather(B, 3, dev, lag=1)
    stream = torch.cuda.current_stream()
    seen = []
    for k in range(5):
        init = sc.query_init.copy()
        init[:, 0] += np.float32(0.002 * k)
        ref, _ = g.match_batch(init, pts, offs)
        d_init = torch.from_numpy(init).to(dev)
        g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), 1081, gat.next_local().data_ptr(), 0, stream.cuda_stream)
        gat.launch(stream)
        seen.append(ref)
        if gat.landed:
            assert np.array_equal(bits(gat.out[gat.landed % 2].cpu().numpy()), bits(seen[gat.landed - 1])), k
    assert np.array_equal(bits(gat.last_result().cpu().numpy()), bits(seen[-1]))
    torch.cuda.synchronize()
    gat.check()
    gat.close()
    g.close()
    assert want.shape == (B, 3)


@pytest.mark.parametrize("B", [16, 4096, 5000])
def test_matcher_launch_that_carries_the_exchange(B):
    """hsm_match_batch_device_gather: the exact-order batch forms post every scan's pose from the kernel's epilogue and unpack the batch
    before in extra workgroups at the end of the grid -- no launch of their own for the exchange.  Gathered poses == the matcher's
    poses, bit for bit, batch after batch, for the headline form (4096 scans: the launch carries the exchange), the chain-wavefront
    form (16 scans) and a launch that splits off its part-filled last generation (5000) -- those two leave the step to the
    stand-alone kernel behind them, through the same call; mixing both kinds of step shares the epochs"""
    import torch
    from hector_slam_amd import capi, sharding, synth
    sc = synth.make_scene(n_beams=1081, map_size=512, levels=3, resolution=0.05, n_build=40, n_query=16, room=(20.0, 15.0), seed=99)
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, device=0)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    if B == 4096:  # (... and through a permuted batch: the epilogue posts a scan's pose to the scan's own row)
        g.set_batch_order(capi.ORDER_MORTON)
    rng = np.random.default_rng(3)
    idx = rng.integers(0, len(sc.query_scans), B)
    pts, offs = synth.pack_scans([sc.query_scans[i] for i in idx])
    dev = torch.device("cuda", 0)
    d_pts, d_offs = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev)
    gat = sharding.DirectRowGather(B, 3, dev, lag=1)
    plain = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()
    seen = []
    for k in range(6):
        init = sc.query_init[idx].copy()
        init[:, :2] += rng.uniform(-0.03, 0.03, (B, 2)).astype(np.float32)
        d_init = torch.from_numpy(init).to(dev)
        g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), 1081, plain.data_ptr(), 0, stream.cuda_stream)
        seen.append(plain.cpu().numpy().copy())
        if k % 3 == 2:  # every third batch through the stand-alone kernel: the two forms share the epochs
            g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), 1081, gat.next_local().data_ptr(), 0, stream.cuda_stream)
            gat.launch(stream)
        else:
            gat.match_and_launch(g, B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), 1081, 0, stream)
            assert g.last_launch_config()["kernel"].startswith("gn_match_exact_cached_kernel"), g.last_launch_config()
        assert np.array_equal(bits(gat.next_local().cpu().numpy()), bits(seen[-1])), k  # the local rows are the matcher's
        if gat.landed:
            assert np.array_equal(bits(gat.out[gat.landed % 2].cpu().numpy()), bits(seen[gat.landed - 1])), (k, gat.landed)
    assert np.array_equal(bits(gat.last_result().cpu().numpy()), bits(seen[-1]))
    torch.cuda.synchronize()
    gat.check()
    assert gat.x.epochs() == (6, 6)
    gat.close()
    g.close()

"""GPU tests at BASELINE.json's full configuration sizes (configs[2..4], the single-GPU share of the
multi-GPU ones).

Parity bar, unconditional: in HSM_PARITY_EXACT the pose of EVERY scan of every batch is bit-identical to the
reference CPU matcher's ("hr": the unmodified reference headers, where oracle/_ref is present; else the restatement)
-- the whole batch is compared, settled scans and unsettled ones alike.  The default (fast) summation is then measured
against the exact mode on the same context (same map, same scans, same products, another summation order): its
deviation from the exact mode IS its deviation from the reference, with the bound stated per workload.  On top:
size-independent properties of the path -- determinism, permutation equivariance of the batch, batch == single-scan
calls, idempotence at convergence, and bit-identical maps given identical poses.
"""
import os

import numpy as np
import pytest

from conftest import ang_diff, bits, make_oracle, oracle_kinds

pytestmark = pytest.mark.gpu
TOL_M, TOL_RAD = 1e-4, 1e-4


def pose_err(p, q):
    p = np.asarray(p, np.float64).reshape(-1, 3)
    q = np.asarray(q, np.float64).reshape(-1, 3)
    return np.abs(p[:, :2] - q[:, :2]).max(), ang_diff(p[:, 2], q[:, 2]).max()


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available()
    from hector_slam_amd import capi as m
    m.load_library()
    return m


KIND = oracle_kinds()[-1]  # "hr" (reference-compiled) where available


def record(**kw):
    """parity statistics of the full-size runs, appended as JSON lines to $HSM_PARITY_STATS (tools/gpu_*.sh set it;
    the committed copies live under profiles/)"""
    import json
    import os
    print(json.dumps(kw))
    path = os.environ.get("HSM_PARITY_STATS")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(kw) + "\n")


def build_pair(capi, oracle_mod, sc, oracle_build=True):
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    o = make_oracle(oracle_mod, KIND, sc, build=oracle_build)
    return g, o


def oracle_match_all(oracle_mod, sc, init, pts, offs, threads=16):
    """matchData of the CPU reference for EVERY scan of the batch (threads over the host cores, one private oracle
    -- map + matcher + probability cache -- per thread)"""
    import threading
    B = init.shape[0]
    T = max(1, min(threads, B // 64))
    cpu = np.empty((B, 3), np.float32)

    def work(t):
        o = make_oracle(oracle_mod, KIND, sc)
        b, e = B * t // T, B * (t + 1) // T
        cpu[b:e] = o.match_many(init[b:e], pts[offs[b]:offs[e]], offs[b:e + 1] - offs[b])

    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    return cpu


def batch_properties(capi, oracle_mod, sc, g, init, scans, rng, fast_within_tol, fast_max_m):
    """shared checks for a big batched matchData; returns the exact-mode batch result"""
    from hector_slam_amd import synth
    B = len(scans)
    pts, offs = synth.pack_scans(scans)
    # ---- the default (HSM_PARITY_AUTO): the reference's summation order for EVERY batch (round 4; profiles/r04 scene sweep)
    assert g.parity() == capi.PARITY_AUTO
    pose_auto, _ = g.match_batch(init, pts, offs)
    assert g.last_launch_config()["parity_effective"] == "exact"
    # ---- exact mode: the WHOLE batch against the reference, bit for bit, no predicate
    g.set_parity(capi.PARITY_EXACT)
    pose_x, cov_x = g.match_batch(init, pts, offs)
    cpu = oracle_match_all(oracle_mod, sc, init, pts, offs)
    same = (bits(pose_x) == bits(cpu)).all(1)
    assert same.all(), f"exact mode: {(~same).sum()} of {B} poses differ from the reference ({KIND})"
    # ---- fast mode, measured against the exact mode
    g.set_parity(capi.PARITY_FAST)
    pose, cov = g.match_batch(init, pts, offs)
    assert g.last_launch_config()["parity_effective"] == "fast"
    assert np.array_equal(bits(pose_auto), bits(pose_x)), "HSM_PARITY_AUTO: the default batch result is the exact mode's"
    assert np.isfinite(pose).all() and np.isfinite(cov).all()
    d = np.abs(pose.astype(np.float64) - pose_x)
    dxy, dth = d[:, :2].max(1), ang_diff(pose[:, 2], pose_x[:, 2])
    ok = (dxy <= TOL_M) & (dth <= TOL_RAD)
    ident = (bits(pose) == bits(pose_x)).all(1)
    cfg = g.last_launch_config()
    print(f"B={B} kernel={cfg}: exact == {KIND} on {B}/{B}; fast vs exact: bit-identical {ident.mean():.4f}, within 1e-4 "
          f"{ok.mean():.5f}, median {np.median(dxy):.1e} m, p99.9 {np.percentile(dxy, 99.9):.1e} m, max {dxy.max():.1e} m")
    record(test="batch_properties", scene=f"{sc.map_size}^2 x{sc.levels} levels", batch=B, checker=KIND,
           exact_bit_identical_to_reference=int(same.sum()), fast_bit_identical_to_exact=float(ident.mean()),
           fast_within_1e4=float(ok.mean()), fast_median_dxy_m=float(np.median(dxy)),
           fast_p999_dxy_m=float(np.percentile(dxy, 99.9)), fast_max_dxy_m=float(dxy.max()), kernel=cfg)
    assert ok.mean() >= fast_within_tol, ok.mean()
    assert dxy.max() <= fast_max_m, dxy.max()  # the fast mode never leaves the reference's basin
    # determinism
    pose2, cov2 = g.match_batch(init, pts, offs)
    assert np.array_equal(bits(pose), bits(pose2)) and np.array_equal(bits(cov), bits(cov2))
    # permutation equivariance (scan order inside the batch must not matter, bit for bit)
    perm = rng.permutation(B)
    pts_p, offs_p = synth.pack_scans([scans[i] for i in perm])
    pose_p, _ = g.match_batch(init[perm], pts_p, offs_p)
    assert np.array_equal(bits(pose_p), bits(pose[perm]))
    # batch == single-scan calls: bit for bit in exact mode (any team width), tolerance in fast mode (other tree)
    sel = rng.choice(B, size=6, replace=False)
    for q in sel:
        if ok[q]:
            ps, _ = g.matchData(init[q], scans[q])
            e = pose_err(ps, pose[q])
            assert e[0] <= 2 * TOL_M and e[1] <= 2 * TOL_RAD
    g.set_parity(capi.PARITY_EXACT)
    for q in sel:
        ps, cs = g.matchData(init[q], scans[q])
        assert np.array_equal(bits(ps), bits(pose_x[q])) and np.array_equal(bits(cs), bits(cov_x[q]))
    g.set_parity(capi.PARITY_FAST)
    # idempotence at convergence: one more matchData from the converged pose stays put (median; the reference itself
    # keeps moving on the unsettled tail of the big-room scenes)
    pose3, _ = g.match_batch(pose, pts, offs)
    d3 = np.abs(pose3.astype(np.float64) - pose)
    assert np.median(d3[:, :2]) <= 2e-5
    return pose_x


def test_config3_batch4096_2048map(capi, oracle_mod):
    """configs[2]: batch=4096 concurrent 1081-beam scans, 2048^2 map (3-level pyramid 2048/1024/512)"""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=2048, levels=3, resolution=0.05, n_build=120, n_query=4096,
                          room=(40.0, 30.0), seed=1234)
    g, o = build_pair(capi, oracle_mod, sc)
    for lvl in range(sc.levels):  # the map the GPU built == the map the oracle built, bit for bit
        a, b = g.download_level(lvl), o.download_level(lvl)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1])
    pose = batch_properties(capi, oracle_mod, sc, g, sc.query_init, sc.query_scans, np.random.default_rng(5),
                            fast_within_tol=0.998, fast_max_m=5e-3)
    err = np.abs(pose.astype(np.float64) - sc.query_truth)
    assert np.median(err[:, :2]) < 0.02  # it converges to the ground truth, too


def test_batches_below_4096_scans_take_the_chain_wavefront_form(capi, oracle_mod):
    """round 5: a default-mode (reference-order) batch of up to 3072 scans runs one wavefront per scan plus a chain-only fifth
    wavefront per workgroup (gn_match_exact.h, CW) -- 1 / 5 / 1024 / 3072 scans: that kernel, EVERY pose and covariance equal
    to the reference's, bit for bit; 3073 scans: round 3's rotating-owner form, the same bits for the scans they share"""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=2048, levels=3, resolution=0.05, n_build=120, n_query=3073,
                          room=(40.0, 30.0), seed=4321)
    g, o = build_pair(capi, oracle_mod, sc)
    assert g.parity() == capi.PARITY_AUTO
    pts, offs = synth.pack_scans(sc.query_scans)
    cpu = oracle_match_all(oracle_mod, sc, sc.query_init, pts, offs)
    full = None
    for B in (3073, 3072, 1024, 5, 1):  # (batches beyond one generation of workgroups: test_batch_of_more_than_one_generation...)
        p_b, o_b = synth.pack_scans(sc.query_scans[:B])
        pose, cov = g.match_batch(sc.query_init[:B], p_b, o_b)
        cfg = g.last_launch_config()
        assert cfg["parity_effective"] == "exact" and cfg["texel_cache"], cfg
        if B > 3072:
            assert cfg["block"] == 256 and "chain wavefront" not in cfg["kernel"], cfg
            full = (pose, cov)
        else:
            assert cfg["block"] == 320 and "chain wavefront" in cfg["kernel"] and cfg["grid"] == (B + 3) // 4, cfg
            # (3072 scans: three workgroups per CU, the 80-VGPR instantiation with six cached rows; up to 2048: the full cache)
            assert np.array_equal(bits(pose), bits(full[0][:B])) and np.array_equal(bits(cov), bits(full[1][:B]))
        same = (bits(pose) == bits(cpu[:B])).all(1)
        assert same.all(), f"B={B}: {(~same).sum()} poses differ from the reference ({KIND})"
        record(test="chain_wavefront_form", batch=B, checker=KIND, bit_identical_to_reference=int(same.sum()), kernel=cfg)


def test_batch_of_more_than_one_generation_splits_off_its_tail(capi, oracle_mod):
    """5000 scans = one whole generation of the rotating-owner form (4096) + 904 in the chain-wavefront form, two launches behind
    one call: every pose and covariance equal to the reference's, and equal to the single-launch result (HSM_EXACT_SPLIT_TAIL=0)"""
    import os
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=2048, levels=3, resolution=0.05, n_build=120, n_query=5000,
                          room=(40.0, 30.0), seed=99)
    g, o = build_pair(capi, oracle_mod, sc)
    pts, offs = synth.pack_scans(sc.query_scans)
    pose, cov = g.match_batch(sc.query_init, pts, offs)
    cfg = g.last_launch_config()
    assert "part-filled generation" in cfg["kernel"] and cfg["grid"] == 1250, cfg
    cpu = oracle_match_all(oracle_mod, sc, sc.query_init, pts, offs)
    same = (bits(pose) == bits(cpu)).all(1)
    assert same.all(), f"{(~same).sum()} of 5000 poses differ from the reference ({KIND})"
    os.environ["HSM_EXACT_SPLIT_TAIL"] = "0"
    try:
        g1 = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    finally:
        del os.environ["HSM_EXACT_SPLIT_TAIL"]
    g1.setUpdateFactorFree(0.4)
    g1.setUpdateFactorOccupied(0.9)
    g1.build_map(sc.build_poses, sc.build_scans)
    pose1, cov1 = g1.match_batch(sc.query_init, pts, offs)
    assert "part-filled" not in g1.last_launch_config()["kernel"] and g1.last_launch_config()["grid"] == 1250
    assert np.array_equal(bits(pose), bits(pose1)) and np.array_equal(bits(cov), bits(cov1))
    # the shared-scan form (no offsets: every hypothesis matches the same scan) through the same split
    init = np.repeat(sc.query_init[:1], 4100, axis=0) + np.random.default_rng(3).uniform(-0.03, 0.03, (4100, 3)).astype(np.float32)
    pa, ca = g.match_batch(init, sc.query_scans[0])
    pb, cb = g1.match_batch(init, sc.query_scans[0])
    assert "part-filled" in g.last_launch_config()["kernel"]
    assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(ca), bits(cb))
    record(test="split_tail", batch=5000, checker=KIND, bit_identical_to_reference=int(same.sum()))


def test_batch_of_scans_longer_than_the_cached_rows(capi, oracle_mod):
    """4096 scans of up to 1400 beams: seventeen rows per lane are cached / staged, the beams beyond them stream from memory in
    extra rounds behind the balanced schedule's last barrier (gn_match_exact.h) -- every pose and covariance equal to the reference's"""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1400, map_size=2048, levels=3, resolution=0.05, n_build=60, n_query=4096, room=(40.0, 30.0), seed=77)
    g, o = build_pair(capi, oracle_mod, sc)
    pts, offs = synth.pack_scans(sc.query_scans)
    assert int(np.diff(offs).max()) > 17 * 64
    pose, cov = g.match_batch(sc.query_init, pts, offs)
    cfg = g.last_launch_config()
    assert cfg["kernel"].startswith("gn_match_exact_cached_kernel") and cfg["block"] == 256 and cfg["grid"] == 1024, cfg
    cpu = oracle_match_all(oracle_mod, sc, sc.query_init, pts, offs)
    same = (bits(pose) == bits(cpu)).all(1)
    assert same.all(), f"{(~same).sum()} of 4096 poses differ from the reference ({KIND})"
    record(test="long_scans_batch", batch=4096, checker=KIND, bit_identical_to_reference=int(same.sum()), kernel=cfg)


@pytest.mark.parametrize("beams,rows", [(300, 5), (560, 9), (720, 13)])
def test_full_batch_of_short_scans_takes_the_short_row_forms(capi, oracle_mod, beams, rows):
    """4096 ragged scans of at most 300 / 560 / 720 beams, four workgroups per CU: the 5- / 9- / 13-row instantiations of the headline
    kernel on the balanced run-ahead schedule (every row cached) -- every pose and covariance equal to the reference's"""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=beams, map_size=1024, levels=3, resolution=0.05, n_build=60, n_query=4096, room=(40.0, 30.0), seed=beams)
    g, o = build_pair(capi, oracle_mod, sc)
    rng = np.random.default_rng(beams)
    scans = [sq[: max(0, sq.shape[0] - int(rng.integers(0, 64)))] if q % 5 else sq for q, sq in enumerate(sc.query_scans)]
    scans[17] = scans[17][:0]  # an empty scan passes its start pose through
    pts, offs = synth.pack_scans(scans)
    pose, cov = g.match_batch(sc.query_init, pts, offs)
    cfg = g.last_launch_config()
    assert cfg["kernel"] == "gn_match_exact_cached_kernel" and cfg["block"] == 256 and cfg["grid"] == 1024 and cfg["beams_per_lane"] == rows, cfg
    cpu = oracle_match_all(oracle_mod, sc, sc.query_init, pts, offs)
    same = (bits(pose) == bits(cpu)).all(1)
    assert same.all(), f"{(~same).sum()} of 4096 poses differ from the reference ({KIND})"
    assert np.array_equal(bits(pose[17]), bits(sc.query_init[17]))
    record(test="short_scans_batch", beams=beams, rows=rows, checker=KIND, bit_identical_to_reference=int(same.sum()))


@pytest.mark.parametrize("B", [1500, 4096, 5000])
def test_batch_in_morton_order_gives_the_same_bits(capi, oracle_mod, B):
    """hsm_set_batch_order(HSM_ORDER_MORTON): a one-workgroup counting sort in front of the matcher lays a randomly ordered batch out
    by the map tile of its start poses and the matcher takes its scans through that permutation -- every pose and covariance lands
    at the scan's own index with the bits of the caller's order: the chain-wavefront form (1500 scans), the headline form (4096),
    a launch that splits off its part-filled last generation (5000), the fast tree form; ragged scans, an empty one; the default
    is the caller's order"""
    import torch
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=1024, levels=3, resolution=0.05, n_build=60, n_query=512, room=(40.0, 30.0), seed=9)
    g, o = build_pair(capi, oracle_mod, sc, oracle_build=False)
    rng = np.random.default_rng(B)
    idx = rng.integers(0, 512, B)
    scans = [sc.query_scans[i][: 1081 - int(rng.integers(0, 300))] for i in idx]
    scans[5] = scans[5][:0]
    init = sc.query_init[idx] + rng.uniform(-0.05, 0.05, (B, 3)).astype(np.float32) * np.float32([1, 1, 0.2])
    pts, offs = synth.pack_scans(scans)
    assert g.batch_order() == capi.ORDER_AUTO  # (on a map of 2^20 cells: the caller's order)
    for mode in (capi.PARITY_AUTO, capi.PARITY_FAST):
        g.set_parity(mode)
        g.set_batch_order(capi.ORDER_GIVEN)
        p0, c0 = g.match_batch(init, pts, offs)
        assert not g.last_launch_sorted()
        k0 = g.last_launch_config()["kernel"]
        g.set_batch_order(capi.ORDER_MORTON)
        far = init.copy()
        far[:, :2] += np.float32(3.0)  # (results of another basin into the result buffers: a slot the permutation missed would show)
        g.match_batch(far, pts, offs)
        p1, c1 = g.match_batch(init, pts, offs)
        cached = g.last_launch_config()["texel_cache"]  # (the tree mode runs 1500 scans on teams of wavefronts: the caller's order)
        assert g.last_launch_sorted() == bool(cached) and g.last_launch_config()["kernel"] == k0, g.last_launch_config()
        assert cached or (mode == capi.PARITY_FAST and B == 1500)
        live = np.arange(B) != 5  # (an empty scan leaves its covariance untouched: whatever the output buffer held)
        assert np.array_equal(bits(p0), bits(p1)) and np.array_equal(bits(c0[live]), bits(c1[live])), mode
    assert np.array_equal(bits(p1[5]), bits(init[5]))  # the empty scan's start pose passes through, at its own index
    # shared scan (pose hypotheses of ONE scan), sorted
    g.set_parity(capi.PARITY_AUTO)
    g.set_batch_order(capi.ORDER_GIVEN)
    h0, _ = g.match_batch(init, sc.query_scans[3], None)
    g.set_batch_order(capi.ORDER_MORTON)
    h1, _ = g.match_batch(init, sc.query_scans[3], None)
    assert g.last_launch_sorted() and np.array_equal(bits(h0), bits(h1))
    # the permutation serves eight launches by default; computed for every launch it gives the same bits again
    g.set_batch_order_refresh(1)
    h2, _ = g.match_batch(init, sc.query_scans[3], None)
    h3, _ = g.match_batch(init, sc.query_scans[3], None)
    assert g.last_launch_sorted() and np.array_equal(bits(h0), bits(h2)) and np.array_equal(bits(h0), bits(h3))
    with pytest.raises(capi.HsmError):
        g.set_batch_order_refresh(0)
    if B == 4096:  # more scans than one pass of the sort kernel holds (8192): 20 000 hypotheses of one scan
        hyp = np.tile(init, (5, 1))[:20000] + rng.uniform(-0.2, 0.2, (20000, 3)).astype(np.float32) * np.float32([1, 1, 0.1])
        g.set_batch_order(capi.ORDER_GIVEN)
        a0, _ = g.match_batch(hyp, sc.query_scans[7], None)
        g.set_batch_order(capi.ORDER_MORTON)
        g.match_batch(hyp + np.float32(1.0), sc.query_scans[7], None)
        g.set_batch_order_refresh(1)
        a1, _ = g.match_batch(hyp, sc.query_scans[7], None)
        assert g.last_launch_sorted() and np.array_equal(bits(a0), bits(a1))
    # a small batch keeps the caller's order; a single scan is not a batch
    g.match_batch(init[:64], *synth.pack_scans(scans[:64]))
    assert not g.last_launch_sorted()
    with pytest.raises(capi.HsmError):
        g.set_batch_order(7)
    g.close()


def test_batch_order_environment_word(capi, monkeypatch):
    for word, order in (("morton", capi.ORDER_MORTON), ("given", capi.ORDER_GIVEN), ("auto", capi.ORDER_AUTO)):
        monkeypatch.setenv("HSM_BATCH_ORDER", word)
        g = capi.MapRepMultiMap(0.05, 256, 256, 1)
        assert g.batch_order() == order
        g.close()
    monkeypatch.setenv("HSM_BATCH_ORDER", "sorted")
    with pytest.raises(capi.HsmError):
        capi.MapRepMultiMap(0.05, 256, 256, 1)


def test_batched_matches_can_be_captured_in_a_hip_graph(capi, oracle_mod):
    """hsm_match_batch_device on a caller's stream is kernel launches only: a loop of batched matches can be captured into a hipGraph
    (torch.cuda.CUDAGraph on ROCm) and replayed -- same poses bit for bit, in the default mode and in Morton order (whose sort kernel
    and permutation buffer belong to the stream: allocated by the warm-up launch, not during the capture)"""
    import torch
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=1024, levels=3, resolution=0.05, n_build=60, n_query=4096, room=(40.0, 30.0), seed=21)
    g, o = build_pair(capi, oracle_mod, sc, oracle_build=False)
    g.synchronize()
    dev = torch.device("cuda", 0)
    pts, offs = synth.pack_scans(sc.query_scans)
    d_pts, d_offs, d_init = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(sc.query_init).to(dev)
    d_pose = torch.zeros((4096, 3), dtype=torch.float32, device=dev)
    s = torch.cuda.Stream()

    def launch():
        g.match_batch_device(4096, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), 1081, d_pose.data_ptr(), 0, s.cuda_stream)
    for order in (capi.ORDER_GIVEN, capi.ORDER_MORTON):
        g.set_batch_order(order)
        g.set_batch_order_refresh(2)
        launch()
        torch.cuda.synchronize()
        ref = d_pose.cpu().numpy().copy()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(5):
                launch()
        for _ in range(3):
            d_pose.zero_()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(bits(d_pose.cpu().numpy()), bits(ref)), order
    # a stream whose FIRST sorted launch happens inside a capture: no permutation buffer exists yet and none is allocated while
    # capturing -- those launches keep the caller's order; same bits
    s2 = torch.cuda.Stream()
    g.set_batch_order(capi.ORDER_MORTON)
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2, stream=s2):
        for _ in range(2):
            g.match_batch_device(4096, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), 1081, d_pose.data_ptr(), 0, s2.cuda_stream)
    assert not g.last_launch_sorted()
    d_pose.zero_()
    torch.cuda.synchronize()
    graph2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(bits(d_pose.cpu().numpy()), bits(ref))
    g.close()


def test_clock_probe_in_the_reference_order_batch_kernel(capi, oracle_mod):
    """hsm_set_clock_probe on a batch of 4096 full-length scans in the default mode: the launch takes the instantiation of the
    headline kernel that carries the stamps (PROBE, gn_match_exact.h) -- same poses and covariances bit for bit, and the ratio
    of the two counters is a clock an MI355X can run at; without a probe the launch carries no stamp code"""
    import torch
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=1024, levels=1, resolution=0.05, n_build=60, n_query=4096, room=(40.0, 30.0), seed=5)
    g, o = build_pair(capi, oracle_mod, sc, oracle_build=False)
    pts, offs = synth.pack_scans(sc.query_scans)
    p0, c0 = g.match_batch(sc.query_init, pts, offs)
    cfg = g.last_launch_config()
    assert cfg["kernel"].startswith("gn_match_exact_cached_kernel") and cfg["grid"] == 1024, cfg
    stamps = torch.zeros(4, dtype=torch.int64, device="cuda")
    g.set_clock_probe(stamps.data_ptr())
    p1, c1 = g.match_batch(sc.query_init, pts, offs)
    torch.cuda.synchronize()
    g.set_clock_probe(0)
    st = stamps.cpu().numpy().astype(np.uint64)
    assert np.array_equal(bits(p0), bits(p1)) and np.array_equal(bits(c0), bits(c1))
    assert st[3] > st[1] and st[2] > st[0], st
    ghz = float(st[2] - st[0]) / float(st[3] - st[1]) * 0.1
    assert 0.8 < ghz < 3.0, ghz
    stamps.zero_()
    g.match_batch(sc.query_init, pts, offs)
    torch.cuda.synchronize()
    assert not stamps.cpu().numpy().any()  # switched off: nothing is written
    record(test="clock_probe_exact", ghz=ghz)


def test_config4_share_4096map_pyramid(capi, oracle_mod):
    """configs[3], one GPU's share: 4096 of the 32768 scans, 3-level 4096/2048/1024 pyramid.  0.05 m cells, the
    room scaled to 160 m x 120 m and a 120 m sensor so that the 204.8 m map is actually used.
    (With 0.0125 m cells instead, the 1 cm range noise spans a cell and the REFERENCE's own Gauss-Newton no
    longer settles -- it still moves 1.5 cm when restarted from its own result -- so 2 % of the scans amplify
    last-bit differences beyond 1e-4 m; measured with tests/tools/dev_stats.py)"""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=1081, map_size=4096, levels=3, resolution=0.05, n_build=100, n_query=4096,
                          room=(160.0, 120.0), seed=77, range_max=120.0)
    g, o = build_pair(capi, oracle_mod, sc)
    assert g.level_info(0)[:2] == (4096, 4096) and g.level_info(2)[:2] == (1024, 1024)
    a, b = g.download_level(0), o.download_level(0)
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1])
    # 30 % of these scans have not settled in the REFERENCE (restarted from its own result it still moves > 1 mm: the far
    # walls of the 160 m room are mapped as dotted lines); the exact mode -- the default for batches on a map of this size
    # -- reproduces them all the same; the fast mode agrees to 1e-4 m on 99.2 % (measured) and stays in the same basin on
    # the rest (worst measured 0.07 m)
    batch_properties(capi, oracle_mod, sc, g, sc.query_init, sc.query_scans, np.random.default_rng(6),
                     fast_within_tol=0.99, fast_max_m=0.1)
    # HSM_ORDER_AUTO (the default) on a map of this size: the batch goes through the sort kernel's permutation -- the identity for
    # this batch, which follows the trajectory; a Morton order for the same batch shuffled -- and every pose keeps its bits
    pts, offs = synth.pack_scans(sc.query_scans)
    assert g.batch_order() == capi.ORDER_AUTO
    p_auto, _ = g.match_batch(sc.query_init, pts, offs)
    assert g.last_launch_sorted()
    g.set_batch_order(capi.ORDER_GIVEN)
    p_given, _ = g.match_batch(sc.query_init, pts, offs)
    assert not g.last_launch_sorted() and np.array_equal(bits(p_auto), bits(p_given))
    perm = np.random.default_rng(8).permutation(len(sc.query_scans))
    pts_r, offs_r = synth.pack_scans([sc.query_scans[i] for i in perm])
    g.set_batch_order(capi.ORDER_AUTO)
    p_r, _ = g.match_batch(sc.query_init[perm], pts_r, offs_r)
    assert g.last_launch_sorted() and np.array_equal(bits(p_r), bits(p_given[perm]))


def config5_loop(capi, oracle_mod, steps, seed=31, **ctx_kw):
    """configs[4], one replica: dense 16384-beam scans, 8192^2 map (3 levels), matchData + updateByScan interleaved through
    the HectorSlamProcessor loop with zero update thresholds (every step updates); GPU and reference each run their OWN loop
    (own matched pose into the update and into the next hint).  Yields (t, gpu pose, reference pose) and returns both maps."""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=16384, map_size=8192, levels=3, resolution=0.05, n_build=steps + 1, n_query=2,
                          room=(320.0, 240.0), seed=seed, range_max=240.0)
    assert min(s.shape[0] for s in sc.build_scans) > 14000
    o = make_oracle(oracle_mod, KIND, sc, build=False)
    o.proc_set_thresholds(0.0, 0.0)
    p = capi.HectorSlamProcessor(sc.resolution, sc.map_size, sc.map_size, (0.5, 0.5), sc.levels, **ctx_kw)
    p.setUpdateFactorFree(0.4)
    p.setUpdateFactorOccupied(0.9)
    p.setMapUpdateMinDistDiff(0.0)
    p.setMapUpdateMinAngleDiff(0.0)
    hint_o = hint_g = sc.build_poses[0].copy()
    poses = []
    for t in range(steps):
        o.proc_update(sc.build_scans[t], hint_o)
        p.update(sc.build_scans[t], hint_g)
        po, _ = o.proc_last_pose()
        pg = p.getLastScanMatchPose()
        poses.append((pg.copy(), po.copy()))
        step = sc.build_poses[t + 1] - sc.build_poses[t]
        hint_o, hint_g = po + step, pg + step
    return sc, p, o, poses


def test_config5_dense_scan_8192map_interleaved(capi, oracle_mod):
    """configs[4] in the library DEFAULT (HSM_PARITY_AUTO), SURVEY 8(d)'s T = 64 steps from an empty map: every matched pose,
    and after the last step every cell of every level (log-odds and stamps), bit-identical to the reference's own loop --
    since round 5 the default takes the reference's summation order on the single-scan entry point too (round-4 verdict:
    this test ran the fast dense matcher and tolerated 0.2 % differing cells)."""
    steps = int(os.environ.get("HSM_CONFIG5_STEPS", "64"))
    sc, p, o, poses = config5_loop(capi, oracle_mod, steps)
    assert p.mapRep.parity() == capi.PARITY_AUTO
    cfg = p.mapRep.last_launch_config()
    assert cfg["parity_effective"] == "exact", cfg
    dev = [pose_err(pg, po) for pg, po in poses]
    first_diff = next((t for t, (pg, po) in enumerate(poses) if not np.array_equal(bits(pg), bits(po))), None)
    record(test="config5_default_mode_free_running", steps=steps, checker=KIND, kernel=cfg,
           poses_bit_identical=sum(int(np.array_equal(bits(pg), bits(po))) for pg, po in poses),
           first_differing_step=first_diff, max_pose_dev_m=float(max(e[0] for e in dev)), max_pose_dev_rad=float(max(e[1] for e in dev)),
           per_step_dev_m=[float(e[0]) for e in dev])
    assert first_diff is None, (first_diff, poses[first_diff])
    for lvl in range(sc.levels):
        a, b = p.mapRep.download_level(lvl), o.download_level(lvl)
        assert (b[0] != 0).sum() > 100000
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), lvl
        del a, b


def test_config5_dense_scan_fast_mode_cooperative_matcher(capi, oracle_mod):
    """the same loop with HSM_PARITY_FAST (opt-in): the multi-workgroup dense matcher (tree summation) -- every pose within
    1e-4 m / 1e-4 rad of the reference's own loop; last-bit pose differences may flip a handful of Bresenham end cells"""
    steps = 10
    sc, p, o, poses = config5_loop(capi, oracle_mod, steps, parity=capi.PARITY_FAST)
    for t, (pg, po) in enumerate(poses):
        e = pose_err(pg, po)
        assert e[0] <= TOL_M and e[1] <= TOL_RAD, (t, e)
    cfg = p.mapRep.last_launch_config()
    assert cfg["waves_per_scan"] == -cfg["grid"] and 56 <= cfg["grid"] <= 64  # ~n / 256 cooperating workgroups (multi-CU dense matcher)
    differ = []
    for lvl in range(sc.levels):
        lo_g, _ = p.mapRep.download_level(lvl)
        lo_o, _ = o.download_level(lvl)
        touched = (lo_o != 0).sum()
        assert touched > 100000
        nd = int((bits(lo_g) != bits(lo_o)).sum())
        differ.append({"level": lvl, "touched": int(touched), "cells_differ": nd, "frac": nd / float(touched)})
        # the bound is 4x what MI355X measures (recorded below; default / exact mode: 0, previous and next test)
        assert nd <= 0.002 * touched, differ
    record(test="config5_fast_mode_free_running_maps", steps=steps, checker=KIND, cells_differing_from_reference=differ,
           bound_frac=0.002)
    # identical poses in -> bit-identical maps out (pure index work), at full size
    g2 = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, parity=capi.PARITY_FAST)
    g2.setUpdateFactorFree(0.4)
    g2.setUpdateFactorOccupied(0.9)
    o2 = make_oracle(oracle_mod, KIND, sc, build=False)
    for t in range(4):
        o2.match(sc.build_poses[t], sc.build_scans[t])
        g2.matchData(sc.build_poses[t], sc.build_scans[t])
        o2.update_by_scan(sc.build_poses[t], sc.build_scans[t])
        g2.updateByScan(sc.build_scans[t], sc.build_poses[t])
    for lvl in range(sc.levels):
        a, b = g2.download_level(lvl), o2.download_level(lvl)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), lvl


def test_config5_dense_scan_exact_mode_whole_slam_state(capi, oracle_mod):
    """configs[4] in HSM_PARITY_EXACT: 16384-beam scans, 8192^2 pyramid, match + update interleaved from an empty map --
    every pose bit-identical to the reference, so every update decision and all three maps are, too"""
    from hector_slam_amd import synth
    steps = 6
    sc = synth.make_scene(n_beams=16384, map_size=8192, levels=3, resolution=0.05, n_build=steps + 1, n_query=2,
                          room=(320.0, 240.0), seed=31, range_max=240.0)
    o = make_oracle(oracle_mod, KIND, sc, build=False)
    o.proc_set_thresholds(0.0, 0.0)
    p = capi.HectorSlamProcessor(sc.resolution, sc.map_size, sc.map_size, (0.5, 0.5), sc.levels, parity=capi.PARITY_EXACT)
    p.setUpdateFactorFree(0.4)
    p.setUpdateFactorOccupied(0.9)
    p.setMapUpdateMinDistDiff(0.0)
    p.setMapUpdateMinAngleDiff(0.0)
    hint = sc.build_poses[0].copy()
    for t in range(steps):
        o.proc_update(sc.build_scans[t], hint)
        p.update(sc.build_scans[t], hint)
        po, co = o.proc_last_pose()
        assert np.array_equal(bits(p.getLastScanMatchPose()), bits(po)), t
        assert np.array_equal(bits(p.getLastScanMatchCovariance()), bits(co)), t
        hint = po + (sc.build_poses[t + 1] - sc.build_poses[t])
    assert p.mapRep.last_launch_config()["waves_per_scan"] == 16
    for lvl in range(sc.levels):
        a, b = p.mapRep.download_level(lvl), o.download_level(lvl)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), lvl


def test_parity_sweep_32768_scans(capi, oracle_mod):
    """configs[3]'s full batch -- 32768 scans -- through the 3-level 2048/1024/512 matchData in 8 launches of 4096
    (what the 8 GPUs do in parallel), EVERY pose compared with the reference (threads over the host cores).
    Exact mode: 32768 of 32768 bit-identical.  Fast mode: fraction reported, >= 99.8 % within 1e-4 m / 1e-4 rad."""
    from hector_slam_amd import synth
    B, G = 4096, 8
    sc = synth.make_scene(n_beams=1081, map_size=2048, levels=3, resolution=0.05, n_build=120, n_query=B * G,
                          room=(40.0, 30.0), seed=4242)
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    pts, offs = synth.pack_scans(sc.query_scans)

    def gpu_all():
        return np.concatenate([g.match_batch(sc.query_init[k * B:(k + 1) * B], pts[offs[k * B]:offs[(k + 1) * B]],
                                             offs[k * B:(k + 1) * B + 1] - offs[k * B], want_cov=False)[0]
                               for k in range(G)])

    g.set_parity(capi.PARITY_EXACT)
    exact = gpu_all()
    g.set_parity(capi.PARITY_FAST)
    fast = gpu_all()
    g.set_parity(capi.PARITY_RELAXED)
    relaxed = gpu_all()
    cpu = oracle_match_all(oracle_mod, sc, sc.query_init, pts, offs)
    same = (bits(exact) == bits(cpu)).all(1)
    d = np.abs(fast.astype(np.float64) - cpu)
    ok = (d[:, 0] <= TOL_M) & (d[:, 1] <= TOL_M) & (ang_diff(fast[:, 2], cpu[:, 2]) <= TOL_RAD)
    ident = (bits(fast) == bits(cpu)).all(1)
    print(f"32768 scans vs {KIND}: exact mode bit-identical {same.sum()}/{same.size}; fast mode bit-identical "
          f"{ident.mean():.4f}, within tolerance {ok.mean():.5f}, worst {d[:, :2].max():.2e} m")
    record(test="parity_sweep_32768", checker=KIND, exact_bit_identical_to_reference=int(same.sum()), scans=int(same.size),
           fast_bit_identical_to_reference=float(ident.mean()), fast_within_1e4=float(ok.mean()),
           fast_worst_dxy_m=float(d[:, :2].max()))
    dr = np.abs(relaxed.astype(np.float64) - cpu)
    okr = (dr[:, 0] <= TOL_M) & (dr[:, 1] <= TOL_M) & (ang_diff(relaxed[:, 2], cpu[:, 2]) <= TOL_RAD)
    print(f"relaxed mode: within tolerance {okr.mean():.5f} ({int(okr.sum())}/{okr.size}), bit-identical "
          f"{(bits(relaxed) == bits(cpu)).all(1).mean():.4f}, worst {dr[:, :2].max():.2e} m")
    record(test="parity_sweep_32768_relaxed", checker=KIND, scans=int(okr.size), relaxed_within_1e4=float(okr.mean()),
           relaxed_bit_identical_to_reference=float((bits(relaxed) == bits(cpu)).all(1).mean()), relaxed_worst_dxy_m=float(dr[:, :2].max()))
    assert same.all(), (~same).sum()
    assert ok.mean() >= 0.998 and ident.mean() >= 0.95
    assert d[:, :2].max() <= 1e-3
    assert okr.mean() >= 0.998 and dr[:, :2].max() <= 1e-3  # the tolerance mode: the fast mode's bar

"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs and against the committed golden vectors.

Bars (BASELINE.json north_star / SURVEY.md 8(d)):
  * integer / index work (map update: log-odds + stamps): BIT-EXACT
  * world<->map transforms (host fp32): BIT-EXACT
  * sinf / cosf / expf on the device: BIT-EXACT against the host libm the reference links (glibc's
    algorithms restated in csrc/libm_exact.h)
  * per-beam terms M, dM/dx, dM/dy, rotDeriv and the probability texels: BIT-EXACT
  * pose estimate, fast (tree) summation -- what single scans run by default and batches on request: |dx|, |dy| <= 1e-4 m,
    |dtheta| <= 1e-4 rad  (POSE_TOL below);
    HSM_PARITY_EXACT: bit-identical (tests/test_gpu_exact_parity.py)

Every test that takes an oracle runs twice where oracle/_ref/libhector_ref.so is present: against the plain-C++
restatement ("ho") and against the UNMODIFIED reference headers compiled through the Eigen stand-in ("hr").
"""
import os

import numpy as np
import pytest

from conftest import ang_diff, bits, make_oracle, oracle_kinds, ulp_diff

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
# fast (tree) summation measured against the exact mode on maps made of a handful of scans, where Gauss-Newton has often not
# settled (test_randomised_geometries, test_processor_lifecycle_*): bounds = 2x what MI355X measures (round-3 advisor: a few
# millimetres of drift must not pass; the fast tree is deterministic -- no atomics -- so the measured values repeat)
# (measured: 83 / 84 within tolerance, worst 2.1e-3 m; lifecycle 34/34, 34/34, 33/34 within 1e-4 m, worst 4.1e-4 m)
FAST_RANDOM_WITHIN, FAST_RANDOM_WORST_M = 0.97, 4.2e-3
FAST_LIFECYCLE_WITHIN, FAST_LIFECYCLE_WORST_M = 0.96, 8.2e-4
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def assert_pose_close(p, q, what=""):
    p = np.asarray(p, np.float64).reshape(-1, 3)
    q = np.asarray(q, np.float64).reshape(-1, 3)
    dxy = np.abs(p[:, :2] - q[:, :2]).max()
    dth = ang_diff(p[:, 2], q[:, 2]).max()
    assert dxy <= POSE_TOL_M and dth <= POSE_TOL_RAD, f"{what}: dxy={dxy:.3e} m dth={dth:.3e} rad"
    return dxy, dth


def assert_dtr_close(d, dr, H, n):
    """dTr_k = sum_i g_ik * f_i is a sum of CANCELLING terms near convergence, so its error scales with
    the sum of |terms| <= sqrt(H_kk * n) (Cauchy-Schwarz, f_i <= 1), not with |dTr_k|: both the
    reference's sequential fp32 chain and the device's tree carry ~eps * sqrt(n) of that."""
    tol = 2e-5 * np.abs(dr) + 1e-7 * np.sqrt(np.abs(np.diag(H)) * n)
    assert (np.abs(d - dr) <= tol).all(), (d, dr, tol)


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a HIP device"
    from hector_slam_amd import capi as m
    m.load_library()  # raises if the native library is missing: no silent fallback
    return m


@pytest.fixture(autouse=True)
def fast_tree_by_default(monkeypatch):
    """contexts created in this file start in HSM_PARITY_FAST (see make_gpu), also those constructed directly"""
    monkeypatch.setenv("HSM_PARITY", "fast")


def make_gpu(capi, scene, free=0.4, occ=0.9, build=True, **kw):
    # this file exercises the FAST forms (tree summation: the texel-cache batch kernel, the team kernels, the dense matcher)
    # against their bars; the library default -- HSM_PARITY_AUTO: the reference's order for every batch -- has its own test
    # below, and the exact forms theirs in tests/test_gpu_exact_parity.py
    kw.setdefault("parity", capi.PARITY_FAST)
    g = capi.MapRepMultiMap(scene.resolution, scene.map_size, scene.map_size, scene.levels, **kw)
    g.setUpdateFactorFree(free)
    g.setUpdateFactorOccupied(occ)
    if build:
        g.build_map(scene.build_poses, scene.build_scans)
    return g


@pytest.fixture(scope="module", params=oracle_kinds())
def kind(request):
    """which CPU checker: "ho" = restatement, "hr" = the reference's own headers"""
    return request.param


@pytest.fixture(scope="module")
def pyr(capi, oracle_mod, pyramid_scene, kind):
    return make_gpu(capi, pyramid_scene), make_oracle(oracle_mod, kind, pyramid_scene)


@pytest.fixture(scope="module")
def sml(capi, oracle_mod, small_scene, kind):
    return make_gpu(capi, small_scene), make_oracle(oracle_mod, kind, small_scene)


# ---------------------------------------------------------------- geometry / storage
def test_level_geometry_and_transforms_bit_exact(pyr, pyramid_scene):
    g, o = pyr
    assert g.getMapLevels() == o.levels() == pyramid_scene.levels
    assert np.float32(g.getScaleToMap()) == np.float32(o.scale_to_map())
    rng = np.random.default_rng(11)
    for lvl in range(pyramid_scene.levels):
        assert g.level_info(lvl) == o.level_info(lvl)
        for _ in range(40):
            w = rng.uniform(-12, 12, 3).astype(np.float32)
            m = o.map_coords_pose(lvl, w)
            assert np.array_equal(bits(g.getMapCoordsPose(lvl, w)), bits(m))
            assert np.array_equal(bits(g.getWorldCoordsPose(lvl, m)), bits(o.world_coords_pose(lvl, m)))


def test_map_update_bit_exact(pyr, pyramid_scene):
    """80 scans x 3 levels of updateByScan: log-odds planes and update stamps identical to the oracle"""
    g, o = pyr
    for lvl in range(pyramid_scene.levels):
        lo_g, ui_g = g.download_level(lvl)
        lo_o, ui_o = o.download_level(lvl)
        assert (lo_o > 0).sum() > 100
        assert np.array_equal(ui_g, ui_o), f"level {lvl}: {(ui_g != ui_o).sum()} stamps differ"
        assert np.array_equal(bits(lo_g), bits(lo_o)), f"level {lvl}: {(bits(lo_g) != bits(lo_o)).sum()} cells differ"
        assert g.getUpdateIndex(lvl) == len(pyramid_scene.build_scans) - 1


def test_probability_plane_bit_exact(pyr, pyramid_scene, oracle_mod):
    """p = e^l / (e^l + 1) with glibc's expf: every cell of every level equals the host's value bit for bit, and
    equals what the reference's own sampler returns at integer coordinates"""
    g, o = pyr
    for lvl in range(pyramid_scene.levels):
        lo, _ = o.download_level(lvl)
        got = g.download_prob(lvl)
        _, prob = oracle_mod.libm_expf(lo.reshape(-1), o.kind)
        assert np.array_equal(bits(got).reshape(-1), bits(prob))
        # at integer coordinates interpMapValueWithDerivatives returns M = P(ix, iy) exactly (fractions are 0)
        ys, xs = np.nonzero(lo[:-2, :-2] != 0)
        sel = np.random.default_rng(3).choice(len(xs), size=min(20000, len(xs)), replace=False)
        ref = o.interp(lvl, np.stack([xs[sel], ys[sel]], 1).astype(np.float32))[:, 0]
        assert np.array_equal(bits(got[ys[sel], xs[sel]]), bits(ref))


def test_device_expf_equals_host_libm(sml, oracle_mod):
    """device expf over the whole log-odds range and far beyond (overflow / underflow / denormal results / NaN)"""
    g, o = sml
    rng = np.random.default_rng(22)
    x = np.concatenate([rng.uniform(-60, 60, 2_000_000), rng.uniform(-110, 95, 1_000_000), rng.normal(0, 1e-3, 100_000),
                        [0.0, -0.0, 50.0, 88.0, 88.72, 88.73, 89.0, -87.3, -87.4, -100.0, -103.2, -103.3, -103.97, -103.98,
                         -104.0, 1e-30, -1e-30, 1e38, -1e38, np.inf, -np.inf]]).astype(np.float32)
    e, p = g.debug_expf(x)
    eh, ph = oracle_mod.libm_expf(x, o.kind)
    assert np.array_equal(bits(e), bits(eh))
    fin = np.isfinite(eh)
    assert np.array_equal(bits(p[fin]), bits(ph[fin]))
    assert np.isnan(g.debug_expf(np.array([np.nan], np.float32))[0]).all()


def test_upload_then_download_roundtrip_and_rebuild(capi, pyr, pyramid_scene):
    g, o = pyr
    g2 = make_gpu(capi, pyramid_scene, build=False)
    for lvl in range(pyramid_scene.levels):
        lo, ui = o.download_level(lvl)
        g2.upload_level(lvl, lo, ui)
        lo2, ui2 = g2.download_level(lvl)
        assert np.array_equal(bits(lo2), bits(lo)) and np.array_equal(ui2, ui)
        # texels rebuilt from an uploaded plane == texels maintained incrementally by the updates
        assert np.array_equal(bits(g2.download_prob(lvl)), bits(g.download_prob(lvl)))
    sc = pyramid_scene
    a = g.matchData(sc.query_init[0], sc.query_scans[0])
    b = g2.matchData(sc.query_init[0], sc.query_scans[0])
    assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1]))


# ---------------------------------------------------------------- per-beam / per-step
def test_device_sincos_equals_host_libm(sml, oracle_mod):
    """device sincosf == the host's sincosf (glibc, what the reference's sin(pose[2]) / cos(pose[2]) compile to) for
    3M angles in all three argument ranges of the algorithm (|x| < pi/4, < 120, beyond) and the special values"""
    g, o = sml
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.uniform(-3.2, 3.2, 1_000_000), rng.uniform(-100, 100, 1_000_000),
                        rng.uniform(-1e5, 1e5, 800_000), rng.uniform(-1e9, 1e9, 100_000), rng.normal(0, 1e-3, 100_000),
                        [0.0, -0.0, np.pi, -np.pi, np.pi / 2, np.pi / 4, 0.78539, 0.7854, 1e-30, 2.44e-4, 2.45e-4, 119.99,
                         120.0, 120.01, 1048575.9, 1048576.0, 3e7, 1e30, 3.4e38, -3.4e38]]).astype(np.float32)
    s, c = g.debug_sincos(x)
    sh, ch = oracle_mod.libm_sincosf(x, o.kind)
    assert np.array_equal(bits(s), bits(sh))
    assert np.array_equal(bits(c), bits(ch))
    s2, c2 = g.debug_sincos(np.array([np.inf, -np.inf, np.nan], np.float32))
    assert np.isnan(s2).all() and np.isnan(c2).all()


@pytest.mark.parametrize("layout", ["quad", "plane"])
def test_per_beam_terms_bit_exact(capi, oracle_mod, pyramid_scene, layout, kind):
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = make_gpu(capi, sc, build=False, layout=capi.LAYOUT_QUAD if layout == "quad" else capi.LAYOUT_PLANE)
    for lvl in range(sc.levels):
        g.upload_level(lvl, *o.download_level(lvl))
    checked = 0
    for q in range(len(sc.query_scans)):
        for lvl in range(sc.levels):
            pts = sc.query_scans[q] * np.float32(1.0 / 2 ** lvl)
            pm = o.map_coords_pose(lvl, sc.query_init[q])
            got = g.eval_beams(lvl, pm, pts)
            # oracle per-beam terms: M, gx, gy from interp at the transformed point; the transform
            # uses the same fp32 expression t + (c*x + (-s)*y), sin/cos from the host libm
            s, c = (v[0] for v in oracle_mod.libm_sincosf(pm[2:3], kind))
            tx = pm[0] + (c * pts[:, 0] + (-s) * pts[:, 1])
            ty = pm[1] + (s * pts[:, 0] + c * pts[:, 1])
            ref = o.interp(lvl, np.stack([tx, ty], 1).astype(np.float32))
            assert np.array_equal(bits(got[:, :3]), bits(ref)), (q, lvl)
            rot = ((-s * pts[:, 0] - c * pts[:, 1]) * ref[:, 1] + (c * pts[:, 0] - s * pts[:, 1]) * ref[:, 2])
            assert np.array_equal(bits(got[:, 3]), bits(rot.astype(np.float32))), (q, lvl)
            checked += 1
    assert checked == len(sc.query_scans) * sc.levels


def test_hessian_derivs_match_oracle(pyr, pyramid_scene):
    """H and dTr of one evaluation: same per-beam terms, different summation order -> ~1e-6 relative"""
    g, o = pyr
    sc = pyramid_scene
    for q in range(8):
        for lvl in range(sc.levels):
            pts = sc.query_scans[q] * np.float32(1.0 / 2 ** lvl)
            pm = o.map_coords_pose(lvl, sc.query_init[q])
            Hg, dg = g.hessian_derivs(lvl, pm, pts)
            Ho, do = o.hessian_derivs(lvl, pm, pts)
            scale = np.abs(Ho).max()
            assert np.abs(Hg - Ho).max() <= 2e-5 * scale
            assert_dtr_close(dg, do, Ho, pts.shape[0])
            assert np.array_equal(Hg, Hg.T)


# ---------------------------------------------------------------- pose parity
def test_config1_single_level_pose_parity(sml, small_scene):
    """BASELINE configs[0]: 181 beams, 256x256 single-res map, 5 GN iterations"""
    g, o = sml
    sc = small_scene
    for q in range(len(sc.query_scans)):
        pg, cg = g.match_level(0, sc.query_init[q], sc.query_scans[q], 5)
        po, co = o.match_level(0, sc.query_init[q], sc.query_scans[q], 5)
        assert_pose_close(pg, po, f"q{q}")
        assert np.abs(cg - co).max() <= 1e-4 * np.abs(co).max()


def test_config2_pyramid_pose_parity(pyr, pyramid_scene):
    """configs[1]-shaped: 1081 beams, 3-level pyramid, full matchData (4+4+6 GN steps)"""
    g, o = pyr
    sc = pyramid_scene
    worst = (0.0, 0.0)
    for q in range(len(sc.query_scans)):
        pg, cg = g.matchData(sc.query_init[q], sc.query_scans[q])
        po, co = o.match(sc.query_init[q], sc.query_scans[q])
        d = assert_pose_close(pg, po, f"q{q}")
        worst = (max(worst[0], d[0]), max(worst[1], d[1]))
        assert np.abs(cg - co).max() <= 1e-4 * np.abs(co).max()
    print(f"worst pose deviation vs oracle: {worst[0]:.2e} m, {worst[1]:.2e} rad")


def test_empty_scan_and_degenerate_map(capi, pyr, pyramid_scene):
    g, o = pyr
    sc = pyramid_scene
    cov_in = np.arange(9, dtype=np.float32)
    p, c = g.matchData(sc.query_init[0], np.zeros((0, 2), np.float32), cov_in)
    assert np.array_equal(bits(p), bits(sc.query_init[0])) and np.array_equal(c, cov_in)
    # fresh map: p = 0.5 everywhere -> H(0,0) == 0 -> the pose must not move
    fresh = make_gpu(capi, sc, build=False)
    p, c = fresh.matchData(sc.query_init[1], sc.query_scans[1])
    fo = o.__class__(o.kind, sc.resolution, sc.map_size, sc.map_size, sc.levels)
    po, co = fo.match(sc.query_init[1], sc.query_scans[1])
    assert np.array_equal(bits(p), bits(po)) and np.array_equal(c, co) and not c.any()


def test_far_starts_clamp_and_out_of_map_beams(pyr, pyramid_scene):
    """large initial errors (angle clamp active) and a start near the map border (most beams OOB).
    GN from far outside the basin is chaotic, so only starts the oracle itself converges are held
    to the pose tolerance; all must stay finite and agree on the first coarse step."""
    g, o = pyr
    sc = pyramid_scene
    rng = np.random.default_rng(7)
    for q in range(8):
        init = sc.query_truth[q] + np.array([rng.uniform(-.5, .5), rng.uniform(-.5, .5), rng.uniform(-.5, .5)], np.float32)
        lvl = sc.levels - 1
        pts = sc.query_scans[q] * np.float32(1.0 / 2 ** lvl)
        pg, _ = g.match_level(lvl, init, pts, 0)  # exactly one GN step on the coarsest level
        po, _ = o.match_level(lvl, init, pts, 0)
        assert_pose_close(pg, po, f"first step q{q}")
        pg, _ = g.matchData(init, sc.query_scans[q])
        assert np.isfinite(pg).all()
    # start near the map border: ~80 % of the beams fall outside the map and contribute exact zeros.
    # What is left is (nearly) rank deficient -- cond(H) ~ 1e10, so the reference's OWN fp32 solve is
    # rounding noise there (it jumps 2.5 m) and the pose cannot be a parity target.  The out-of-map
    # handling itself is checked where it is well defined: per-beam terms bit-exact, H/dTr to
    # summation-order accuracy, and the step stays finite.
    far = np.array([11.5, 9.0, 0.3], np.float32)
    pm = o.map_coords_pose(0, far)
    pts = sc.query_scans[0]
    from oracle import pyoracle
    s_, c_ = (v[0] for v in pyoracle.libm_sincosf(pm[2:3], o.kind))
    tx = pm[0] + (c_ * pts[:, 0] + (-s_) * pts[:, 1])
    ty = pm[1] + (s_ * pts[:, 0] + c_ * pts[:, 1])
    ref = o.interp(0, np.stack([tx, ty], 1).astype(np.float32))
    oob = (tx < 0) | (tx > sc.map_size - 2) | (ty < 0) | (ty > sc.map_size - 2)
    assert 0.5 < oob.mean() < 0.95 and not ref[oob].any()
    got = g.eval_beams(0, pm, pts)
    assert np.array_equal(bits(got[:, :3]), bits(ref))
    Hg, dg = g.hessian_derivs(0, pm, pts)
    Ho, do = o.hessian_derivs(0, pm, pts)
    assert np.abs(Hg - Ho).max() <= 2e-5 * np.abs(Ho).max()
    assert_dtr_close(dg, do, Ho, pts.shape[0])
    assert np.linalg.cond(Ho.astype(np.float64)) > 1e8
    pg, _ = g.match_level(0, far, pts, 0)
    assert np.isfinite(pg).all()


# ---------------------------------------------------------------- batched path
def test_batch_equals_singles_and_is_deterministic(capi, oracle_mod, pyramid_scene, kind):
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g1 = make_gpu(capi, sc, build=False, waves_per_scan=1)
    for lvl in range(sc.levels):
        g1.upload_level(lvl, *o.download_level(lvl))
    pts, offs = synth.pack_scans(sc.query_scans)
    pb, cb = g1.match_batch(sc.query_init, pts, offs)
    pb2, cb2 = g1.match_batch(sc.query_init, pts, offs)
    assert np.array_equal(bits(pb), bits(pb2)) and np.array_equal(bits(cb), bits(cb2))  # deterministic
    for q in range(len(sc.query_scans)):
        ps, cs = g1.matchData(sc.query_init[q], sc.query_scans[q])  # same WPS -> same reduction tree
        assert np.array_equal(bits(ps), bits(pb[q])) and np.array_equal(bits(cs), bits(cb[q]))
        po, _ = o.match(sc.query_init[q], sc.query_scans[q])
        assert_pose_close(pb[q], po, f"batch q{q}")
    # order independence: reversing the batch reverses the results bit-for-bit
    rev = list(reversed(range(len(sc.query_scans))))
    pts_r, offs_r = synth.pack_scans([sc.query_scans[i] for i in rev])
    pr, _ = g1.match_batch(sc.query_init[rev], pts_r, offs_r)
    assert np.array_equal(bits(pr), bits(pb[rev]))
    # shared-scan mode (pose hypotheses of ONE scan) == CSR with the scan replicated
    hyp = np.repeat(sc.query_init[3:4], 9, 0) + np.linspace(-0.05, 0.05, 9, dtype=np.float32)[:, None]
    ph, _ = g1.match_batch(hyp, sc.query_scans[3], None)
    pts_c, offs_c = synth.pack_scans([sc.query_scans[3]] * 9)
    pc, _ = g1.match_batch(hyp, pts_c, offs_c)
    assert np.array_equal(bits(ph), bits(pc))


@pytest.mark.parametrize("wps", [1, 2, 4, 8, 16])
def test_every_team_width_meets_pose_tolerance(capi, oracle_mod, pyramid_scene, wps, kind):
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = make_gpu(capi, sc, build=False, waves_per_scan=wps)
    for lvl in range(sc.levels):
        g.upload_level(lvl, *o.download_level(lvl))
    for q in range(6):
        pg, _ = g.matchData(sc.query_init[q], sc.query_scans[q])
        assert g.last_launch_config()["waves_per_scan"] == wps
        assert_pose_close(pg, o.match(sc.query_init[q], sc.query_scans[q])[0], f"wps{wps} q{q}")
    # ragged scan lengths around the team width: n beams spread evenly over the fan so the
    # system stays well conditioned (1- and 2-beam scans make H singular: the reference then
    # feeds NaN coordinates into an int cast and crashes, so only the GPU is run on those)
    full = sc.query_scans[2]
    for n in (63, 64, 65, 64 * wps - 1, 64 * wps + 1, 1000):
        pts = full[np.linspace(0, full.shape[0] - 1, n).astype(int)]
        pg, _ = g.matchData(sc.query_init[2], pts)
        po, _ = o.match(sc.query_init[2], pts)
        assert_pose_close(pg, po, f"wps{wps} n={n}")
    for n in (1, 2):
        g.matchData(sc.query_init[2], full[:n])  # must not fault


# ---------------------------------------------------------------- processor loop
def test_slam_loop_match_update_interleaved(capi, oracle_mod, pyramid_scene, kind):
    """HectorSlamProcessor::update from an EMPTY map: match -> threshold -> updateByScan, 25 scans.
    Poses stay within tolerance of the oracle's at every step; the maps agree except for the few
    cells whose Bresenham endpoints flip because the poses differ in the last bits."""
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc, build=False)
    o.proc_set_thresholds(0.05, 0.02)
    p = capi.HectorSlamProcessor(sc.resolution, sc.map_size, sc.map_size, (0.5, 0.5), sc.levels)
    p.setUpdateFactorFree(0.4)
    p.setUpdateFactorOccupied(0.9)
    p.setMapUpdateMinDistDiff(0.05)
    p.setMapUpdateMinAngleDiff(0.02)
    origo = np.array([0.3, -0.1], np.float32) * np.float32(sc.scale_to_map)
    hint_o = hint_g = sc.build_poses[0].copy()
    for t in range(25):
        mwm = t == 7
        o.proc_update(sc.build_scans[t], hint_o, origo=origo, map_without_matching=mwm)
        p.update(sc.build_scans[t], hint_g, mwm, origo=origo)
        po, _ = o.proc_last_pose()
        pg = p.getLastScanMatchPose()
        assert_pose_close(pg, po, f"t={t}")
        step = sc.build_poses[t + 1] - sc.build_poses[t]
        hint_o, hint_g = po + step, pg + step
    for lvl in range(sc.levels):
        lo_g, ui_g = p.mapRep.download_level(lvl)
        lo_o, ui_o = o.download_level(lvl)
        touched = (ui_o >= 0).sum()
        assert touched > 1000
        assert (bits(lo_g) != bits(lo_o)).sum() <= 0.002 * touched
    # with identical poses fed to both, the maps are bit-identical (pure index work)
    g2 = make_gpu(capi, sc, build=False)
    o2 = make_oracle(oracle_mod, kind, sc, build=False)
    for t in range(10):
        o2.match(sc.build_poses[t], sc.build_scans[t], origo)       # retains the coarse containers
        g2.matchData(sc.build_poses[t], sc.build_scans[t], None, origo)
        o2.update_by_scan(sc.build_poses[t], sc.build_scans[t], origo)
        g2.updateByScan(sc.build_scans[t], sc.build_poses[t], origo)
    for lvl in range(sc.levels):
        a, b = g2.download_level(lvl), o2.download_level(lvl)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), lvl


def test_update_edge_cases_bit_exact(capi, oracle_mod, small_scene, kind):
    """empty scan, begin == end beams, beams leaving the map, robot outside the map, the occupied
    clamp at 50 and the free->occupied revert -- all bit-exact against the oracle"""
    sc = small_scene
    g = make_gpu(capi, sc, build=False)
    o = make_oracle(oracle_mod, kind, sc, build=False)
    s = np.float32(sc.scale_to_map)
    cases = [
        (np.array([0, 0, 0], np.float32), np.zeros((0, 2), np.float32)),                       # empty
        (np.array([0, 0, 0.3], np.float32), np.array([[0.2, 0.1], [0.4, -0.3]], np.float32)),  # begin == end
        (np.array([10, 0, 0], np.float32), np.array([[5 * s, 0], [2 * s, 1 * s], [1 * s, 40 * s]], np.float32)),  # leaves map
        (np.array([40, 40, 0], np.float32), np.array([[-30 * s, -30 * s]], np.float32)),       # robot outside
    ]
    for pose, pts in cases:
        g.update_by_scan_level(0, pose, pts)
        o.update_by_scan_level(0, pose, pts)
    # hammer one wall until cells saturate at the 50 clamp; crossing beams exercise the revert
    ring = np.stack([np.cos(np.linspace(0, 2 * np.pi, 720, endpoint=False)),
                     np.sin(np.linspace(0, 2 * np.pi, 720, endpoint=False))], 1).astype(np.float32) * (3 * s)
    for k in range(40):
        pose = np.array([0.01 * k, -0.02 * k, 0.05 * k], np.float32)
        g.update_by_scan_level(0, pose, ring)
        o.update_by_scan_level(0, pose, ring)
    lo_g, ui_g = g.download_level(0)
    lo_o, ui_o = o.download_level(0)
    assert lo_o.max() >= 50.0
    assert np.array_equal(ui_g, ui_o) and np.array_equal(bits(lo_g), bits(lo_o))
    assert g.getUpdateIndex(0) == len(cases) + 40 - 1
    bb = g.last_update_bbox(0)
    ys, xs = np.nonzero(ui_o == ui_o.max())
    assert bb[0] <= xs.min() and bb[2] >= xs.max() and bb[1] <= ys.min() and bb[3] >= ys.max()
    rows = g.download_rows(0, int(bb[1]), int(bb[3]) + 1)
    assert np.array_equal(bits(rows), bits(lo_o[bb[1]:bb[3] + 1]))
    # reset: every cell back to (0, -1), probability 0.5
    g.reset()
    lo_g, ui_g = g.download_level(0)
    assert not lo_g.any() and (ui_g == -1).all() and (g.download_prob(0) == 0.5).all()


def test_multi_level_update_at_the_low_map_edge_is_bit_exact(capi, oracle_mod, pyramid_scene, kind):
    """Beams (and the begin cell) just outside the LOW x / y edge of level 0 are dropped there but valid on the coarser
    levels: (int) truncates towards zero, so level k keeps an end point e (level-0 cell units) with e > -1.5 * 2^k.  The
    multi-level updateByScan must not derive the coarse levels' apply boxes from level 0's box then (round-2 advisor
    finding): every level bit-exact against the oracle, through the usual match -> update flow."""
    sc = pyramid_scene
    g = make_gpu(capi, sc, build=False)
    o = make_oracle(oracle_mod, kind, sc, build=False)
    res, half = sc.resolution, sc.map_size // 2
    rng = np.random.default_rng(5)
    cases = []
    for cx, cy, th in [(6.0, 9.0, 0.0), (5.2, 30.0, 0.3), (40.0, 4.4, -0.2), (-1.2, 7.0, 0.0), (7.0, -2.4, 0.1), (-2.0, -2.0, 0.0)]:
        # robot at level-0 cell (cx, cy): the last three begin OFF the map at level 0 (cell -1 / -2) but inside at level 1 / 2
        pose = np.array([(cx - half) * res, (cy - half) * res, th], np.float32)
        ex = np.concatenate([np.linspace(-7.0, 4.0, 45), rng.uniform(-7.0, 60.0, 60)])
        ey = np.concatenate([rng.uniform(-7.0, 60.0, 45), np.linspace(-7.0, 4.0, 60)])
        c, s_ = np.cos(-th), np.sin(-th)
        dx, dy = ex - cx, ey - cy
        pts = np.stack([c * dx - s_ * dy, s_ * dx + c * dy], 1).astype(np.float32)  # robot frame, level-0 cell units
        cases.append((pose, pts))
    origo = np.zeros(2, np.float32)
    for pose, pts in cases:
        o.match(pose, pts, origo)  # retains the coarse containers (MapRepMultiMap.h:143)
        g.matchData(pose, pts, None, origo)
        o.update_by_scan(pose, pts, origo)
        g.updateByScan(pts, pose, origo)
    for lvl in range(sc.levels):
        a, b = g.download_level(lvl), o.download_level(lvl)
        assert (b[1] >= 0).sum() > 50, lvl
        assert np.array_equal(a[1], b[1]) and np.array_equal(bits(a[0]), bits(b[0])), lvl
        # the probability plane (and with it the texels) was rewritten wherever a cell changed
        _, prob = oracle_mod.libm_expf(b[0].reshape(-1), o.kind)
        assert np.array_equal(bits(g.download_prob(lvl)).reshape(-1), bits(prob)), lvl


# ---------------------------------------------------------------- golden vectors
def test_golden_config1(capi):
    g1 = np.load(os.path.join(GOLD, "config1_181beam_256map.npz"))
    g = capi.MapRepMultiMap(float(g1["resolution"]), int(g1["map_size"]), int(g1["map_size"]), 1)
    g.upload_level(0, g1["logodds"], g1["update_index"])
    for q in range(4):
        pts = g1[f"q{q}_pts"]
        pose, cov = g.match_level(0, g1[f"q{q}_init"], pts, 5)
        assert_pose_close(pose, g1[f"q{q}_pose"], f"golden q{q}")
        assert np.abs(cov - g1[f"q{q}_cov"]).max() <= 1e-4 * np.abs(g1[f"q{q}_cov"]).max()
        for k in range(7):
            H, d = g.hessian_derivs(0, g1[f"q{q}_step_pose_map"][k], pts)
            Hr, dr = g1[f"q{q}_step_H"][k], g1[f"q{q}_step_dTr"][k]
            assert np.abs(H - Hr).max() <= 2e-5 * np.abs(Hr).max()
            assert_dtr_close(d, dr, Hr, pts.shape[0])


def test_golden_pyramid(capi):
    g2 = np.load(os.path.join(GOLD, "pyramid_1081beam_512map.npz"))
    g = capi.MapRepMultiMap(float(g2["resolution"]), int(g2["map_size"]), int(g2["map_size"]), 3)
    for lvl in range(3):
        g.upload_level(lvl, g2[f"logodds{lvl}"], g2[f"update_index{lvl}"])
    for q in range(8):
        pose, cov = g.matchData(g2["init"][q], g2[f"q{q}_pts"])
        assert_pose_close(pose, g2["pose"][q], f"golden pyramid q{q}")
        assert np.abs(cov - g2["cov"][q]).max() <= 1e-4 * np.abs(g2["cov"][q]).max()


# ---------------------------------------------------------------- ABI robustness
def test_capi_error_paths_fail_loudly(capi, small_scene):
    """bad arguments come back as negative status codes with a message, never a crash or a silent no-op"""
    import ctypes as C
    lib = capi.load_library()
    sc = small_scene
    g = make_gpu(capi, sc, build=False)
    f3 = np.zeros(3, np.float32)
    f9 = np.zeros(9, np.float32)
    pts = np.zeros((4, 2), np.float32)
    assert lib.hsm_match(g._h, f3, pts.ctypes.data, -1, np.zeros(2, np.float32), f3, f9) == -1
    assert lib.hsm_match(g._h, f3, None, 4, np.zeros(2, np.float32), f3, f9) == -1
    assert b"bad argument" in lib.hsm_last_error()
    assert lib.hsm_level_info(g._h, 7, None, None, None, None) == -1 and b"level" in lib.hsm_last_error()
    assert lib.hsm_update_by_scan_level(g._h, -1, f3, pts.ctypes.data, 4, np.zeros(2, np.float32)) == -1
    big = np.zeros((1_048_576, 2), np.float32)  # HSM_MAX_UPDATE_BEAMS + 1
    assert lib.hsm_update_by_scan(g._h, f3, big.ctypes.data, big.shape[0], np.zeros(2, np.float32)) == -4  # HSM_ERR_TOO_LARGE
    h = C.c_void_p()
    opts = capi.HsmOpts(-1, 0, 0)
    assert lib.hsm_create(0.05, 64, 64, 9, 0.5, 0.5, C.byref(opts), C.byref(h)) == -1      # > HSM_MAX_LEVELS
    assert lib.hsm_create(0.05, 64, 64, 7, 0.5, 0.5, C.byref(opts), C.byref(h)) == -1      # coarsest level would be 1x1
    assert lib.hsm_create(-1.0, 64, 64, 1, 0.5, 0.5, C.byref(opts), C.byref(h)) == -1
    assert lib.hsm_create(0.05, 20000, 20000, 1, 0.5, 0.5, C.byref(opts), C.byref(h)) == -4  # > 2^28 cells
    assert lib.hsm_create(0.05, 64, 64, 1, 0.5, 0.5, C.byref(capi.HsmOpts(99, 0, 0)), C.byref(h)) == -1  # no such device
    assert lib.hsm_create(0.05, 64, 64, 1, 0.5, 0.5, C.byref(capi.HsmOpts(-1, 0, 3)), C.byref(h)) == -1  # waves_per_scan
    with pytest.raises(capi.HsmError):
        g.match_ingested(f3)  # nothing ingested yet
    # after all that the context still works
    p, _ = g.matchData(sc.query_init[0], sc.query_scans[0])
    assert np.isfinite(p).all()


def test_concurrent_callers_are_serialised(capi, oracle_mod, pyramid_scene, kind):
    """the reference contract is one writer + one reader thread; the context's mutex must keep concurrent
    matchData / occupancy reads on ONE context, and independent contexts in parallel, all correct"""
    import threading
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = make_gpu(capi, sc)
    g2 = make_gpu(capi, sc)
    expect = [o.match(sc.query_init[q], sc.query_scans[q])[0] for q in range(8)]
    errs = []

    def matcher(ctx, reps):
        for _ in range(reps):
            for q in range(8):
                p, _ = ctx.matchData(sc.query_init[q], sc.query_scans[q])
                d = np.abs(p.astype(np.float64) - expect[q])
                if d[0] > POSE_TOL_M or d[1] > POSE_TOL_M or d[2] > POSE_TOL_RAD:
                    errs.append((q, p))

    def reader(ctx, reps):
        ref = o.occupancy_grid(0)
        for _ in range(reps):
            if not np.array_equal(ctx.occupancy_grid(0), ref):
                errs.append("grid")

    ts = [threading.Thread(target=matcher, args=(g, 6)), threading.Thread(target=matcher, args=(g, 6)),
          threading.Thread(target=reader, args=(g, 20)), threading.Thread(target=matcher, args=(g2, 6))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:3]


def test_randomised_geometries(capi, oracle_mod, kind):
    """odd map sizes (partial edge rows, cells % 4 != 0), 1..4 levels, off-centre start coordinates, random update
    factors, rooms larger than the map.  Every step matches the same scan against IDENTICAL maps on both sides
    (both maps are then updated with the oracle's pose, and stay bit-identical on every level to the end).
    No convergence predicate: HSM_PARITY_EXACT equals the reference on EVERY step, pose and covariance, bit for bit --
    also on the young, tiny maps of the first steps, where Gauss-Newton has not settled and the result is a chaotic
    function of the last bits; the default (fast) summation is MEASURED against the exact mode on the same context, with
    the bound stated below."""
    from hector_slam_amd import synth
    rng = np.random.default_rng(20240925)
    steps_total = fast_within = 0
    fast_worst = 0.0
    for trial in range(6):
        size = int(rng.choice([96, 125, 250, 333, 512]))
        levels = int(rng.integers(1, 5))
        while (size >> (levels - 1)) < 8:
            levels -= 1
        res = float(rng.choice([0.05, 0.1, 0.2]))
        start = (float(rng.uniform(0.3, 0.7)), float(rng.uniform(0.3, 0.7)))
        free, occ = float(rng.uniform(0.3, 0.49)), float(rng.uniform(0.55, 0.95))
        ext = size * res
        grow = 1.15 if trial % 2 else 0.6
        world = synth.World.make(ext * grow, ext * grow * 0.75, n_boxes=4, seed=int(rng.integers(1 << 30)), keep_clear=0.5)
        s = float(np.float32(1.0) / np.float32(res))
        n_beams = int(rng.choice([181, 400, 1081]))
        poses = synth.loop_trajectory(world, 14, frac=0.25).astype(np.float32)
        poses[:, 0] += (0.5 - start[0]) * ext * 0.3
        noise = np.random.default_rng(trial)
        scans = [synth.make_scan(world, p, n_beams, s, noise, range_max=min(30.0, ext)) for p in poses]
        origos = rng.uniform(-2, 2, (14, 2)).astype(np.float32)

        o = oracle_mod.Oracle(kind, res, size, size, levels, start)
        g = capi.MapRepMultiMap(res, size, size, levels, start)
        o.set_update_factor_free(free)
        g.setUpdateFactorFree(free)
        o.set_update_factor_occupied(occ)
        g.setUpdateFactorOccupied(occ)
        for lvl in range(levels):
            assert g.level_info(lvl) == o.level_info(lvl)
        pose = poses[0].copy()
        for t in range(14):
            hint = pose + (poses[t] - poses[max(t - 1, 0)])
            po, co = o.match(hint, scans[t], origos[t])
            g.set_parity(capi.PARITY_EXACT)
            px, cx = g.matchData(hint, scans[t], None, origos[t])
            g.set_parity(capi.PARITY_FAST)
            assert np.array_equal(bits(px), bits(po)) and np.array_equal(bits(cx), bits(co)), \
                f"exact mode: trial {trial} size {size} res {res} levels {levels} t={t}"
            pg, cg = g.matchData(hint, scans[t], None, origos[t])  # (the retained containers are the same scan either way)
            assert np.isfinite(pg).all()
            steps_total += 1
            # north_star's 1e-4 m is 2e-3 level-0 cells at its 0.05 m resolution; on the coarser random maps here
            # (0.1 / 0.2 m cells) the same 2e-3 cells are 2e-4 / 4e-4 m
            d = np.abs(pg.astype(np.float64) - px)
            tol_m = max(POSE_TOL_M, 2e-3 * res)
            fast_within += bool(d[0] <= tol_m and d[1] <= tol_m and ang_diff(pg[2], px[2]) <= POSE_TOL_RAD)
            fast_worst = max(fast_worst, float(d[:2].max()))
            o.update_by_scan(po, scans[t], origos[t])
            o.on_map_updated()
            g.updateByScan(scans[t], po, origos[t])
            pose = po
        for lvl in range(levels):
            a, b = g.download_level(lvl), o.download_level(lvl)
            assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), (trial, lvl)
            assert np.array_equal(g.occupancy_grid(lvl), o.occupancy_grid(lvl))
    # fast mode against the exact mode (= the reference) over all 84 steps, young maps included.  Measured on MI355X:
    # see the printed line; the bounds leave a margin for the summation tree's sensitivity to the launch shape
    print(f"fast vs exact: {fast_within}/{steps_total} within tolerance, worst {fast_worst:.2e} m")
    assert fast_worst <= FAST_RANDOM_WORST_M and fast_within >= FAST_RANDOM_WITHIN * steps_total, (fast_within, steps_total, fast_worst)


def test_dense_scan_cooperative_matcher(capi, oracle_mod, pyramid_scene, kind):
    """single scans >= 4096 beams take the multi-workgroup cooperative matcher (gn_match_coop_kernel): same poses
    as the one-workgroup kernel and as the oracle, for both layouts, incl. the hook trace"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    s = float(np.float32(1.0) / np.float32(sc.resolution))
    rng = np.random.default_rng(77)
    for layout in (capi.LAYOUT_QUAD, capi.LAYOUT_PLANE):
        g = make_gpu(capi, sc, build=False, layout=layout)
        g16 = make_gpu(capi, sc, build=False, layout=layout, waves_per_scan=16)  # forces the one-workgroup kernel
        for lvl in range(sc.levels):
            lo, ui = o.download_level(lvl)
            g.upload_level(lvl, lo, ui)
            g16.upload_level(lvl, lo, ui)
        for n_beams in (4096, 8192, 16384):
            for q in range(3):
                truth = sc.query_truth[q]
                pts = synth.make_scan(sc.world, truth, n_beams, s, rng)
                init = sc.query_init[q]
                pc, cc = g.matchData(init, pts)
                cfg = g.last_launch_config()
                assert cfg["waves_per_scan"] < 0 and cfg["grid"] == min((pts.shape[0] + 255) // 256, 64), cfg
                p1, c1 = g16.matchData(init, pts)
                assert g16.last_launch_config()["waves_per_scan"] == 16
                po, co = o.match(init, pts)
                assert_pose_close(pc, po, f"coop vs oracle n={n_beams} q={q}")
                assert_pose_close(pc, p1, f"coop vs one-workgroup n={n_beams} q={q}")
                assert np.abs(cc - co).max() <= 1e-4 * np.abs(co).max()
    # hook trace through the cooperative kernel: 14 records, last H == returned cov
    import ctypes as C
    lib = capi.load_library()
    pts = np.ascontiguousarray(synth.make_scan(sc.world, sc.query_truth[0], 8192, s, rng), np.float32)
    trace = np.zeros(14 * 12, np.float32)
    pose, cov, steps = np.zeros(3, np.float32), np.zeros(9, np.float32), C.c_int()
    capi._check(lib.hsm_match_trace(g._h, sc.query_init[0], pts.ctypes.data, pts.shape[0], np.zeros(2, np.float32), pose,
                                    cov, trace, 14, C.byref(steps)), "hsm_match_trace")
    assert steps.value == 14 and np.array_equal(bits(trace[13 * 12 + 3:14 * 12]), bits(cov))
    po, _ = o.match(sc.query_init[0], pts)
    assert_pose_close(pose, po, "trace path")
    # the grid barrier's arrival counter is monotonic across launches and wraps at 2^32: matches that straddle the
    # wrap (8 workgroups x 14 steps = 112 arrivals per launch) give the same bits as before it
    ref = [g.matchData(sc.query_init[q], pts) for q in range(3)]
    g.debug_set_coop_barrier(0xffffffff - 150)
    for rep in range(4):
        for q in range(3):
            pq, cq = g.matchData(sc.query_init[q], pts)
            assert np.array_equal(bits(pq), bits(ref[q][0])) and np.array_equal(bits(cq), bits(ref[q][1])), (rep, q)


def test_default_parity_mode_is_the_reference_order_on_every_entry_point(capi, oracle_mod, pyramid_scene, kind, monkeypatch):
    """HSM_PARITY_AUTO (what a context starts in): batched matches, single-scan matchData (with and without the hook trace),
    single-level matches and the Hessian probe all run in the reference's summation order -- poses, covariances, H and dTr
    bit-identical to the reference (round 5; until round 4 single scans kept the tree); hsm_last_launch_parity says which
    order ran; HSM_PARITY accepts its four words only"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    monkeypatch.delenv("HSM_PARITY")
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    assert g.parity() == capi.PARITY_AUTO
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    pts, offs = synth.pack_scans(sc.query_scans)
    pb, cb = g.match_batch(sc.query_init, pts, offs)
    cfg = g.last_launch_config()
    assert cfg["parity"] == "auto" and cfg["parity_effective"] == "exact", cfg
    ref = [o.match(sc.query_init[q], sc.query_scans[q]) for q in range(len(sc.query_scans))]
    for q, (po, co) in enumerate(ref):
        assert np.array_equal(bits(pb[q]), bits(po)) and np.array_equal(bits(cb[q]), bits(co)), q
    p1, _ = g.match_batch(sc.query_init[:1], *synth.pack_scans(sc.query_scans[:1]))  # a batch of one is a batch
    assert g.last_launch_config()["parity_effective"] == "exact" and np.array_equal(bits(p1[0]), bits(pb[0]))
    for q, (po, co) in enumerate(ref):  # the reference's own entry point: one scan, MapRepMultiMap::matchData
        ps, cs = g.matchData(sc.query_init[q], sc.query_scans[q])
        assert g.last_launch_config()["parity_effective"] == "exact"
        assert np.array_equal(bits(ps), bits(po)) and np.array_equal(bits(cs), bits(co)), ("single scan", q, ps, po)
    import ctypes as C
    lib = capi.load_library()
    a = np.ascontiguousarray(sc.query_scans[0], np.float32)
    pt, ct, trace, nst = np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(14 * 12, np.float32), C.c_int(0)
    capi._check(lib.hsm_match_trace(g._h, sc.query_init[0], a.ctypes.data, a.shape[0], np.zeros(2, np.float32), pt, ct, trace, 14,
                                    C.byref(nst)), "hsm_match_trace")
    assert np.array_equal(bits(pt), bits(ref[0][0])) and np.array_equal(bits(ct), bits(ref[0][1])), "hook-trace form"
    pl, cl = g.match_level(0, sc.query_init[1], sc.query_scans[1], 5)
    pol, col = o.match_level(0, sc.query_init[1], sc.query_scans[1], 5)
    assert np.array_equal(bits(pl), bits(pol)) and np.array_equal(bits(cl), bits(col)), "single-level matchData"
    pm = o.map_coords_pose(0, ref[0][0])
    Hg, dg = g.hessian_derivs(0, pm, sc.query_scans[0])
    Ho, do = o.hessian_derivs(0, pm, sc.query_scans[0])
    assert np.array_equal(bits(Hg), bits(Ho)) and np.array_equal(bits(dg), bits(do)), "getCompleteHessianDerivs probe"
    g.set_parity(capi.PARITY_FAST)
    ps, _ = g.matchData(sc.query_init[0], sc.query_scans[0])
    assert g.last_launch_config()["parity_effective"] == "fast"
    assert_pose_close(ps, pb[0], "single scan (fast tree, opt-in) vs the default's exact result")
    g.close()
    for word, ok in (("exact", True), ("fast", True), ("relaxed", True), ("auto", True), ("Exact", False), ("1", False)):
        monkeypatch.setenv("HSM_PARITY", word)
        if ok:
            capi.MapRepMultiMap(sc.resolution, 64, 64, 1).close()
        else:
            with pytest.raises(capi.HsmError):
                capi.MapRepMultiMap(sc.resolution, 64, 64, 1)


def test_dense_matcher_exchange_forms_agree(capi, pyramid_scene, monkeypatch):
    """the multi-workgroup matcher exchanges its partial sums through tagged 16-byte records (default: an ordinary launch, no
    grid barrier) or through the counter grid barrier (HSM_COOP_TAGGED=0: a cooperative launch): same sums in the same
    order -> bit-identical poses and covariances, across the wrap of the barrier counter as well; and a match while a long
    batch kernel occupies the device on another stream still completes (the workgroups become co-resident as it retires)"""
    import torch
    from hector_slam_amd import synth
    sc = pyramid_scene
    s = float(np.float32(1.0) / np.float32(sc.resolution))
    rng = np.random.default_rng(78)
    scans = [synth.make_scan(sc.world, sc.query_truth[q], n, s, rng) for q, n in ((0, 4096), (1, 8192), (2, 16384))]
    got = {}
    for tagged in ("1", "0"):
        monkeypatch.setenv("HSM_COOP_TAGGED", tagged)
        g = make_gpu(capi, sc)
        if tagged == "0":
            g.debug_set_coop_barrier(0xffffffff - 700)
        got[tagged] = [g.matchData(sc.query_init[q], scans[q]) for q in range(3) for _ in range(2)]
        assert g.last_launch_config()["waves_per_scan"] < 0
        if tagged == "1":
            # a 4096-scan batch on a caller-owned stream right before the dense match: the device is busy when it launches
            dev = torch.device("cuda", 0)
            B = 4096
            pts, offs = synth.pack_scans([sc.query_scans[i % len(sc.query_scans)] for i in range(B)])
            init = np.repeat(sc.query_init, B // len(sc.query_init) + 1, 0)[:B]
            d_i, d_p, d_o = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (init, pts, offs))
            d_out = torch.zeros((B, 3), dtype=torch.float32, device=dev)
            side = torch.cuda.Stream(device=dev)
            for rep in range(3):
                for _ in range(4):
                    g.match_batch_device(B, d_i.data_ptr(), d_p.data_ptr(), d_o.data_ptr(), 1081, d_out.data_ptr(), 0, side.cuda_stream)
                pb, cb = g.matchData(sc.query_init[1], scans[1])
                assert np.array_equal(bits(pb), bits(got["1"][2][0])) and np.array_equal(bits(cb), bits(got["1"][2][1])), rep
            torch.cuda.synchronize()
        g.close()
    for a, b in zip(got["1"], got["0"]):
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(bits(a[1]), bits(b[1]))


def test_dense_matcher_exchange_timeout_falls_back_to_the_one_workgroup_matcher(capi, pyramid_scene):
    """a workgroup of the multi-workgroup dense matcher that never publishes its records (test hook; in the field: K
    workgroups that cannot become co-resident) makes every workgroup's exchange time out -- ONE bounded wait, then the launch
    ends -- and hsm_match matches the scan again on the one-workgroup matcher instead of returning an error (round-4
    advisor): a pose within the fast mode's bar, the fallback counted, and the exchange works again afterwards"""
    import time
    from hector_slam_amd import synth
    sc = pyramid_scene
    s = float(np.float32(1.0) / np.float32(sc.resolution))
    pts = synth.make_scan(sc.world, sc.query_truth[1], 8192, s, np.random.default_rng(79))
    g = make_gpu(capi, sc)
    p0, c0 = g.matchData(sc.query_init[1], pts)
    assert g.last_launch_config()["waves_per_scan"] < 0 and g.debug_coop_fallbacks() == 0
    g.debug_set_coop_mute(3)  # workgroup 2 stays silent
    t0 = time.perf_counter()
    p1, c1 = g.matchData(sc.query_init[1], pts)
    dt = time.perf_counter() - t0
    assert g.debug_coop_fallbacks() == 1
    assert g.last_launch_config()["waves_per_scan"] == 16, g.last_launch_config()  # one workgroup of 16 wavefronts
    assert_pose_close(p1, p0, "fallback vs the cooperative result")
    assert np.abs(c1 - c0).max() <= 1e-4 * np.abs(c0).max()
    assert dt < 60.0, dt  # one timeout, not one per remaining GN step
    print(f"exchange timeout + fallback: {dt:.2f} s")
    g.debug_set_coop_mute(0)
    # back-off (round 6): after a timeout the NEXT dense match does not try the multi-workgroup form at all -- on a device another
    # process keeps busy every match would otherwise pay the bounded wait and a second launch; the one after tries again, and the
    # exchange that completes resets the back-off
    pb, cb = g.matchData(sc.query_init[1], pts)
    assert g.last_launch_config()["waves_per_scan"] == 16 and g.debug_coop_fallbacks() == 1
    assert np.array_equal(bits(pb), bits(p1)) and np.array_equal(bits(cb), bits(c1))
    p2, c2 = g.matchData(sc.query_init[1], pts)
    assert g.last_launch_config()["waves_per_scan"] < 0 and g.debug_coop_fallbacks() == 1
    assert np.array_equal(bits(p2), bits(p0)) and np.array_equal(bits(c2), bits(c0))
    g.close()


def test_teardown_leaves_no_pending_runtime_error(capi, pyramid_scene):
    """HIP keeps the last failing call per thread until somebody asks for it: a failure swallowed inside hsm_destroy used to
    surface as `hsm_create failed: hipGetLastError(): invalid argument` in the NEXT context of the thread (seen twice in ~11
    runs of a 2000-example hypothesis soak, round 5).  hsm_destroy now names its first failing call in hsm_last_error() and
    clears the runtime's state; here: contexts whose match + queued update + match loop used every teardown branch (the
    overlapped upload's stream, event and both retained buffers; the pinned blocks) leave nothing behind, and an empty first
    scan of a fresh context does not ask the runtime for the device address of a block that was never allocated."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipPeekAtLastError.restype = ctypes.c_int
    lib = capi.load_library()
    sc = pyramid_scene
    rng = np.random.default_rng(3)
    for k in range(12):
        g = make_gpu(capi, sc, build=False, parity=(capi.PARITY_EXACT, capi.PARITY_FAST, capi.PARITY_AUTO)[k % 3])
        p, c = g.matchData(sc.query_init[0], np.zeros((0, 2), np.float32))  # empty first scan of a fresh context
        assert np.array_equal(bits(p), bits(sc.query_init[0]))
        n = (300, 5000, 17000)[(k // 3) % 3]
        pts = rng.uniform(-0.3 * sc.map_size, 0.3 * sc.map_size, (n, 2)).astype(np.float32)
        pose = sc.build_poses[0]
        for t in range(3):
            pose, _ = g.matchData(pose, pts)
            g.updateByScan(pts, pose)  # queued, not waited for: the next matchData uploads on the copy stream
        assert hip.hipPeekAtLastError() == 0
        g.close()
        assert hip.hipPeekAtLastError() == 0
        assert "hsm_destroy" not in lib.hsm_last_error().decode(), lib.hsm_last_error().decode()


def test_update_serial_wrap_is_bit_exact(capi, oracle_mod, pyramid_scene, kind):
    """the key planes carry a 12-bit per-scan generation (the other 20 bits are the beam index); after 4095 updates
    (100 s at 40 Hz) it wraps and the planes are cleared once.  Updates straddling the wrap -- with stale keys of
    generation 4093..4095 left in the planes -- must still be bit-exact."""
    sc = pyramid_scene
    g = make_gpu(capi, sc, build=False)
    o = make_oracle(oracle_mod, kind, sc, build=False)
    lib = capi.load_library()
    for t in range(24):
        if t == 6:
            for lvl in range(sc.levels):
                capi._check(lib.hsm_debug_set_update_serial(g._h, lvl, 4093 - lvl), "set serial")  # wraps at t = 8..10
        o.match(sc.build_poses[t], sc.build_scans[t])
        g.matchData(sc.build_poses[t], sc.build_scans[t])
        o.update_by_scan(sc.build_poses[t], sc.build_scans[t])
        g.updateByScan(sc.build_scans[t], sc.build_poses[t])
        o.on_map_updated()
    for lvl in range(sc.levels):
        a, b = g.download_level(lvl), o.download_level(lvl)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), lvl
        assert np.array_equal(g.occupancy_grid(lvl), o.occupancy_grid(lvl))
    pg, _ = g.matchData(sc.query_init[0], sc.query_scans[0])
    po, _ = o.match(sc.query_init[0], sc.query_scans[0])
    assert_pose_close(pg, po, "after the wrap")


def test_update_with_more_than_65535_beams_is_bit_exact(capi, oracle_mod, kind):
    """the reference has no limit on the scan length; the update keys carry 20 bits of beam index (1 048 575 beams).
    A 150 000-beam scan (a 3D sensor projected into the plane), three poses, two levels: maps bit-identical, and the
    matcher takes the same scan (fast mode: tolerance; exact mode: bits)"""
    from hector_slam_amd import synth
    res, size, levels = 0.05, 1024, 2
    world = synth.World.make(40.0, 30.0, seed=7)
    s = float(np.float32(1.0) / np.float32(res))
    rng = np.random.default_rng(5)
    poses = synth.loop_trajectory(world, 3).astype(np.float32)
    ang = np.linspace(-np.pi, np.pi, 150_000, endpoint=False)
    scans = []
    for p in poses:
        r = world.raycast(p, ang) + rng.normal(0, 0.01, ang.size)
        keep = (r > 0.3) & (r < 29.9)
        scans.append(np.stack([np.cos(ang[keep]) * r[keep] * s, np.sin(ang[keep]) * r[keep] * s], 1).astype(np.float32))
    assert min(sc.shape[0] for sc in scans) > 100_000
    g = capi.MapRepMultiMap(res, size, size, levels)
    o = oracle_mod.Oracle(kind, res, size, size, levels)
    for m in (g.setUpdateFactorFree, o.set_update_factor_free):
        m(0.4)
    for m in (g.setUpdateFactorOccupied, o.set_update_factor_occupied):
        m(0.9)
    for p, sc in zip(poses, scans):
        o.match(p, sc)
        g.matchData(p, sc)
        o.update_by_scan(p, sc)
        g.updateByScan(sc, p)
        o.on_map_updated()
    for lvl in range(levels):
        a, b = g.download_level(lvl), o.download_level(lvl)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), lvl
    init = poses[1] + np.array([0.05, -0.04, 0.01], np.float32)
    po, co = o.match(init, scans[1])
    pg, _ = g.matchData(init, scans[1])
    assert_pose_close(pg, po, "150k-beam scan")
    g.set_parity(capi.PARITY_EXACT)
    px, cx = g.matchData(init, scans[1])
    assert np.array_equal(bits(px), bits(po)) and np.array_equal(bits(cx), bits(co))


def test_group_gathers_with_rccl(capi, pyramid_scene, monkeypatch):
    """hsm_group_match_batch_device through RCCL (ncclCommInitAll over the group's devices, ONE grouped ncclAllGather of the
    [n,3] poses and the [n,9] Hessians on the replicas' own streams; ragged shards: grouped ncclSend / ncclRecv to the root):
    one replica per device the box has -- a single-rank communicator on the 1-GPU box --, bit-identical to one context and to
    the peer-copy gather.  A device listed twice cannot have two RCCL ranks: AUTO settles on peer copies and says why,
    RCCL asked for explicitly is refused."""
    import torch
    from hector_slam_amd import synth
    sc = pyramid_scene
    ndev = torch.cuda.device_count()
    devices = list(range(ndev))
    one = make_gpu(capi, sc)
    pts, offs = synth.pack_scans(sc.query_scans)
    want_p, want_c = one.match_batch(sc.query_init, pts, offs)
    B = len(sc.query_scans)
    for ragged in (False, True):
        per = B // ndev
        bounds = [(r * per, (r + 1) * per if r + 1 < ndev else (B if not ragged else B - 1)) for r in range(ndev)]
        if ragged and ndev > 1:
            bounds[0] = (0, per - 1)
            bounds[1] = (per - 1, bounds[1][1])
        nb = bounds[-1][1]
        shards = []
        for r, (b, e) in enumerate(bounds):
            dev = torch.device("cuda", devices[r])
            sp, so = synth.pack_scans(sc.query_scans[b:e])
            shards.append([torch.from_numpy(np.ascontiguousarray(sc.query_init[b:e])).to(dev), torch.from_numpy(sp).to(dev), torch.from_numpy(so).to(dev)])
        got = {}
        for mode in (capi.GATHER_RCCL, capi.GATHER_PEER, capi.GATHER_DIRECT):
            grp = capi.MapRepGroup(sc.resolution, sc.map_size, sc.map_size, sc.levels, devices)
            grp.set_update_factors(0.4, 0.9)
            for r in range(ndev):
                grp.member(r).build_map(sc.build_poses, sc.build_scans)
            grp.set_gather(mode)
            assert grp.gather_mode()[0] == {capi.GATHER_RCCL: "rccl", capi.GATHER_PEER: "peer", capi.GATHER_DIRECT: "direct"}[mode]
            root = ndev - 1
            rdev = torch.device("cuda", devices[root])
            for rep in range(3):
                d_all = torch.zeros((nb, 3), dtype=torch.float32, device=rdev)
                d_cov = torch.zeros((nb, 9), dtype=torch.float32, device=rdev)
                torch.cuda.synchronize()
                grp.match_batch_device([e - b for b, e in bounds], [s_[0].data_ptr() for s_ in shards], [s_[1].data_ptr() for s_ in shards],
                                       [s_[2].data_ptr() for s_ in shards], 0, root, d_all.data_ptr(), d_cov.data_ptr())
                grp.synchronize()
                assert np.array_equal(bits(d_all.cpu().numpy()), bits(want_p[:nb])), (mode, ragged, rep)
                assert np.array_equal(bits(d_cov.cpu().numpy()), bits(want_c[:nb])), (mode, ragged, rep)
            if (mode == capi.GATHER_DIRECT or (mode == capi.GATHER_RCCL and not ragged)) and ndev > 1:  # an all-gather: every other replica holds all poses too
                ptr = grp.gathered(0)
                assert ptr != 0
            if mode == capi.GATHER_RCCL:
                # the send / receive form (what unequal shards take) for EVERY shard, the root's own included -- a send to self:
                # the one way a single-device box runs grouped ncclSend / ncclRecv on hardware (round-4 verdict, item 6b)
                grp.debug_force_p2p(True)
                d_all = torch.zeros((nb, 3), dtype=torch.float32, device=rdev)
                d_cov = torch.zeros((nb, 9), dtype=torch.float32, device=rdev)
                torch.cuda.synchronize()
                grp.match_batch_device([e - b for b, e in bounds], [s_[0].data_ptr() for s_ in shards], [s_[1].data_ptr() for s_ in shards],
                                       [s_[2].data_ptr() for s_ in shards], 0, root, d_all.data_ptr(), d_cov.data_ptr())
                grp.synchronize()
                assert np.array_equal(bits(d_all.cpu().numpy()), bits(want_p[:nb])), ("send/recv form", ragged)
                assert np.array_equal(bits(d_cov.cpu().numpy()), bits(want_c[:nb])), ("send/recv form", ragged)
                grp.debug_force_p2p(False)
            got[mode] = d_all.cpu().numpy()
            grp.close()
        assert np.array_equal(bits(got[capi.GATHER_RCCL]), bits(got[capi.GATHER_PEER]))
        assert np.array_equal(bits(got[capi.GATHER_DIRECT]), bits(got[capi.GATHER_PEER]))
    # one device listed twice: AUTO takes the device-side exchange (round 6; it needs no communicator); no second RCCL rank
    # on one device -- RCCL asked for explicitly is refused, and says why
    grp = capi.MapRepGroup(sc.resolution, sc.map_size, sc.map_size, sc.levels, [0, 0])
    mode, note = grp.gather_mode()
    assert mode == "direct", (mode, note)
    with pytest.raises(capi.HsmError, match="more than once"):
        grp.set_gather(capi.GATHER_RCCL)
    grp.close()
    monkeypatch.setenv("HSM_GROUP_GATHER", "Rccl")  # a typo must not silently select something
    with pytest.raises(capi.HsmError):
        capi.MapRepGroup(sc.resolution, sc.map_size, sc.map_size, sc.levels, [0])
    one.close()


def test_single_process_device_group(capi, oracle_mod, pyramid_scene, kind):
    """hsm_group_*: R replicas in one process -- on R DISTINCT devices where the box has them (peer copies, per-device
    contexts and streams), else all on device 0 (the sharding, the persistent worker threads and the replica consistency
    logic are the same).  Batched matching over the group == one context, bit for bit and in order; the SLAM cycle
    keeps every replica's map identical to a single context's and to the oracle's"""
    import torch
    from hector_slam_amd import synth
    sc = pyramid_scene
    ndev = torch.cuda.device_count()
    devices = [0, 1 % ndev, 2 % ndev]
    grp = capi.MapRepGroup(sc.resolution, sc.map_size, sc.map_size, sc.levels, devices)
    assert grp.size() == 3
    grp.set_update_factors(0.4, 0.9)
    one = make_gpu(capi, sc, build=False)
    o = make_oracle(oracle_mod, kind, sc, build=False)
    o.proc_set_thresholds(0.0, 0.0)
    origo = np.array([0.2, -0.1], np.float32) * np.float32(sc.scale_to_map)
    hint = sc.build_poses[0].copy()
    cov = np.zeros(9, np.float32)
    for t in range(20):
        pg, cov = grp.process_scan(hint, sc.build_scans[t], origo, True, cov)
        o.proc_update(sc.build_scans[t], hint, origo=origo)
        p1, _ = one.matchData(hint, sc.build_scans[t], None, origo)
        one.updateByScan(sc.build_scans[t], p1, origo)
        po, _ = o.proc_last_pose()
        assert np.array_equal(bits(pg), bits(p1))
        assert_pose_close(pg, po, f"group t={t}")
        hint = pg + (sc.build_poses[t + 1] - sc.build_poses[t])
    for lvl in range(sc.levels):
        ref = one.download_level(lvl)
        for r in range(3):
            got = grp.member(r).download_level(lvl)
            assert np.array_equal(bits(got[0]), bits(ref[0])) and np.array_equal(got[1], ref[1]), (lvl, r)
    # batched matching: 16 scans over 3 replicas (uneven shards 5/5/6) == one context
    pts, offs = synth.pack_scans(sc.query_scans)
    pa, ca = grp.match_batch(sc.query_init, pts, offs)
    pb, cb = one.match_batch(sc.query_init, pts, offs)
    assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(ca), bits(cb))
    hyp = np.repeat(sc.query_init[3:4], 10, 0) + np.linspace(-0.05, 0.05, 10, dtype=np.float32)[:, None]
    ha, _ = grp.match_batch(hyp, sc.query_scans[3], None)
    hb, _ = one.match_batch(hyp, sc.query_scans[3], None)
    assert np.array_equal(bits(ha), bits(hb))
    # fewer scans than replicas, and none
    p2, _ = grp.match_batch(sc.query_init[:2], *synth.pack_scans(sc.query_scans[:2]))
    assert np.array_equal(bits(p2), bits(pb[:2]))
    assert grp.match_batch(np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), np.zeros(1, np.int32))[0].shape == (0, 3)
    # device-resident shards (uneven: 5 / 0 / 11 scans), gathered on replica 2's device (AUTO: the device-side exchange;
    # then peer copies); repeated so that the persistent workers, the mailboxes and the per-replica result blocks are reused
    bounds = [(0, 5), (5, 5), (5, 16)]
    shards = []
    for r, (b, e) in enumerate(bounds):
        dev = torch.device("cuda", devices[r])
        sh_pts, sh_offs = synth.pack_scans(sc.query_scans[b:e]) if e > b else (np.zeros((0, 2), np.float32), np.zeros(1, np.int32))
        shards.append([torch.from_numpy(np.ascontiguousarray(sc.query_init[b:e])).to(dev), torch.from_numpy(sh_pts).to(dev),
                       torch.from_numpy(sh_offs).to(dev)])
    root = 2
    rdev = torch.device("cuda", devices[root])
    assert grp.gather_mode()[0] == "direct"
    for rep in range(6):
        if rep == 3:
            grp.set_gather(capi.GATHER_PEER)
        d_all = torch.zeros((16, 3), dtype=torch.float32, device=rdev)
        d_cov = torch.zeros((16, 9), dtype=torch.float32, device=rdev)
        torch.cuda.synchronize()
        grp.match_batch_device([e - b for b, e in bounds], [t[0].data_ptr() if t[0].numel() else 0 for t in shards],
                               [t[1].data_ptr() if t[1].numel() else 0 for t in shards], [t[2].data_ptr() for t in shards],
                               0, root, d_all.data_ptr(), d_cov.data_ptr())
        grp.synchronize()
        assert np.array_equal(bits(d_all.cpu().numpy()), bits(pb)), rep
        assert np.array_equal(bits(d_cov.cpu().numpy()), bits(cb)), rep
    grp.close()


@pytest.mark.parametrize("levels", [1, 2, 5])
def test_processor_lifecycle_levels_reset_and_factor_changes(capi, oracle_mod, levels, kind):
    """level counts other than the default 3 (MapRepSingleMap-like 1, launch-file default 2, deep 5), a reset in the
    middle of a run, update factors changed on the fly (they only affect later updates), a first scan mapped without
    matching: HSM_PARITY_EXACT equals the reference on every step (no convergence predicate); the fast mode is measured
    against it with the bound stated at the end; with identical poses the maps are bit-identical"""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=720, map_size=1024, levels=levels, resolution=0.05, n_build=40, n_query=2,
                          room=(30.0, 22.0), seed=31 + levels)
    o = oracle_mod.Oracle(kind, sc.resolution, sc.map_size, sc.map_size, levels)
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, levels)
    assert g.getMapLevels() == levels == o.levels() and g.gn_iterations_per_match() == 6 + 4 * (levels - 1)

    def both(fn_g, fn_o, *a):
        fn_g(*a)
        fn_o(*a)

    both(g.setUpdateFactorFree, o.set_update_factor_free, 0.4)
    both(g.setUpdateFactorOccupied, o.set_update_factor_occupied, 0.9)
    pose = sc.build_poses[0].copy()
    fast_n = fast_within = 0
    fast_worst = 0.0
    for t in range(36):
        if t == 14:  # syscommand "reset" (HectorMappingRos.cpp:383-392)
            g.reset()
            o.reset()
        if t == 22:
            both(g.setUpdateFactorFree, o.set_update_factor_free, 0.45)
            both(g.setUpdateFactorOccupied, o.set_update_factor_occupied, 0.7)
        hint = pose + (sc.build_poses[t] - sc.build_poses[max(t - 1, 0)])
        if t in (0, 14):  # map_without_matching: the pose is taken as given, coarse levels use stale containers
            po = pg = hint.astype(np.float32)
        else:
            po, co = o.match(hint, sc.build_scans[t])
            g.set_parity(capi.PARITY_EXACT)
            px, cx = g.matchData(hint, sc.build_scans[t])
            g.set_parity(capi.PARITY_FAST)
            assert np.array_equal(bits(px), bits(po)) and np.array_equal(bits(cx), bits(co)), (levels, t)
            pg, cg = g.matchData(hint, sc.build_scans[t])
            d = np.abs(pg.astype(np.float64) - px)
            fast_n += 1
            fast_within += bool(d[0] <= POSE_TOL_M and d[1] <= POSE_TOL_M and ang_diff(pg[2], px[2]) <= POSE_TOL_RAD)
            fast_worst = max(fast_worst, float(d[:2].max()))
        o.update_by_scan(po, sc.build_scans[t])
        o.on_map_updated()
        g.updateByScan(sc.build_scans[t], po)
        pose = po
    print(f"levels {levels}: fast vs exact: {fast_within}/{fast_n} within 1e-4, worst {fast_worst:.2e} m")
    assert fast_worst <= FAST_LIFECYCLE_WORST_M and fast_within >= FAST_LIFECYCLE_WITHIN * fast_n, (fast_within, fast_n, fast_worst)
    for lvl in range(levels):
        a, b = g.download_level(lvl), o.download_level(lvl)
        assert np.array_equal(bits(a[0]), bits(b[0])) and np.array_equal(a[1], b[1]), lvl
        assert g.getUpdateIndex(lvl) == 35  # reset() does not rewind the update counter (GridMapBase::reset)


def test_ragged_batch_with_empty_and_tiny_scans(capi, pyr, pyramid_scene):
    """CSR batch mixing empty scans, 1..70-beam fragments and full scans: every entry equals its single-scan call
    (same team width), empty scans pass their start pose through and leave cov untouched"""
    from hector_slam_amd import synth
    g, o = pyr
    sc = pyramid_scene
    g1 = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, waves_per_scan=1)
    for lvl in range(sc.levels):
        g1.upload_level(lvl, *o.download_level(lvl))
    full = sc.query_scans[0]
    sizes = [0, 1081 if full.shape[0] >= 1081 else full.shape[0], 1, 0, 70, 64, 2, 500, 0]
    scans = [full[np.linspace(0, full.shape[0] - 1, n).astype(int)] if n else np.zeros((0, 2), np.float32) for n in sizes]
    init = np.repeat(sc.query_init[0:1], len(sizes), 0) + np.arange(len(sizes), dtype=np.float32)[:, None] * 1e-3
    pts, offs = synth.pack_scans(scans)
    pose, cov = g1.match_batch(init, pts, offs)
    for i, n in enumerate(sizes):
        if n == 0:
            assert np.array_equal(bits(pose[i]), bits(init[i])) and not cov[i].any()
        else:
            ps, cs = g1.matchData(init[i], scans[i])
            assert np.array_equal(bits(ps), bits(pose[i])) and np.array_equal(bits(cs), bits(cov[i])), (i, n)
            if n >= 500:
                po, _ = o.match(init[i], scans[i])
                assert_pose_close(pose[i], po, f"ragged {i}")


def test_queued_update_equals_blocking_update(capi, pyramid_scene, monkeypatch):
    """updateByScan returns once queued: the caller may scribble over its scan buffer at once, later calls see
    the completed update, and the map is bit-identical to a context that blocks in every update
    (HSM_ASYNC_UPDATE=0) and to one that waits on the end-of-kernel signal (HSM_SPIN_WAIT=0)"""
    sc = pyramid_scene
    monkeypatch.setenv("HSM_ASYNC_UPDATE", "0")
    monkeypatch.setenv("HSM_SPIN_WAIT", "0")
    blocking = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    monkeypatch.delenv("HSM_ASYNC_UPDATE")
    monkeypatch.delenv("HSM_SPIN_WAIT")
    queued = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    for m in (blocking, queued):
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
    rng = np.random.default_rng(77)
    for t in range(40):
        scan = np.ascontiguousarray(sc.build_scans[t]).copy()
        if t % 7 == 3:
            scan = scan[: 1 + (t * 13) % 60]       # tiny scans exercise the staging slots as well
        hint = sc.build_poses[t] + (rng.uniform(-0.03, 0.03, 3) * [1, 1, 0.2]).astype(np.float32)
        pb, cb = blocking.matchData(hint, scan)
        buf = scan.copy()
        pq, cq = queued.matchData(hint, buf)
        assert np.array_equal(bits(pb), bits(pq)) and np.array_equal(bits(cb), bits(cq)), t
        blocking.updateByScan(scan, pb)
        queued.updateByScan(buf, pq)
        buf[:] = np.float32(1e6)                    # the call has copied what it needs
        if t % 5 == 0:
            assert np.array_equal(queued.take_dirty_bbox(0), blocking.take_dirty_bbox(0))
        if t % 11 == 0:
            la, lb = queued.download_level(0), blocking.download_level(0)
            assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1]), t
    queued.synchronize()
    for lvl in range(sc.levels):
        la, lb = queued.download_level(lvl), blocking.download_level(lvl)
        assert (la[0] != 0).sum() > 1000
        assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1]), lvl
    assert np.array_equal(queued.occupancy_grid(0), blocking.occupancy_grid(0))


def test_texel_cache_form_is_bit_identical(capi, pyr, pyramid_scene, monkeypatch):
    """gn_match_cached_kernel (endpoints in LDS, last texel of every beam kept in VGPRs, gathers only in the
    lanes whose cell changed) == gn_match_kernel bit for bit: poses and covariances, full pyramid, ragged scans"""
    from hector_slam_amd import synth
    g, o = pyr
    sc = pyramid_scene
    monkeypatch.setenv("HSM_TEXEL_CACHE", "1")
    cached = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, waves_per_scan=1,
                                 layout=capi.LAYOUT_QUAD)
    monkeypatch.setenv("HSM_TEXEL_CACHE", "0")
    plain = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, waves_per_scan=1,
                                layout=capi.LAYOUT_QUAD)
    for lvl in range(sc.levels):
        lv = o.download_level(lvl)
        cached.upload_level(lvl, *lv)
        plain.upload_level(lvl, *lv)
    rng = np.random.default_rng(31)
    nq = len(sc.query_scans)
    for sizes in ([1081] * 96, list(rng.integers(330, 1081, 64)) + [0, 577, 576, 1081, 0], [450] * 40):
        scans, init = [], []
        for j, n in enumerate(sizes):
            full = sc.query_scans[j % nq]
            n = min(int(n), full.shape[0])
            scans.append(full[np.sort(rng.choice(full.shape[0], n, replace=False))] if n else np.zeros((0, 2), np.float32))
            init.append(sc.query_init[j % nq] + (rng.uniform(-0.05, 0.05, 3) * [1, 1, 0.2]).astype(np.float32))
        init = np.asarray(init, np.float32)
        pts, offs = synth.pack_scans(scans)
        pc, cc = cached.match_batch(init, pts, offs)
        cfg = cached.last_launch_config()
        pp, cp = plain.match_batch(init, pts, offs)
        assert cfg["texel_cache"] and not plain.last_launch_config()["texel_cache"], cfg
        assert np.array_equal(bits(pc), bits(pp)) and np.array_equal(bits(cc), bits(cp)), sizes[:3]
    po = np.stack([o.match(init[j], scans[j])[0] for j in range(8)])
    assert_pose_close(pc[:8], po, "texel-cache form vs oracle")


def test_relaxed_mode_meets_the_pose_tolerance(capi, pyr, pyramid_scene):
    """HSM_PARITY_RELAXED (opt-in; gn_match_cached_kernel<.., RELAXED>: multiply-add pairs contracted to v_fma_f32): the batch
    kernel's poses stay within north_star's 1e-4 m / 1e-4 rad of the reference on every scan of the batch, ragged scans and
    all three levels included; single scans and every other entry run as in the fast mode (bit-identical to it)"""
    from hector_slam_amd import synth
    g, o = pyr
    sc = pyramid_scene
    m = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, waves_per_scan=1, layout=capi.LAYOUT_QUAD)
    for lvl in range(sc.levels):
        m.upload_level(lvl, *o.download_level(lvl))
    rng = np.random.default_rng(33)
    nq = len(sc.query_scans)
    scans, init = [], []
    for j, n in enumerate([1081] * 48 + list(rng.integers(400, 1081, 30)) + [0, 600]):
        full = sc.query_scans[j % nq]
        n = min(int(n), full.shape[0])
        scans.append(full[np.sort(rng.choice(full.shape[0], n, replace=False))] if n else np.zeros((0, 2), np.float32))
        init.append(sc.query_init[j % nq] + (rng.uniform(-0.05, 0.05, 3) * [1, 1, 0.2]).astype(np.float32))
    init = np.asarray(init, np.float32)
    pts, offs = synth.pack_scans(scans)
    m.set_parity(capi.PARITY_FAST)
    pf, _ = m.match_batch(init, pts, offs)
    ps_fast, _ = m.matchData(init[0], scans[0])
    m.set_parity(capi.PARITY_RELAXED)
    assert m.parity() == capi.PARITY_RELAXED and m.last_launch_config()["parity"] == "relaxed"
    pr, cr = m.match_batch(init, pts, offs)
    assert m.last_launch_config()["texel_cache"]
    assert np.isfinite(pr).all() and np.isfinite(cr).all()
    assert not np.array_equal(bits(pr), bits(pf))  # it IS another arithmetic ...
    po = np.stack([o.match(init[j], scans[j])[0] for j in range(len(scans))])
    assert_pose_close(pr, po, "relaxed batch vs oracle")  # ... within the tolerance on every scan
    ps_rel, _ = m.matchData(init[0], scans[0])
    assert np.array_equal(bits(ps_rel), bits(ps_fast))  # single scans: the fast kernels
    m.close()


def test_dense_update_forms_agree_bit_for_bit(capi, oracle_mod, kind, monkeypatch):
    """scans of >= 4096 beams: the byte-map form of updateByScan (update_mark_free_dense_kernel / update_apply_dense_kernel)
    == the keyed form (HSM_DENSE_BITS=0) == the oracle, both layouts, all levels -- incl. a map whose rows are NOT a multiple
    of 64 cells, where the library falls back to the keyed form by itself"""
    from hector_slam_amd import synth
    for size, levels in ((512, 3), (500, 2)):
        sc = synth.make_scene(n_beams=6000, map_size=size, levels=levels, resolution=0.05, n_build=6, n_query=1,
                              room=(20.0, 15.0), seed=17)
        o = make_oracle(oracle_mod, kind, sc, build=False)
        gs = []
        for env, lay in (("1", capi.LAYOUT_QUAD), ("1", capi.LAYOUT_PLANE), ("0", capi.LAYOUT_QUAD)):
            monkeypatch.setenv("HSM_DENSE_BITS", env)
            gs.append(make_gpu(capi, sc, build=False, layout=lay))
        for t in range(6):
            o.match(sc.build_poses[t], sc.build_scans[t])
            o.update_by_scan(sc.build_poses[t], sc.build_scans[t])
            for g in gs:
                g.matchData(sc.build_poses[t], sc.build_scans[t])
                g.updateByScan(sc.build_scans[t], sc.build_poses[t])
        for lvl in range(levels):
            lo_o, ui_o = o.download_level(lvl)
            assert (ui_o >= 0).sum() > 1000
            _, prob = oracle_mod.libm_expf(lo_o.reshape(-1), o.kind)
            for g in gs:
                lo_g, ui_g = g.download_level(lvl)
                assert np.array_equal(bits(lo_g), bits(lo_o)) and np.array_equal(ui_g, ui_o), (size, lvl)
                assert np.array_equal(bits(g.download_prob(lvl)).reshape(-1), bits(prob)), (size, lvl)
        # the quad-layout contexts hold identical texels (the matcher's view of the map): same pose from the same start
        p0, _ = gs[0].matchData(sc.query_init[0], sc.query_scans[0])
        p2, _ = gs[2].matchData(sc.query_init[0], sc.query_scans[0])
        assert np.array_equal(bits(p0), bits(p2)), size
        for g in gs:
            g.close()


def test_dense_and_sparse_scans_alternate_on_one_map(capi, oracle_mod, kind):
    """the dense path (>= 4096 beams: mark bytes with the end-cell flag, no bitmap) and the keyed path (fewer beams: key planes
    + end-cell bitmap) share the key planes and the serial tags of ONE map: scans of 6000, 700, 4100, 90 ... beams in turn,
    matchData + updateByScan each, must leave every level bit-identical to the reference -- nothing one path leaves behind
    (a mark byte, a bitmap word, a key of an older serial) may leak into the other"""
    from hector_slam_amd import synth
    sc = synth.make_scene(n_beams=6000, map_size=512, levels=3, resolution=0.05, n_build=10, n_query=1, room=(20.0, 15.0), seed=23)
    o = make_oracle(oracle_mod, kind, sc, build=False)
    g = make_gpu(capi, sc, build=False, layout=capi.LAYOUT_QUAD)
    keep = (6000, 700, 4100, 90, 6000, 1081, 5000, 4095, 4096, 300)
    for t, n in enumerate(keep):
        scan = sc.build_scans[t]
        scan = scan[np.linspace(0, scan.shape[0] - 1, min(n, scan.shape[0])).astype(np.int64)]
        o.match(sc.build_poses[t], scan)
        o.update_by_scan(sc.build_poses[t], scan)
        g.matchData(sc.build_poses[t], scan)
        g.updateByScan(scan, sc.build_poses[t])
        o.on_map_updated()
        for lvl in range(3):  # after EVERY update: a leak would show at the step it happens
            lo_o, ui_o = o.download_level(lvl)
            lo_g, ui_g = g.download_level(lvl)
            assert np.array_equal(bits(lo_g), bits(lo_o)) and np.array_equal(ui_g, ui_o), (t, n, lvl)
    # a dense scan matched, then ANOTHER dense scan of the same length handed to updateByScan: level 0 takes the new
    # endpoints, the coarse levels the container the match retained (MapRepMultiMap.h:143) -- the update must not take the
    # matcher's device copy for the caller's scan just because the lengths agree
    a, b = sc.build_scans[0][:5000], sc.build_scans[4][:5000].copy()
    assert a.shape == b.shape and not np.array_equal(a, b)
    for (mscan, uscan) in ((a, b), (b, b), (a, a)):
        o.match(sc.build_poses[2], mscan)
        o.update_by_scan(sc.build_poses[2], uscan)
        g.matchData(sc.build_poses[2], mscan)
        g.updateByScan(uscan, sc.build_poses[2])
        o.on_map_updated()
        for lvl in range(3):
            lo_o, ui_o = o.download_level(lvl)
            lo_g, ui_g = g.download_level(lvl)
            assert np.array_equal(bits(lo_g), bits(lo_o)) and np.array_equal(ui_g, ui_o), ("matched != updated scan", lvl)
    g.set_parity(capi.PARITY_EXACT)  # the matcher's view of that map (texels): the reference's pose, bit for bit
    p_o, c_o = o.match(sc.query_init[0], sc.query_scans[0])
    p_g, c_g = g.matchData(sc.query_init[0], sc.query_scans[0])
    assert np.array_equal(bits(p_g), bits(p_o)) and np.array_equal(bits(c_g), bits(c_o))
    g.close()


def test_workgroup_to_xcd_mapping_is_a_permutation(capi, pyr, pyramid_scene, monkeypatch):
    """xcd_block(): whichever mapping a launch uses -- chunks of 16 workgroups dealt to the XCDs (default), odd chunk
    sizes, one contiguous eighth per XCD -- every scan is matched exactly once: a 1003-scan batch (251 workgroups: one
    full round of 8 x 16 chunks + a ragged remainder) gives the same bits under all of them, and no row is left unwritten"""
    from hector_slam_amd import synth
    g, o = pyr
    sc = pyramid_scene
    rng = np.random.default_rng(77)
    nq = len(sc.query_scans)
    B = 1003
    scans = [sc.query_scans[j % nq] for j in range(B)]
    init = np.stack([sc.query_init[j % nq] + (rng.uniform(-0.05, 0.05, 3) * [1, 1, 0.2]).astype(np.float32) for j in range(B)])
    pts, offs = synth.pack_scans(scans)
    results = []
    for chunk in ("16", "0", "5", "64"):
        monkeypatch.setenv("HSM_XCD_CHUNK", chunk)
        m = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, waves_per_scan=1, layout=capi.LAYOUT_QUAD)
        for lvl in range(sc.levels):
            m.upload_level(lvl, *o.download_level(lvl))
        p, c = m.match_batch(init, pts, offs)
        assert m.last_launch_config()["texel_cache"]
        results.append((p, c))
        m.close()
    for p, c in results[1:]:
        assert np.array_equal(bits(p), bits(results[0][0])) and np.array_equal(bits(c), bits(results[0][1]))
    assert not np.any(np.all(results[0][0] == 0.0, axis=1))  # every scan was written
    for j in (0, 511, 512, 1002):
        assert_pose_close(results[0][0][j:j + 1], np.stack([o.match(init[j], scans[j])[0]]), f"scan {j} vs oracle")


def test_clock_probe_reports_a_plausible_shader_clock(capi, pyr, pyramid_scene):
    """hsm_set_clock_probe: the wave of scan 0 stamps {s_memtime, 100 MHz wall clock} at its first GN step and at its end;
    the probe changes no result and the ratio is a clock an MI355X can run at"""
    import torch
    from hector_slam_amd import synth
    g, o = pyr
    sc = pyramid_scene
    nq = len(sc.query_scans)
    B = 256
    scans = [sc.query_scans[j % nq] for j in range(B)]
    init = np.stack([sc.query_init[j % nq] for j in range(B)])
    pts, offs = synth.pack_scans(scans)
    m = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, waves_per_scan=1, layout=capi.LAYOUT_QUAD)
    for lvl in range(sc.levels):
        m.upload_level(lvl, *o.download_level(lvl))
    p0, c0 = m.match_batch(init, pts, offs)
    assert m.last_launch_config()["texel_cache"]
    stamps = torch.zeros(4, dtype=torch.int64, device="cuda")
    m.set_clock_probe(stamps.data_ptr())
    p1, c1 = m.match_batch(init, pts, offs)
    torch.cuda.synchronize()
    m.set_clock_probe(0)
    st = stamps.cpu().numpy().astype(np.uint64)
    assert np.array_equal(bits(p0), bits(p1)) and np.array_equal(bits(c0), bits(c1))
    assert st[3] > st[1] and st[2] > st[0], st
    ghz = float(st[2] - st[0]) / float(st[3] - st[1]) * 0.1
    assert 0.8 < ghz < 3.0, ghz
    m.close()


def test_queued_updates_are_ordered_against_caller_streams(capi, pyramid_scene, monkeypatch):
    """hsm_match_batch_device runs on a CALLER-owned stream while map updates are queued on the context's own:
    a batch match must see every update queued before it, and an update must not rewrite the map under a batch
    match that is still running.  Interleaved tightly and compared with a context that blocks in every update."""
    import torch
    sc = pyramid_scene
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("HSM_ASYNC_UPDATE", "0")
    ref = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    monkeypatch.delenv("HSM_ASYNC_UPDATE")
    dut = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    for m in (ref, dut):
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
    B = 256
    full = sc.query_scans[0]
    init = np.repeat(sc.query_init[0:1], B, 0) + np.random.default_rng(3).uniform(-0.03, 0.03, (B, 3)).astype(np.float32) * np.float32([1, 1, 0.2])
    d_init = torch.from_numpy(init).to(dev)
    d_pts = torch.from_numpy(np.ascontiguousarray(full)).to(dev)
    side = torch.cuda.Stream(device=dev)
    from hector_slam_amd import synth
    rng = np.random.default_rng(8)
    sfac = float(np.float32(1.0) / np.float32(sc.resolution))
    dense = [synth.make_scan(sc.world, sc.build_poses[t], 16384, sfac, rng) for t in range(30)]  # long-running updates
    outs = {"ref": [], "dut": []}
    for name, m in (("ref", ref), ("dut", dut)):
        d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        for t in range(30):
            m.matchData(sc.build_poses[t], dense[t])                   # retains the coarse containers
            m.updateByScan(dense[t], sc.build_poses[t])                # queued (dut) / blocking (ref); ~0.1 ms of GPU work
            m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), 0, full.shape[0], d_pose.data_ptr(), 0,
                                 side.cuda_stream)
            if name == "ref":
                side.synchronize()
            if t % 3 == 2:                                             # leave some matches in flight under the next update
                side.synchronize()
                outs[name].append(d_pose.cpu().numpy().copy())
        side.synchronize()
        m.synchronize()
        outs[name].append(d_pose.cpu().numpy().copy())
    for a, b in zip(outs["ref"], outs["dut"]):
        assert np.array_equal(bits(a), bits(b))
    for lvl in range(sc.levels):
        la, lb = ref.download_level(lvl), dut.download_level(lvl)
        assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1])


def test_scans_longer_than_the_length_hint(capi, pyr, pyramid_scene):
    """hsm_match_batch_device sizes its kernel form from a HINT (`shared_n` with CSR offsets: typical beams per scan,
    0 = unknown).  Scans longer than the hint -- beyond the 64 * BPL beams the texel-cache / register-resident forms
    keep on chip -- must be matched completely: same bits as the form that streams every beam from memory."""
    import torch
    from hector_slam_amd import synth
    _, o = pyr
    sc = pyramid_scene
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, waves_per_scan=1)  # the throughput forms
    for lvl in range(sc.levels):
        g.upload_level(lvl, *o.download_level(lvl))
    long_scan = np.concatenate([sc.query_scans[3], sc.query_scans[3][::2] + np.float32(0.01)])  # 1622 beams
    scans = [sc.query_scans[0], long_scan, sc.query_scans[1][:300], long_scan[:1100], sc.query_scans[2]]
    init = np.stack([sc.query_init[0], sc.query_init[3], sc.query_init[1], sc.query_init[3], sc.query_init[2]])
    pts, offs = synth.pack_scans(scans)
    ref_pose, ref_cov = g.match_batch(init, pts, offs)  # host entry: sized from the true maximum (memory loop)
    assert g.last_launch_config()["beams_per_lane"] == 0
    dev = torch.device("cuda", 0)
    d_init, d_pts, d_offs = (torch.from_numpy(a).to(dev) for a in (init, pts, offs))
    stream = torch.cuda.current_stream()
    forms = set()
    for hint in (0, 1081, 500, 300, 100):  # texel-cache BPL 17 / 9, register-resident BPL 5 / 2
        d_pose = torch.zeros((len(scans), 3), dtype=torch.float32, device=dev)
        d_cov = torch.zeros((len(scans), 9), dtype=torch.float32, device=dev)
        g.match_batch_device(len(scans), d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), hint,
                             d_pose.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize()
        cfg = g.last_launch_config()
        forms.add((cfg["texel_cache"], cfg["beams_per_lane"]))
        assert np.array_equal(bits(d_pose.cpu().numpy()), bits(ref_pose)), (hint, cfg)
        assert np.array_equal(bits(d_cov.cpu().numpy()), bits(ref_cov)), (hint, cfg)
    assert len(forms) >= 4, forms
    po, _ = o.match(init[1], long_scan)
    assert_pose_close(ref_pose[1], po, "1622-beam scan")


def test_two_caller_streams_are_each_ordered_against_updates(capi, pyramid_scene, monkeypatch):
    """the same with matches in flight on TWO caller-owned streams inside one update epoch: each stream is ordered
    behind the queued updates on its own, and the next update waits for both (the ordering state is per stream)"""
    import torch
    from hector_slam_amd import synth
    sc = pyramid_scene
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("HSM_ASYNC_UPDATE", "0")
    ref = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    monkeypatch.delenv("HSM_ASYNC_UPDATE")
    dut = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    for m in (ref, dut):
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
    B = 256
    full = sc.query_scans[0]
    init = np.repeat(sc.query_init[0:1], B, 0) + np.random.default_rng(3).uniform(-0.03, 0.03, (B, 3)).astype(np.float32) * np.float32([1, 1, 0.2])
    d_init = torch.from_numpy(init).to(dev)
    d_pts = torch.from_numpy(np.ascontiguousarray(full)).to(dev)
    sides = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    rng = np.random.default_rng(8)
    sfac = float(np.float32(1.0) / np.float32(sc.resolution))
    dense = [synth.make_scan(sc.world, sc.build_poses[t], 16384, sfac, rng) for t in range(24)]
    outs = {"ref": [], "dut": []}
    for name, m in (("ref", ref), ("dut", dut)):
        d_pose = [torch.zeros((B, 3), dtype=torch.float32, device=dev) for _ in sides]
        for t in range(24):
            m.matchData(sc.build_poses[t], dense[t])
            m.updateByScan(dense[t], sc.build_poses[t])
            for k, side in enumerate(sides):  # second stream: first call of this epoch on that stream, too
                m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), 0, full.shape[0], d_pose[k].data_ptr(), 0,
                                     side.cuda_stream)
                if name == "ref":
                    side.synchronize()
            if t % 3 == 2:
                for side in sides:
                    side.synchronize()
                outs[name].append(np.stack([p.cpu().numpy() for p in d_pose]))
        for side in sides:
            side.synchronize()
        m.synchronize()
        outs[name].append(np.stack([p.cpu().numpy() for p in d_pose]))
    for a, b in zip(outs["ref"], outs["dut"]):
        assert np.array_equal(bits(a), bits(b))
    for lvl in range(sc.levels):
        la, lb = ref.download_level(lvl), dut.download_level(lvl)
        assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1])

"""HSM_PARITY_EXACT: the GPU matcher is BIT-IDENTICAL to the reference CPU matcher -- H, dTr, every Gauss-Newton
step, the pose and the covariance -- on every scan, settled or not.

What makes that possible: the per-beam products were bit-exact all along; sinf / cosf / expf
are glibc's algorithms operation for operation (csrc/libm_exact.h); and in this mode the nine sums of
getCompleteHessianDerivs (OccGridMapUtil.h:76-98) run in the reference's order, beam 0 .. n-1, as nine sequential
fp32 chains (gn_match.h exact_round).  So no tolerance appears in this file: every comparison is on uint32 views.

Runs against both CPU checkers where oracle/_ref/libhector_ref.so is present: "ho" (restatement) and "hr" (the
unmodified reference headers).
"""
import numpy as np
import pytest

from conftest import bits, make_oracle, oracle_kinds

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available(), "gpu-marked tests need a HIP device"
    from hector_slam_amd import capi as m
    m.load_library()
    return m


@pytest.fixture(scope="module", params=oracle_kinds())
def kind(request):
    return request.param


def same(a, b):
    return np.array_equal(bits(a), bits(b))


def exact_gpu(capi, sc, o=None, **kw):
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, parity=capi.PARITY_EXACT, **kw)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    assert g.parity() == capi.PARITY_EXACT
    if o is None:
        g.build_map(sc.build_poses, sc.build_scans)
    else:
        for lvl in range(sc.levels):
            g.upload_level(lvl, *o.download_level(lvl))
    return g


@pytest.fixture(scope="module")
def pyr(capi, oracle_mod, pyramid_scene, kind):
    o = make_oracle(oracle_mod, kind, pyramid_scene)
    return exact_gpu(capi, pyramid_scene), o  # the GPU builds its own map with its own update kernels


def test_hessian_and_gradient_bit_identical(pyr, pyramid_scene):
    """one getCompleteHessianDerivs evaluation: all 9 + 3 numbers equal the reference's, every level"""
    g, o = pyr
    sc = pyramid_scene
    for q in range(len(sc.query_scans)):
        for lvl in range(sc.levels):
            pts = sc.query_scans[q] * np.float32(1.0 / 2 ** lvl)
            pm = o.map_coords_pose(lvl, sc.query_init[q])
            Hg, dg = g.hessian_derivs(lvl, pm, pts)
            Ho, do = o.hessian_derivs(lvl, pm, pts)
            assert same(Hg, Ho) and same(dg, do), (q, lvl)


@pytest.mark.parametrize("layout", ["quad", "plane"])
@pytest.mark.parametrize("wps", [0, 1, 2, 4, 8, 16])
def test_single_scan_match_bit_identical_for_every_team_width(capi, oracle_mod, pyramid_scene, kind, wps, layout):
    """hsm_match through 1..16 wavefronts per scan: pose AND covariance bit-identical (the team width changes how the
    beams are dealt to lanes, not the order of the nine sums)"""
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = exact_gpu(capi, sc, o, waves_per_scan=wps, layout=capi.LAYOUT_QUAD if layout == "quad" else capi.LAYOUT_PLANE)
    for q in range(len(sc.query_scans)):
        pg, cg = g.matchData(sc.query_init[q], sc.query_scans[q])
        po, co = o.match(sc.query_init[q], sc.query_scans[q])
        assert same(pg, po) and same(cg, co), (q, pg, po)
    full = sc.query_scans[2]
    # beam counts around every boundary of the chain: 16 beams per chain iteration, rounds of 64 x team width, groups of five rounds
    # (from 15 beams up: with one or two beams H is singular, the reference's estimate turns NaN and it indexes the map with it)
    for n in (15, 16, 17, 40, 63, 64, 65, 127, 129, 255, 256, 257, 319, 320, 321, 639, 641, 1000, 1023, 1025):
        pts = full[np.linspace(0, full.shape[0] - 1, n).astype(int)]
        pg, cg = g.matchData(sc.query_init[2], pts)
        po, co = o.match(sc.query_init[2], pts)
        assert same(pg, po) and same(cg, co), n
    cov_in = np.arange(9, dtype=np.float32)
    p, c = g.matchData(sc.query_init[0], np.zeros((0, 2), np.float32), cov_in)  # empty scan: pass-through
    assert same(p, sc.query_init[0]) and same(c, cov_in)


def test_config1_single_level_five_iterations(capi, oracle_mod, small_scene, kind):
    """BASELINE configs[0]: 181 beams, 256^2 single-resolution map, 5 GN iterations"""
    sc = small_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = exact_gpu(capi, sc)
    for q in range(len(sc.query_scans)):
        for it in (0, 1, 5):
            pg, cg = g.match_level(0, sc.query_init[q], sc.query_scans[q], it)
            po, co = o.match_level(0, sc.query_init[q], sc.query_scans[q], it)
            assert same(pg, po) and same(cg, co), (q, it)


def test_far_starts_and_out_of_map_beams(pyr, pyramid_scene):
    """starts far outside the basin (angle clamp active, Gauss-Newton not settling) and a start at the map border
    (80 % of the beams outside the map, cond(H) ~ 1e10): chaotic for any other summation order, identical here"""
    g, o = pyr
    sc = pyramid_scene
    rng = np.random.default_rng(7)
    for q in range(8):
        init = sc.query_truth[q] + np.array([rng.uniform(-.5, .5), rng.uniform(-.5, .5), rng.uniform(-.5, .5)], np.float32)
        pg, cg = g.matchData(init, sc.query_scans[q])
        po, co = o.match(init, sc.query_scans[q])
        assert same(pg, po) and same(cg, co), q
    far = np.array([11.5, 9.0, 0.3], np.float32)
    pts = sc.query_scans[0]
    for it in (0, 3):
        pg, cg = g.match_level(0, far, pts, it)
        po, co = o.match_level(0, far, pts, it)
        assert same(pg, po) and same(cg, co), it


@pytest.mark.parametrize("form", ["auto", "cached", "cached-tail", "cached-long", "cached-9rows", "cached-5rows", "plane-layout",
                                  "one-wave-per-scan", "cached-13rows", "cached/rotating-owner", "cached-tail/rotating-owner",
                                  "cached-13rows/rotating-owner", "cached-9rows/rotating-owner", "cached-5rows/rotating-owner"])
def test_batch_ragged_bit_identical(capi, oracle_mod, pyramid_scene, kind, form, monkeypatch):
    """the batched entry: ragged CSR batch with empty, tiny, regular and over-long scans -- through the team kernel
    a small batch picks by itself ("auto"), the texel-cache exact form every quad-layout batch takes (gn_match_exact.h: every
    wavefront a producer, packed rotating chain jobs; 17 scans = two full workgroups and a partial one; "cached-tail" /
    "cached-long": scans four / nine rows longer than the 17 cached ones stream their tail rows), the plane layout (which has
    no texel-cache exact form) and the one-wavefront-per-scan form (HSM_EXACT_CACHED=0).  Round 2's producer / chain-wavefront
    form left the default library in round 4 (-DHSM_EXPERIMENTS).  Since round 5 a launch of at most 3072 scans -- this one --
    takes the texel-cache form WITH a chain-only fifth wavefront per workgroup (block 320); ".../rotating-owner" switches it
    off (HSM_EXACT_CHAIN_WAVE=0): the form larger launches take, whose producers run the chain jobs in turn (block 256)"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    rotating = form.endswith("/rotating-owner")
    form = form.split("/")[0]
    monkeypatch.setenv("HSM_EXACT_CHAIN_WAVE", "0" if rotating else "1")
    monkeypatch.setenv("HSM_EXACT_CACHED", "0" if form == "one-wave-per-scan" else "1")
    kw = {} if form == "auto" else {"waves_per_scan": 1}
    if form == "plane-layout":
        kw["layout"] = capi.LAYOUT_PLANE
    g = exact_gpu(capi, sc, o, **kw)
    rng = np.random.default_rng(5)
    scans, init = [], []
    for q in range(len(sc.query_scans)):
        full = sc.query_scans[q]
        # (fragments of a few beams make H singular: the reference then casts NaN coordinates to int and crashes)
        n = [full.shape[0], 0, 40, 64, 65, 700, 1081][q % 7]
        scans.append(full[np.linspace(0, full.shape[0] - 1, min(n, full.shape[0])).astype(int)] if n else full[:0])
        init.append(sc.query_init[q])
    long_scan = np.concatenate([sc.query_scans[3], sc.query_scans[3][::2] + np.float32(0.01)])  # 1622 beams > 64 * 17
    if form.endswith("-tail"):
        long_scan = long_scan[:1300]  # 21 rows: the texel-cache form streams rows 17..20
    elif form.startswith("cached") and form != "cached-long":  # ("cached-long": all 1622 beams, 26 rows)
        long_scan = long_scan[:1081]
    scans.append(long_scan)
    init.append(sc.query_init[3])
    if form in ("cached-9rows", "cached-5rows", "cached-13rows"):
        # the 9- and 5-row instantiations of the texel-cache exact form (short scans); 720 beams: the 17-row instantiation leaves
        # its round loop behind the longest scan's last row (chain-wavefront form: 1 / 2 / 11 / 12 rounds in these workgroups)
        cap = {"cached-9rows": 560, "cached-5rows": 300, "cached-13rows": 720}[form]
        scans = [sq[np.linspace(0, sq.shape[0] - 1, min(cap, sq.shape[0])).astype(int)] if sq.shape[0] else sq for sq in scans]
    init = np.stack(init)
    pts, offs = synth.pack_scans(scans)
    pb, cb = g.match_batch(init, pts, offs)
    if form in ("plane-layout", "one-wave-per-scan"):
        assert not g.last_launch_config()["texel_cache"], g.last_launch_config()
    if form.startswith("cached"):
        cfg = g.last_launch_config()
        assert cfg["texel_cache"] and cfg["block"] == (256 if rotating else 320), cfg
        assert ("chain wavefront" in cfg["kernel"]) == (not rotating), cfg
        if form.endswith("rows"):
            assert cfg["beams_per_lane"] == {"13rows": 13, "9rows": 9, "5rows": 5}[form.split("-")[1]], cfg
    for q, sq in enumerate(scans):
        po, co = o.match(init[q], sq, cov=np.zeros(9, np.float32))
        assert same(pb[q], po), (q, sq.shape[0])
        if sq.shape[0]:
            assert same(cb[q], co), q
    # shared-scan mode: pose hypotheses of ONE scan
    hyp = np.repeat(sc.query_init[3:4], 9, 0) + np.linspace(-0.05, 0.05, 9, dtype=np.float32)[:, None]
    ph, ch = g.match_batch(hyp, sc.query_scans[3], None)
    for k in range(9):
        po, co = o.match(hyp[k], sc.query_scans[3])
        assert same(ph[k], po) and same(ch[k], co), k


def test_mid_size_batch_of_long_scans_takes_one_wavefront_per_scan(capi, oracle_mod, pyramid_scene, kind):
    """300 scans of 1622 beams (26 rows: nine stream from memory in every step) in the default launch rule: more scans than the
    device has CUs -> one wavefront per scan + chain wavefront instead of teams; every pose equal to the reference's"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = exact_gpu(capi, sc, o)
    rng = np.random.default_rng(11)
    scans, init = [], []
    for q in range(300):
        a = sc.query_scans[q % len(sc.query_scans)]
        scans.append(np.concatenate([a, a[::2] + np.float32(0.01)]))
        init.append(sc.query_init[q % len(sc.query_init)] + rng.uniform(-0.02, 0.02, 3).astype(np.float32))
    init = np.stack(init).astype(np.float32)
    pts, offs = synth.pack_scans(scans)
    pb, cb = g.match_batch(init, pts, offs)
    cfg = g.last_launch_config()
    assert cfg["texel_cache"] and cfg["block"] == 320 and "chain wavefront" in cfg["kernel"], cfg
    for q in range(0, 300, 7):
        po, co = o.match(init[q], scans[q], cov=np.zeros(9, np.float32))
        assert same(pb[q], po) and same(cb[q], co), q
    g.close()


@pytest.mark.parametrize("chain_wave", ["1", "0"])
def test_batch_with_a_workgroup_of_empty_scans(capi, oracle_mod, pyramid_scene, kind, chain_wave, monkeypatch):
    """twelve scans, the middle four empty: that workgroup (four scans each, in both texel-cache exact forms) leaves before the
    first barrier with every wavefront -- the chain wavefront included --, its scans pass their start estimates through, and the
    workgroups around it are unaffected"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    monkeypatch.setenv("HSM_EXACT_CHAIN_WAVE", chain_wave)
    g = exact_gpu(capi, sc, o, waves_per_scan=1)
    scans = [sc.query_scans[q % len(sc.query_scans)] for q in range(12)]
    for q in range(4, 8):
        scans[q] = scans[q][:0]
    init = np.stack([sc.query_init[q % len(sc.query_init)] for q in range(12)])
    pts, offs = synth.pack_scans(scans)
    pb, cb = g.match_batch(init, pts, offs)
    cfg = g.last_launch_config()
    assert cfg["texel_cache"] and cfg["grid"] == 3 and cfg["block"] == (320 if chain_wave == "1" else 256), cfg
    for q, sq in enumerate(scans):
        po, co = o.match(init[q], sq, cov=np.zeros(9, np.float32))
        assert same(pb[q], po), q
        if sq.shape[0]:
            assert same(cb[q], co), q
        else:
            assert same(pb[q], init[q]) and not cb[q].any(), q
    g.close()


def test_dense_scan_bit_identical(capi, oracle_mod, pyramid_scene, kind):
    """16k-beam scans (configs[4] shape): the exact form keeps the scan on one 16-wave workgroup"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = exact_gpu(capi, sc, o)
    s = float(np.float32(1.0) / np.float32(sc.resolution))
    rng = np.random.default_rng(77)
    for n_beams in (4096, 16384):
        for q in range(2):
            pts = synth.make_scan(sc.world, sc.query_truth[q], n_beams, s, rng)
            pg, cg = g.matchData(sc.query_init[q], pts)
            assert g.last_launch_config()["waves_per_scan"] == 16
            po, co = o.match(sc.query_init[q], pts)
            assert same(pg, po) and same(cg, co), (n_beams, q)


@pytest.mark.parametrize("layout", ["quad", "plane"])
def test_dense_scan_forms_are_bit_identical(capi, oracle_mod, pyramid_scene, kind, layout, monkeypatch):
    """Dense scans in the reference's order, three forms of the same sums:
      * round 5, the default: gn_match_exact_dense_kernel -- one wavefront adds, fifteen produce one round ahead of it;
      * round 6, opt-in (HSM_EXACT_SPEC=1): gn_match_spec_kernel -- the nine chains cut into segments that run in parallel from
        speculated carries, stitched by the exact shift rule, re-run literally where the rule does not apply (csrc/spec_chain.h);
      * the 16-wavefront team form (HSM_EXACT_DENSE=0).
    Pose, covariance and every hook-trace record are bit-identical between the three and to the reference, for scan lengths
    on both sides of the round / segment boundaries, in the explicit EXACT mode and in the library default; a batch of dense
    scans takes the same form; the stitching counters show that both the shift and the re-run path ran"""
    import ctypes as C
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    lay = capi.LAYOUT_QUAD if layout == "quad" else capi.LAYOUT_PLANE
    monkeypatch.setenv("HSM_EXACT_DENSE_MIN", "1920")
    monkeypatch.setenv("HSM_EXACT_SPEC", "1")
    g = exact_gpu(capi, sc, o, layout=lay)
    monkeypatch.delenv("HSM_EXACT_SPEC")
    lit = exact_gpu(capi, sc, o, layout=lay)
    monkeypatch.setenv("HSM_EXACT_DENSE", "0")
    team = exact_gpu(capi, sc, o, layout=lay)
    monkeypatch.delenv("HSM_EXACT_DENSE")
    auto = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, layout=lay)  # the library default
    for lvl in range(sc.levels):
        auto.upload_level(lvl, *o.download_level(lvl))
    s = float(np.float32(1.0) / np.float32(sc.resolution))
    rng = np.random.default_rng(79)
    lib = capi.load_library()
    scans = []
    g.debug_spec_stats(True)
    for n_beams in (2561, 2880, 2881, 3584, 3585, 4096, 5000, 8192, 16384, 16385, 17290):  # (up to 2560 beams a single scan runs on 8 wavefronts)
        q = n_beams % 3
        pts = synth.make_scan(sc.world, sc.query_truth[q], n_beams, s, rng, pad_to_full=True)  # exactly n_beams endpoints
        assert pts.shape[0] == n_beams
        scans.append((q, pts))
        pg, cg = g.matchData(sc.query_init[q], pts)
        cfg = g.last_launch_config()
        assert cfg["kernel"] == "gn_match_spec_kernel" and cfg["block"] == 1024 and cfg["waves_per_scan"] == 16, (n_beams, pts.shape, cfg)
        pl, cl = lit.matchData(sc.query_init[q], pts)
        assert lit.last_launch_config()["kernel"] == "gn_match_exact_dense_kernel", lit.last_launch_config()
        pt, ct = team.matchData(sc.query_init[q], pts)
        assert team.last_launch_config()["kernel"].startswith("gn_match_kernel"), team.last_launch_config()
        po, co = o.match(sc.query_init[q], pts)
        assert same(pg, po) and same(cg, co), (n_beams, "vs the reference")
        assert same(pl, po) and same(cl, co), (n_beams, "literal dense form vs the reference")
        assert same(pg, pt) and same(cg, ct), (n_beams, "vs the team form")
        pa, ca = auto.matchData(sc.query_init[q], pts)
        assert same(pa, po) and same(ca, co), (n_beams, "library default")
        if pts.shape[0] >= 4096:
            assert auto.last_launch_config()["kernel"] == "gn_match_exact_dense_kernel"
        # the hook trace (draw / debug interfaces of the facade): 14 records, each the reference's step
        a = np.ascontiguousarray(pts, np.float32)
        tr = {}
        for name, ctx in (("spec", g), ("literal", lit), ("team", team)):
            pose, cov, trace, nst = np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(14 * 12, np.float32), C.c_int(0)
            capi._check(lib.hsm_match_trace(ctx._h, sc.query_init[q], a.ctypes.data, a.shape[0], np.zeros(2, np.float32), pose, cov, trace, 14,
                                            C.byref(nst)), "hsm_match_trace")
            assert nst.value == 14 and same(pose, po) and same(cov, co), (name, n_beams)
            tr[name] = trace
        assert same(tr["spec"], tr["team"]) and same(tr["literal"], tr["team"]), n_beams
    walked, exact, shifted, rerun = g.debug_spec_stats(False)
    assert walked > 0 and shifted + rerun == walked and shifted > 0 and rerun > 0, (walked, exact, shifted, rerun)
    assert shifted / walked > 0.8, (walked, shifted, rerun)  # real chains: 10-16 re-runs per chain of 55-512 segments (tools/study/spec_chain_stats.py)
    # single-level matchData (ScanMatcher::matchData with an explicit iteration count) on a dense scan, library default
    q, pts = scans[-2]
    lvl_pts = pts * np.float32(0.5)
    pol, col = o.match_level(1, sc.query_init[q], lvl_pts, 7)
    for ctx, name in ((auto, "gn_match_exact_dense_kernel"), (g, "gn_match_spec_kernel")):
        pl, cl = ctx.match_level(1, sc.query_init[q], lvl_pts, 7)
        assert ctx.last_launch_config()["kernel"] == name
        assert same(pl, pol) and same(cl, col), ("single-level dense match", name)
    # a batch of dense scans (ragged lengths, host offsets: the bound of the lengths is known): one workgroup per scan
    sel = [sc_ for sc_ in scans if sc_[1].shape[0] >= 4000][:5]
    pts, offs = synth.pack_scans([p for _, p in sel])
    init = np.stack([sc.query_init[q] for q, _ in sel])
    for ctx, name in ((g, "gn_match_spec_kernel"), (lit, "gn_match_exact_dense_kernel")):
        pb, cb = ctx.match_batch(init, pts, offs)
        assert ctx.last_launch_config()["kernel"] == name and ctx.last_launch_config()["grid"] == len(sel)
        for k, (q, p) in enumerate(sel):
            po, co = o.match(sc.query_init[q], p)
            assert same(pb[k], po) and same(cb[k], co), (name, k)
    # device-resident CSR offsets: the lengths are not known to the host, the shared_n argument is only a hint -- the form whose
    # scratch is sized from a bound is not taken
    import torch
    d_init, d_pts, d_offs = (torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (init, pts, offs))
    d_pose = torch.zeros((len(sel), 3), dtype=torch.float32, device="cuda")
    g.match_batch_device(len(sel), d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), 4096, d_pose.data_ptr(), 0, 0)
    g.synchronize()
    assert g.last_launch_config()["kernel"] == "gn_match_exact_dense_kernel"
    assert same(d_pose.cpu().numpy(), pb)
    for ctx in (g, lit, team, auto):
        ctx.close()


def test_speculative_carry_form_on_short_scans_and_many_poses(capi, oracle_mod, pyramid_scene, kind, monkeypatch):
    """the same form forced onto the node's scan lengths (HSM_EXACT_DENSE_MIN=128: segments of 32 beams) over every query
    scan from near and far starts: bit-identical to the reference -- the shift rule is exact, whatever the data"""
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    monkeypatch.setenv("HSM_EXACT_DENSE_MIN", "128")
    monkeypatch.setenv("HSM_EXACT_SPEC", "1")
    g = exact_gpu(capi, sc, o)
    rng = np.random.default_rng(5)
    n = 0
    for q in range(len(sc.query_scans)):
        for k in range(4):
            init = sc.query_init[q].copy()  # (starts the reference itself survives: it indexes the grid with (int)NaN once it diverges)
            init[:2] += rng.uniform(-0.1, 0.1, 2).astype(np.float32) * np.float32(k)
            init[2] += np.float32(rng.uniform(-0.04, 0.04) * k)
            pts = sc.query_scans[q][: [len(sc.query_scans[q]), 720, 361, 130][k]]
            pg, cg = g.matchData(init, pts)
            if len(pts) >= 320:  # (shorter scans run on fewer wavefronts: the team form)
                assert g.last_launch_config()["kernel"] == "gn_match_spec_kernel", (len(pts), g.last_launch_config())
                n += 1
            po, co = o.match(init, pts)
            if np.isnan(po).any():  # the reference's own iteration diverged (a far start on few beams): NaN payloads are not pinned
                assert np.array_equal(np.isnan(pg), np.isnan(po)), (q, k)
                continue
            assert same(pg, po) and same(cg, co), (q, k)
    assert n >= 2 * len(sc.query_scans)
    g.close()


@pytest.mark.parametrize("layout", ["quad", "plane"])
def test_single_scan_on_chip_speculative_form(capi, oracle_mod, pyramid_scene, kind, layout, monkeypatch):
    """HSM_EXACT_SPEC1=1: ONE scan of the node's size through hsm_match in the reference's order, the nine chains cut into (up to)
    64 segments that live in the chain wavefronts' registers (gn_match_spec1_kernel): pose, covariance and every hook-trace record
    bit-identical to the reference and to the team form, for lengths from 321 to 2048 beams (segment lengths 8 .. 32, ragged last
    segments), near and far starts, single-level matches"""
    import ctypes as C
    from hector_slam_amd import synth
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    lay = capi.LAYOUT_QUAD if layout == "quad" else capi.LAYOUT_PLANE
    team = exact_gpu(capi, sc, o, layout=lay)
    monkeypatch.setenv("HSM_EXACT_SPEC1", "1")
    g = exact_gpu(capi, sc, o, layout=lay)
    lib = capi.load_library()
    s = float(np.float32(1.0) / np.float32(sc.resolution))
    rng = np.random.default_rng(11)
    for n_beams in (321, 512, 513, 720, 1080, 1081, 1082, 1500, 2047, 2048):
        q = n_beams % len(sc.query_scans)
        pts = synth.make_scan(sc.world, sc.query_truth[q], n_beams, s, rng, pad_to_full=True)
        assert pts.shape[0] == n_beams
        for k in range(2):
            init = sc.query_init[q].copy()
            init[:2] += rng.uniform(-0.08, 0.08, 2).astype(np.float32) * np.float32(k)
            pg, cg = g.matchData(init, pts)
            assert g.last_launch_config()["kernel"] == "gn_match_spec1_kernel", (n_beams, g.last_launch_config())
            po, co = o.match(init, pts)
            assert same(pg, po) and same(cg, co), (n_beams, k, "vs the reference")
            pt, ct = team.matchData(init, pts)
            assert same(pg, pt) and same(cg, ct), (n_beams, k, "vs the team form")
        a = np.ascontiguousarray(pts, np.float32)
        tr = {}
        for name, ctx in (("spec1", g), ("team", team)):
            pose, cov, trace, nst = np.zeros(3, np.float32), np.zeros(9, np.float32), np.zeros(14 * 12, np.float32), C.c_int(0)
            capi._check(lib.hsm_match_trace(ctx._h, sc.query_init[q], a.ctypes.data, a.shape[0], np.zeros(2, np.float32), pose, cov, trace, 14,
                                            C.byref(nst)), "hsm_match_trace")
            assert nst.value == 14
            tr[name] = (pose, cov, trace)
        assert all(same(x, y) for x, y in zip(tr["spec1"], tr["team"])), n_beams
    # every query scan of the scene, and a single-level match
    for q in range(len(sc.query_scans)):
        pg, cg = g.matchData(sc.query_init[q], sc.query_scans[q])
        po, co = o.match(sc.query_init[q], sc.query_scans[q])
        assert same(pg, po) and same(cg, co), q
    lvl_pts = sc.query_scans[0] * np.float32(0.5)
    pl, cl = g.match_level(1, sc.query_init[0], lvl_pts, 7)
    pol, col = o.match_level(1, sc.query_init[0], lvl_pts, 7)
    assert same(pl, pol) and same(cl, col)
    g.close()
    team.close()


def test_slam_loop_from_empty_map_bit_identical(capi, oracle_mod, pyramid_scene, kind):
    """HectorSlamProcessor::update from an EMPTY map, 30 scans: identical poses at every step, hence identical update
    decisions and bit-identical maps on all levels at the end -- the whole SLAM state, not just one match"""
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc, build=False)
    o.proc_set_thresholds(0.05, 0.02)
    p = capi.HectorSlamProcessor(sc.resolution, sc.map_size, sc.map_size, (0.5, 0.5), sc.levels, parity=capi.PARITY_EXACT)
    p.setUpdateFactorFree(0.4)
    p.setUpdateFactorOccupied(0.9)
    p.setMapUpdateMinDistDiff(0.05)
    p.setMapUpdateMinAngleDiff(0.02)
    hint = sc.build_poses[0].copy()
    for t in range(30):
        o.proc_update(sc.build_scans[t], hint)
        p.update(sc.build_scans[t], hint)
        po, co = o.proc_last_pose()
        assert same(p.getLastScanMatchPose(), po) and same(p.getLastScanMatchCovariance(), co), t
        hint = po + (sc.build_poses[t + 1] - sc.build_poses[t])
    for lvl in range(sc.levels):
        a, b = p.mapRep.download_level(lvl), o.download_level(lvl)
        assert same(a[0], b[0]) and np.array_equal(a[1], b[1]), lvl


def test_randomised_geometries_bit_identical(capi, oracle_mod, kind):
    """odd map sizes, 1-4 levels, off-centre start coordinates, rooms larger than the map, maps made of one or two
    scans, laser origins off the robot centre: no 'settled' or 'well conditioned' predicate -- every step is equal"""
    from hector_slam_amd import synth
    rng = np.random.default_rng(2024)
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
        g = capi.MapRepMultiMap(res, size, size, levels, start, parity=capi.PARITY_EXACT)
        o.set_update_factor_free(free)
        g.setUpdateFactorFree(free)
        o.set_update_factor_occupied(occ)
        g.setUpdateFactorOccupied(occ)
        pose = poses[0].copy()
        for t in range(14):
            hint = pose + (poses[t] - poses[max(t - 1, 0)])
            po, co = o.match(hint, scans[t], origos[t])
            pg, cg = g.matchData(hint, scans[t], None, origos[t])
            if np.isfinite(po).all():
                assert same(pg, po) and same(cg, co), (trial, size, res, levels, t, pg, po)
            else:  # a singular H: the reference divides by a zero determinant; NaN payloads are not pinned
                assert np.array_equal(np.isnan(pg), np.isnan(po))
                po = hint
            o.update_by_scan(po, scans[t], origos[t])
            o.on_map_updated()
            g.updateByScan(scans[t], po, origos[t])
            pose = po
        for lvl in range(levels):
            a, b = g.download_level(lvl), o.download_level(lvl)
            assert same(a[0], b[0]) and np.array_equal(a[1], b[1]), (trial, lvl)


def test_fast_mode_deviation_is_measured_against_exact_mode(capi, oracle_mod, pyramid_scene, kind):
    """the default (fast) summation on the same context: same products, another order.  Its deviation from the exact
    mode IS its deviation from the reference; bound stated here, statistics printed."""
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = exact_gpu(capi, sc, o)
    from hector_slam_amd import synth
    pts, offs = synth.pack_scans(sc.query_scans)
    pe, ce = g.match_batch(sc.query_init, pts, offs)
    g.set_parity(capi.PARITY_FAST)
    pf, cf = g.match_batch(sc.query_init, pts, offs)
    assert g.last_launch_config()["parity"] == "fast"
    g.set_parity(capi.PARITY_EXACT)
    pe2, _ = g.match_batch(sc.query_init, pts, offs)
    assert same(pe, pe2)
    assert same(pe, o.match_many(sc.query_init, pts, offs))
    d = np.abs(pf.astype(np.float64) - pe)
    print(f"fast vs exact: max |dxy| {d[:, :2].max():.2e} m, max |dtheta| {d[:, 2].max():.2e} rad, "
          f"bit-identical {(bits(pf) == bits(pe)).all(1).mean():.3f}")
    assert d[:, :2].max() <= 1e-4 and d[:, 2].max() <= 1e-4
    assert np.abs(cf - ce).max() <= 1e-4 * np.abs(ce).max()


def test_committed_golden_vectors_bit_identical(capi):
    """tests/golden/*.npz hold outputs of the reference's own headers (make_golden.py): poses, covariances and the
    per-step H / dTr of the golden matches are reproduced bit for bit -- no oracle library involved at all"""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g1 = np.load(os.path.join(gold, "config1_181beam_256map.npz"))
    g = capi.MapRepMultiMap(float(g1["resolution"]), int(g1["map_size"]), int(g1["map_size"]), 1, parity=capi.PARITY_EXACT)
    g.upload_level(0, g1["logodds"], g1["update_index"])
    for q in range(4):
        pts = g1[f"q{q}_pts"]
        pose, cov = g.match_level(0, g1[f"q{q}_init"], pts, 5)
        assert same(pose, g1[f"q{q}_pose"]) and same(cov, g1[f"q{q}_cov"]), q
        for k in range(7):
            H, d = g.hessian_derivs(0, g1[f"q{q}_step_pose_map"][k], pts)
            assert same(H, g1[f"q{q}_step_H"][k]) and same(d, g1[f"q{q}_step_dTr"][k]), (q, k)
    g2 = np.load(os.path.join(gold, "pyramid_1081beam_512map.npz"))
    g = capi.MapRepMultiMap(float(g2["resolution"]), int(g2["map_size"]), int(g2["map_size"]), 3, parity=capi.PARITY_EXACT)
    for lvl in range(3):
        g.upload_level(lvl, g2[f"logodds{lvl}"], g2[f"update_index{lvl}"])
    for q in range(8):
        pose, cov = g.matchData(g2["init"][q], g2[f"q{q}_pts"])
        assert same(pose, g2["pose"][q]) and same(cov, g2["cov"][q]), q


@pytest.mark.parametrize("layout", ["quad", "plane"])
def test_likelihood_residual_and_sigma_point_covariance_bit_identical(capi, oracle_mod, pyramid_scene, kind, layout):
    """f3 in exact mode: getResidualForState's chain (residual += funval, beam 0 .. n-1) in the reference's order ->
    residuals, likelihoods, the seven sigma-point likelihoods and both covariance matrices bit for bit"""
    sc = pyramid_scene
    o = make_oracle(oracle_mod, kind, sc)
    g = exact_gpu(capi, sc, o, layout=capi.LAYOUT_QUAD if layout == "quad" else capi.LAYOUT_PLANE)
    rng = np.random.default_rng(9)
    for lvl in range(sc.levels):
        f = np.float32(1.0 / 2 ** lvl)
        for q in range(3):
            pm = o.map_coords_pose(lvl, sc.query_truth[q])
            cloud = (pm[None, :] + rng.normal(0, [2.0, 2.0, 0.05], (1500, 3))).astype(np.float32)
            far = np.array([[-50.0, 3.0, 0.1], [1e6, 1e6, 0.0]], np.float32)  # (almost) everything out of the map
            states = np.concatenate([pm[None, :], cloud, far])
            pts = sc.query_scans[q][: [1081, 700, 65][q]]
            assert same(g.likelihood_states(lvl, states, pts), o.likelihood_states(lvl, states, pts * f)), (lvl, q)
            assert same(g.residual_states(lvl, states, pts), o.residual_states(lvl, states, pts * f)), (lvl, q)
            poses = states[:40]
            cm, cw, lh = g.covariance_for_poses(lvl, poses, pts)
            om, ow, ol = o.covariance_for_poses(lvl, poses, pts * f)
            assert same(lh, ol), (lvl, q)
            assert same(cm, om) and same(cw, ow), (lvl, q)

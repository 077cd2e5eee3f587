"""Property tests (hypothesis): random map geometry, pyramid depth, update factors, sensor fan,
start coordinates, laser origin and a short SLAM run from an empty map.

  CPU  ("ho" vs "hr"): the restatement equals the reference's own headers bit for bit at every step -- poses,
       covariances, maps -- whatever the geometry, and the condition number of the last Hessian is reported.
  GPU  (HSM_PARITY_EXACT vs the reference-compiled checker where present): the same, through the C ABI.

Every example is a whole match/update loop, so the number of examples is kept small; the strategies draw the
structure (sizes, levels, seeds), the worlds and scans come from hector_slam_amd.synth."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, assume, given, seed, settings
from hypothesis import strategies as st

from conftest import bits, oracle_kinds

geometry = st.fixed_dictionaries({
    "size": st.sampled_from([64, 96, 125, 200, 256, 333]),
    "levels": st.integers(1, 4),
    "res": st.sampled_from([0.025, 0.05, 0.1, 0.2]),
    "start": st.tuples(st.floats(0.3, 0.7), st.floats(0.3, 0.7)),
    "free": st.floats(0.3, 0.49),
    "occ": st.floats(0.55, 0.95),
    "beams": st.sampled_from([90, 181, 400, 1081]),
    "grow": st.sampled_from([0.6, 0.9, 1.15]),
    "seed": st.integers(0, 2 ** 20),
    "origo": st.tuples(st.floats(-2, 2), st.floats(-2, 2)),
})


def reference_undefined(impl) -> bool:
    """the restatement ("ho") counts map reads with a NaN coordinate -- where the reference's own Gauss-Newton step has divided by
    a zero determinant and the reference then indexes the grid with (int)NaN (OccGridMapUtil.h:295,302: undefined behaviour, it
    segfaults; hypothesis finds such geometries within ~100 examples).  An input like that has no reference result to compare with."""
    o = impl.get("keep")
    return hasattr(o, "undefined_reads") and o.undefined_reads() > 0


def run_loop(g, make_a, make_b, steps=8, make_guard=None):
    """the same short SLAM loop on two implementations; yields (step, pose_a, cov_a, pose_b, cov_b).  make_guard: a restatement
    instance stepped AHEAD of b when b is the reference itself, so that an input on which the reference would crash is discarded
    (hypothesis.assume) instead of taking the test process down with it"""
    from hector_slam_amd import synth
    size, levels, res = g["size"], g["levels"], g["res"]
    while (size >> (levels - 1)) < 8:
        levels -= 1
    ext = size * res
    world = synth.World.make(ext * g["grow"], ext * g["grow"] * 0.75, n_boxes=3, seed=g["seed"], keep_clear=0.5)
    s = float(np.float32(1.0) / np.float32(res))
    poses = synth.loop_trajectory(world, steps + 1, frac=0.25).astype(np.float32)
    noise = np.random.default_rng(g["seed"])
    scans = [synth.make_scan(world, p, g["beams"], s, noise, range_max=min(30.0, ext)) for p in poses]
    origo = np.asarray(g["origo"], np.float32)
    a, b = make_a(res, size, levels, g["start"], g["free"], g["occ"]), make_b(res, size, levels, g["start"], g["free"], g["occ"])
    guard = make_guard(res, size, levels, g["start"], g["free"], g["occ"]) if make_guard else None
    pose = poses[0].copy()
    cond = None
    for t in range(steps):
        hint = pose + (poses[t] - poses[max(t - 1, 0)])
        pa, ca = a["match"](hint, scans[t], origo)
        assume(not reference_undefined(a))
        if guard is not None:
            guard["match"](hint, scans[t], origo)
            assume(not reference_undefined(guard))
        pb, cb = b["match"](hint, scans[t], origo)
        assume(not reference_undefined(b))
        if not np.isfinite(pa).all():  # singular H: the reference divides by a zero determinant (NaN payloads not pinned)
            assert np.array_equal(np.isnan(pa), np.isnan(pb))
            return cond
        assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(ca), bits(cb)), (g, t, pa, pb)
        H = ca.reshape(3, 3).astype(np.float64)
        cond = float(np.linalg.cond(H)) if np.abs(H).max() > 0 else float("inf")
        a["update"](pa, scans[t], origo)
        b["update"](pa, scans[t], origo)
        if guard is not None:
            guard["update"](pa, scans[t], origo)
        for impl in (a, b):
            if "check" in impl:
                impl["check"](g, t)
        pose = pa
    for lvl in range(levels):
        la, lb = a["level"](lvl), b["level"](lvl)
        assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1]), (g, lvl)
    return cond


# Round 4: the DENSE form of updateByScan (>= 4096 beams on maps whose rows are a multiple of 64 cells: byte marks +
# block-owned apply pass, map_update.h) under the same property -- with the map's start coordinates drawn from [-0.04, 1.04],
# so the robot's loop runs along, across and outside the map's borders (start < 0 or > 1 puts the world origin off the map).
# Both tests below draw from the SAME seeded strategy (hypothesis' @seed, no example database), so the examples the GPU box
# runs are exactly the ones the CPU pin (restatement == reference headers) ran here first.
dense_geometry = st.fixed_dictionaries({
    "size": st.sampled_from([64, 128, 192, 256, 320]),
    "levels": st.integers(1, 3),
    "res": st.sampled_from([0.05, 0.1]),
    "start": st.tuples(st.floats(-0.04, 1.04), st.floats(-0.04, 1.04)),
    "free": st.floats(0.3, 0.49),
    "occ": st.floats(0.55, 0.95),
    "beams": st.sampled_from([4096, 6000, 16384]),
    "grow": st.sampled_from([0.6, 0.9, 1.15]),
    "seed": st.integers(0, 2 ** 20),
    "origo": st.tuples(st.floats(-2, 2), st.floats(-2, 2)),
})
DENSE_SEED = 20260924


def oracle_impl(pyoracle, kind):
    def make(res, size, levels, start, free, occ):
        o = pyoracle.Oracle(kind, res, size, size, levels, start)
        o.set_update_factor_free(free)
        o.set_update_factor_occupied(occ)

        def upd(p, sc, og):
            o.update_by_scan(p, sc, og)
            o.on_map_updated()
        return {"match": lambda h, sc, og: o.match(h, sc, og), "update": upd, "level": o.download_level, "keep": o}
    return make


@pytest.mark.skipif("hr" not in oracle_kinds(), reason="oracle/_ref/libhector_ref.so not built (needs /root/reference)")
@settings(max_examples=int(os.environ.get("HSM_HYPOTHESIS_EXAMPLES_CPU", "40")), deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(g=geometry)
def test_restatement_equals_reference_for_random_geometries(oracle_mod, g):
    cond = run_loop(g, oracle_impl(oracle_mod, "ho"), oracle_impl(oracle_mod, "hr"))
    print(f"cond(H) of the last step: {cond}")


# 60 examples per strategy by default (18 ms each on the GPU box: round 6; rounds 4-5 ran 12 / 10 unless HSM_HYPOTHESIS_EXAMPLES said
# otherwise); the long runs under profiles/ set 2 000 ... 20 000
GPU_EXAMPLES = int(os.environ.get("HSM_HYPOTHESIS_EXAMPLES", "0"))


@pytest.mark.gpu
@settings(max_examples=GPU_EXAMPLES or 60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(g=geometry)
def test_gpu_default_mode_equals_reference_for_random_geometries(oracle_mod, g):
    """the library DEFAULT (HSM_PARITY_AUTO; no parity argument): every pose, covariance and map of the loop bit-identical"""
    from hector_slam_amd import capi

    def make_gpu(res, size, levels, start, free, occ):
        m = capi.MapRepMultiMap(res, size, size, levels, start)
        assert m.parity() == capi.PARITY_AUTO
        m.setUpdateFactorFree(free)
        m.setUpdateFactorOccupied(occ)
        return {"match": lambda h, sc, og: m.matchData(h, sc, None, og), "update": lambda p, sc, og: m.updateByScan(sc, p, og),
                "level": m.download_level, "keep": m}
    kind = oracle_kinds()[-1]
    cond = run_loop(g, make_gpu, oracle_impl(oracle_mod, kind), make_guard=oracle_impl(oracle_mod, "ho") if kind == "hr" else None)
    print(f"cond(H) of the last step: {cond}")


@pytest.mark.skipif("hr" not in oracle_kinds(), reason="oracle/_ref/libhector_ref.so not built (needs /root/reference)")
@seed(DENSE_SEED)
@settings(max_examples=10, deadline=None, database=None, suppress_health_check=list(HealthCheck))
@given(g=dense_geometry)
def test_restatement_equals_reference_for_dense_scans_at_the_borders(oracle_mod, g):
    run_loop(g, oracle_impl(oracle_mod, "ho"), oracle_impl(oracle_mod, "hr"), steps=6)


@pytest.mark.gpu
@seed(DENSE_SEED)
@settings(max_examples=GPU_EXAMPLES or 60, deadline=None, database=None, suppress_health_check=list(HealthCheck))
@given(g=dense_geometry)
def test_gpu_dense_update_equals_reference_at_the_borders(oracle_mod, g):
    """exact mode: every pose, covariance and map of the loop bit-identical to the reference, and after every update the
    dense path's byte map and the keyed path's end-cell bitmap are all zero again"""
    from hector_slam_amd import capi

    def make_gpu(res, size, levels, start, free, occ):
        m = capi.MapRepMultiMap(res, size, size, levels, start, parity=capi.PARITY_EXACT)
        m.setUpdateFactorFree(free)
        m.setUpdateFactorOccupied(occ)

        def check(gg, t):
            for lvl in range(m.getMapLevels()):
                assert m.debug_marks_nonzero(lvl) == (0, 0), (gg, t, lvl)
        return {"match": lambda h, sc, og: m.matchData(h, sc, None, og), "update": lambda p, sc, og: m.updateByScan(sc, p, og),
                "level": m.download_level, "keep": m, "check": check}
    kind = oracle_kinds()[-1]
    run_loop(g, make_gpu, oracle_impl(oracle_mod, kind), steps=6, make_guard=oracle_impl(oracle_mod, "ho") if kind == "hr" else None)


# Round 5: the BATCHED entry in the default mode.  A batch of up to 3072 scans of up to 1088 beams runs one wavefront per scan
# plus a chain-only fifth wavefront per workgroup (gn_match_exact.h, CW; the 5- / 9- / 13- / 17-row instantiations by the
# longest scan); this property draws the map, the pyramid, the fan, the batch size and a ragged set of scan lengths (empty
# scans and fragments included) and holds every pose and covariance of the batch to the reference's bits.
batch_geometry = st.fixed_dictionaries({
    "size": st.sampled_from([64, 128, 200, 256, 512]),
    "levels": st.integers(1, 3),
    "res": st.sampled_from([0.05, 0.1]),
    "start": st.tuples(st.floats(0.3, 0.7), st.floats(0.3, 0.7)),
    "free": st.floats(0.3, 0.49),
    "occ": st.floats(0.55, 0.95),
    "beams": st.sampled_from([90, 181, 300, 400, 560, 720, 1081]),
    "grow": st.sampled_from([0.6, 0.9, 1.15]),
    "seed": st.integers(0, 2 ** 20),
    "batch": st.integers(1, 48),
    "ragged": st.booleans(),
})


@pytest.mark.gpu
@settings(max_examples=GPU_EXAMPLES or 60, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(g=batch_geometry)
def test_gpu_default_mode_batches_equal_reference_for_random_geometries(oracle_mod, g):
    from hector_slam_amd import capi, synth
    size, levels, res = g["size"], g["levels"], g["res"]
    while (size >> (levels - 1)) < 8:
        levels -= 1
    ext = size * res
    world = synth.World.make(ext * g["grow"], ext * g["grow"] * 0.75, n_boxes=3, seed=g["seed"], keep_clear=0.5)
    s = float(np.float32(1.0) / np.float32(res))
    B, n_build = g["batch"], 6
    poses = synth.loop_trajectory(world, n_build + B, frac=0.25).astype(np.float32)
    rng = np.random.default_rng(g["seed"])
    scans = [synth.make_scan(world, p, g["beams"], s, rng, range_max=min(30.0, ext)) for p in poses]
    m = capi.MapRepMultiMap(res, size, size, levels, g["start"])
    assert m.parity() == capi.PARITY_AUTO
    kind = oracle_kinds()[-1]
    impls = [oracle_impl(oracle_mod, kind)(res, size, levels, g["start"], g["free"], g["occ"])]
    if kind == "hr":  # the restatement runs ahead of the reference and says where the reference would crash
        impls.insert(0, oracle_impl(oracle_mod, "ho")(res, size, levels, g["start"], g["free"], g["occ"]))
    m.setUpdateFactorFree(g["free"])
    m.setUpdateFactorOccupied(g["occ"])
    zero = np.zeros(2, np.float32)
    for t in range(n_build):
        m.updateByScan(scans[t], poses[t])
        for o in impls:
            o["update"](poses[t], scans[t], zero)
    for lvl in range(levels):
        la, lb = m.download_level(lvl), impls[-1]["level"](lvl)
        assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1]), (g, lvl)
    query, init = [], []
    for q in range(B):
        sc = scans[n_build + q]
        if g["ragged"] and sc.shape[0] > 0:
            n = int(rng.choice([0, 1, 7, 64, 65, sc.shape[0] // 2, sc.shape[0]]))
            sc = sc[np.sort(rng.choice(sc.shape[0], size=min(n, sc.shape[0]), replace=False))]
        query.append(np.ascontiguousarray(sc, np.float32))
        init.append(poses[n_build + q] + np.array([rng.uniform(-0.05, 0.05), rng.uniform(-0.05, 0.05), rng.uniform(-0.02, 0.02)], np.float32))
    init = np.stack(init).astype(np.float32)
    pts, offs = synth.pack_scans(query)
    pose, cov = m.match_batch(init, pts, offs)
    cfg = m.last_launch_config()
    assert cfg["parity_effective"] == "exact", cfg
    if max(q.shape[0] for q in query) <= 1088 and cfg["texel_cache"]:
        assert "chain wavefront" in cfg["kernel"] and cfg["block"] == 320, cfg
    for q in range(B):
        for o in impls:
            po, co = o["match"](init[q], query[q], zero)
            assume(not reference_undefined(o))
        if not np.isfinite(po).all():  # singular H: the reference divides by a zero determinant (NaN payloads not pinned)
            assert np.array_equal(np.isnan(pose[q]), np.isnan(po)), (g, q)
            continue
        assert np.array_equal(bits(pose[q]), bits(po)), (g, q, query[q].shape, pose[q], po)
        if query[q].shape[0]:
            assert np.array_equal(bits(cov[q]), bits(co)), (g, q)
    m.close()

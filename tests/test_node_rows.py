"""Rows next to the hot path: LaserScan ingestion (rosLaserScanToDataContainer,
hector_mapping/src/HectorMappingRos.cpp:483-507), PointCloud ingestion (rosPointCloudToDataContainer, :509-542,
the node's default with use_tf_scan_transformation) and occupancy export (publishMap's cell loop, :449-468).
Integer / byte / index work: BIT-EXACT against the oracle -- and, since round 4, against the reference's ROS node source
itself: HectorMappingRos.cpp compiled UNMODIFIED through roscpp / tf / boost stand-ins (oracle/node_shim.cpp,
oracle/_ref/libhector_node_ref.so, `pyoracle.NodeRef`), which pins the restatement of these rows."""
import numpy as np
import pytest

from conftest import bits, make_oracle


def synthetic_ranges(rng, n, lo=0.0, hi=35.0):
    r = rng.uniform(lo, hi, n).astype(np.float32)
    # sprinkle the values a real driver produces: inf (no return), NaN, 0, exactly the gates
    idx = rng.choice(n, size=max(n // 10, 1), replace=False)
    r[idx[0::5]] = np.inf
    r[idx[1::5]] = np.nan
    r[idx[2::5]] = 0.0
    r[idx[3::5]] = np.float32(0.4)
    r[idx[4::5]] = np.float32(30.0) - np.float32(0.1)
    return r


def test_oracle_node_rows_restatement_matches_reference_types(oracle_mod, small_scene):
    """CPU: the two checkers agree (occupancy goes through the reference's own isFree/isOccupied in hr)"""
    if not oracle_mod.available("hr"):
        pytest.skip("oracle/_ref not built")
    sc = small_scene
    a, b = make_oracle(oracle_mod, "ho", sc), make_oracle(oracle_mod, "hr", sc)
    ga, gb = a.occupancy_grid(0), b.occupancy_grid(0)
    assert np.array_equal(ga, gb) and set(np.unique(ga)) == {-1, 0, 100}
    rng = np.random.default_rng(1)
    for n in (0, 1, 181, 1081):
        r = synthetic_ranges(rng, n) if n else np.zeros(0, np.float32)
        pa = a.laser_scan_to_container(r, -2.35619449, 0.00436332, 0.4, 30.0, 20.0)
        pb = b.laser_scan_to_container(r, -2.35619449, 0.00436332, 0.4, 30.0, 20.0)
        assert np.array_equal(bits(pa), bits(pb))
        if n:
            keep = (r > np.float32(0.4)) & (r < np.float32(30.0) - np.float32(0.1))
            assert pa.shape[0] == keep.sum()


def rigid_rows(rng, tilt=0.05, shift=0.3):
    """a laser->base tf::Transform as 12 doubles [R | t] (small roll/pitch, any yaw)"""
    r, p, y = rng.uniform(-tilt, tilt), rng.uniform(-tilt, tilt), rng.uniform(-np.pi, np.pi)
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    R = np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                  [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                  [-sp, cp * sr, cp * cr]])
    t = rng.uniform(-shift, shift, 3)
    return np.concatenate([R, t[:, None]], 1).reshape(12).astype(np.float64)


def synthetic_cloud(rng, n):
    c = np.zeros((n, 3), np.float32)
    rad = rng.uniform(0.0, 35.0, n)
    ang = rng.uniform(-np.pi, np.pi, n)
    c[:, 0], c[:, 1] = rad * np.cos(ang), rad * np.sin(ang)
    c[:, 2] = rng.normal(0, 0.4, n)
    if n >= 10:
        idx = rng.choice(n, size=max(n // 10, 4), replace=False)
        c[idx[0::4], 0] = np.nan
        c[idx[1::4], 1] = np.inf
        c[idx[2::4], :2] = rng.uniform(-0.7, 0.7, (len(idx[2::4]), 2))  # the x<0 && d2<0.5 rule
        c[idx[3::4], 2] = rng.choice([-1.0, 1.0, 2.5, -3.0], len(idx[3::4]))  # exactly on / beyond the z gate
    return c


def numpy_point_cloud_to_container(c, T, smin, smax, zmin, zmax, scale):
    """independent vectorised statement of HectorMappingRos.cpp:509-542 (float gates, double tf)"""
    c = c.astype(np.float32)
    T = T.reshape(3, 4)
    with np.errstate(invalid="ignore", over="ignore"):
        d2 = c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1]
        keep = (d2 > np.float32(smin)) & (d2 < np.float32(smax)) & ~((c[:, 0] < 0) & (d2 < np.float32(0.5)))
        v = c.astype(np.float64)
        b = ((T[:, 0] * v[:, 0:1] + T[:, 1] * v[:, 1:2]) + T[:, 2] * v[:, 2:3]) + T[:, 3]
        zl = (b[:, 2] - T[2, 3]).astype(np.float32)
        keep &= (zl > np.float32(zmin)) & (zl < np.float32(zmax))
        out = b[keep, :2].astype(np.float32) * np.float32(scale)
    return out, (T[:2, 3].astype(np.float32) * np.float32(scale))


def test_oracle_point_cloud_restatement_is_pinned(oracle_mod):
    """CPU: ho == hr == an independent numpy statement of :509-542; identity tf + flat cloud + open gates
    reproduces the cloud; projectLaser + cloud ingestion agrees with the LaserScan ingestion (:483-507) to
    float rounding (double vs float trig)"""
    kinds = [k for k in ("ho", "hr") if oracle_mod.available(k)]
    os_ = [oracle_mod.Oracle(k, 0.05, 64, 64, 1) for k in kinds]
    rng = np.random.default_rng(11)
    for n in (0, 1, 10, 1081, 5000):
        c = synthetic_cloud(rng, n)
        T = rigid_rows(rng)
        want, worigo = numpy_point_cloud_to_container(c, T, 0.16, 900.0, -1.0, 1.0, 20.0)
        for o in os_:
            got, origo = o.point_cloud_to_container(c, T, 0.16, 900.0, -1.0, 1.0, 20.0)
            assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), n
            assert np.array_equal(bits(origo), bits(worigo))
    ident = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float64)
    c = synthetic_cloud(rng, 300)
    c[:, 2] = 0
    c = c[np.isfinite(c).all(1)]
    got, origo = os_[0].point_cloud_to_container(c, ident, -1.0, 1e9, -1.0, 1.0, 1.0)
    sel = ~((c[:, 0] < 0) & (c[:, 0] ** 2 + c[:, 1] ** 2 < 0.5))
    assert np.array_equal(bits(got), bits(c[sel, :2])) and not origo.any()
    r = rng.uniform(0.8, 29.0, 1081).astype(np.float32)  # beyond the x<0 && d2<0.5 rule
    a0, inc = -2.35619449, 0.00436332
    cloud = os_[0].project_laser(r, a0, inc, 0.1, 30.0, 30.0)
    assert cloud.shape == (1081, 3) and not cloud[:, 2].any()
    via_cloud, _ = os_[0].point_cloud_to_container(cloud, ident, 0.16, 900.0, -1.0, 1.0, 20.0)
    direct = os_[0].laser_scan_to_container(r, a0, inc, 0.4, 30.1, 20.0)
    assert via_cloud.shape == direct.shape and np.abs(via_cloud - direct).max() < 2e-2  # running fp32 angle drifts
    for o in os_:
        assert np.array_equal(bits(o.project_laser(r, a0, inc, 0.1, 30.0, -1.0)), bits(cloud))


NODE_GATES = [(0.4, 30.0, -1.0, 1.0), (0.25, 12.5, -0.3, 0.7)]  # laser_min_dist, laser_max_dist, laser_z_min / max_value
SCAN_GEOMS = [(-2.35619449, 0.00436332, 1081), (-1.5707964, 0.017453292, 181), (-3.1415927, 0.00038349519, 16384), (0.3, -0.01, 700)]


def test_node_source_pins_the_restatement_of_the_node_rows(oracle_mod, small_scene):
    """CPU: the reference's own HectorMappingRos member functions (compiled from the unmodified .cpp) against the restatement
    the other tests use -- LaserScan ingestion, PointCloud ingestion for two gate settings, scanCallback's projectLaser +
    conversion, publishMap's cells and setServiceGetMapData's metadata: bit for bit, NaN / inf / gate values included"""
    if not oracle_mod.available("node"):
        pytest.skip("oracle/_ref/libhector_node_ref.so not built (needs /root/reference)")
    ho = oracle_mod.Oracle("ho", 0.05, 64, 64, 1)
    rng = np.random.default_rng(21)
    for lmin, lmax, zmin, zmax in NODE_GATES:
        node = oracle_mod.NodeRef(lmin, lmax, zmin, zmax)
        assert node.sqr_min == float(np.float32(lmin * lmin)) and node.sqr_max == float(np.float32(lmax * lmax))
        for a0, inc, n in SCAN_GEOMS + [(0.0, 0.01, 0), (0.0, 0.01, 1)]:
            r = synthetic_ranges(rng, n) if n else np.zeros(0, np.float32)
            for rmin, rmax, scale in ((0.4, 30.0, 20.0), (0.1, 60.0, 40.0)):
                got, origo = node.laser_scan_to_container(r, a0, inc, rmin, rmax, scale)
                want = ho.laser_scan_to_container(r, a0, inc, rmin, rmax, scale)
                assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (n, rmin)
                assert not origo.any()  # dataContainer.setOrigo(Eigen::Vector2f::Zero()), :491
        for n in (0, 1, 10, 1081, 5000):
            c, T = synthetic_cloud(rng, n), rigid_rows(rng)
            got, go = node.point_cloud_to_container(c, T, 20.0)
            want, wo = ho.point_cloud_to_container(c, T, node.sqr_min, node.sqr_max, zmin, zmax, 20.0)
            assert got.shape == want.shape and np.array_equal(bits(got), bits(want)) and np.array_equal(bits(go), bits(wo)), n
        for a0, inc, n in SCAN_GEOMS:
            for cutoff in (30.0, -1.0, 12.5):
                r, T = synthetic_ranges(rng, n), rigid_rows(rng)
                got, go, cloud = node.project_and_convert(r, a0, inc, 0.1, 30.0, cutoff, T, 20.0)
                wc = ho.project_laser(r, a0, inc, 0.1, 30.0, cutoff)
                want, wo = ho.point_cloud_to_container(wc, T, node.sqr_min, node.sqr_max, zmin, zmax, 20.0)
                assert np.array_equal(bits(cloud), bits(wc))
                assert got.shape == want.shape and np.array_equal(bits(got), bits(want)) and np.array_equal(bits(go), bits(wo))
        node.close()
    # publishMap / setServiceGetMapData on a real map
    sc = small_scene
    o = make_oracle(oracle_mod, "ho", sc)
    lo, _ = o.download_level(0)
    node = oracle_mod.NodeRef()
    cells, meta = node.publish_map(sc.resolution, lo)
    assert np.array_equal(cells, o.occupancy_grid(0)) and set(np.unique(cells)) == {-1, 0, 100}
    w = o.world_coords_pose(0, np.zeros(3, np.float32))  # getWorldCoords(Vector2f::Zero()), then -= cellLength * 0.5f (:546-547)
    half = np.float32(o.level_info(0)[2]) * np.float32(0.5)
    assert np.float32(meta[0]) == np.float32(w[0]) - half and np.float32(meta[1]) == np.float32(w[1]) - half
    assert np.float32(meta[2]) == np.float32(o.level_info(0)[2]) and meta[3:] == (sc.map_size, sc.map_size)


def laser_scan_messages(n_scans, seed=3):
    """raw LaserScan ranges along a loop in a 20 m x 15 m room (1081 beams), as a driver would publish them"""
    from hector_slam_amd import synth
    world = synth.World.make(20.0, 15.0, seed=99)
    poses = synth.loop_trajectory(world, 4 * n_scans)[:n_scans]  # ~0.1 m between scans
    ang = synth.beam_angles(1081)
    rng = np.random.default_rng(seed)
    scans = [(world.raycast(p, ang) + rng.normal(0.0, 0.01, ang.shape)).astype(np.float32) for p in poses]
    return scans, float(ang[0]), float(np.float32(synth.SCAN_SHAPES[1081][1]))


def test_node_scan_callback_equals_the_processor_loop(oracle_mod):
    """CPU: HectorMappingRos::scanCallback itself (unmodified node source, no tf path) over 40 raw LaserScans == the
    HectorSlamProcessor loop the other tests drive (container from the restated ingestion, start = last pose): every pose,
    every covariance, the final map and its occupancy grid, bit for bit"""
    if not (oracle_mod.available("node") and oracle_mod.available("hr")):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    scans, a0, inc = laser_scan_messages(40)
    node = oracle_mod.NodeRef(map_size=512, levels=3, resolution=0.05, update_dist_thresh=0.05, update_angle_thresh=0.02)
    o = oracle_mod.Oracle("hr", 0.05, 512, 512, 3)
    o.set_update_factor_free(0.4)
    o.set_update_factor_occupied(0.9)
    o.proc_set_thresholds(0.05, 0.02)
    last = np.zeros(3, np.float32)
    for t, r in enumerate(scans):
        pn, cn = node.scan_callback(r, a0, inc, 0.4, 30.0)
        pts = o.laser_scan_to_container(r, a0, inc, 0.4, 30.0, o.scale_to_map())
        o.proc_update(pts, last)
        po, co = o.proc_last_pose()
        assert np.array_equal(bits(pn), bits(po)) and np.array_equal(bits(cn), bits(co)), t
        last = po
    cells, lo, ui = node.node_map()
    lo_o, _ = o.download_level(0)
    assert np.array_equal(bits(lo), bits(lo_o)) and np.array_equal(cells, o.occupancy_grid(0)) and (cells == 100).sum() > 200
    assert np.abs(last[:2]).max() > 0.5  # the robot did move through the map
    # the node's DEFAULT path (use_tf_scan_transformation: lookupTransform -> projectLaser -> rosPointCloudToDataContainer ->
    # update from the last pose, :260-327) with the laser 12 cm ahead of and 30 cm above base_link
    T = np.array([1, 0, 0, 0.12, 0, 1, 0, -0.05, 0, 0, 1, 0.3], np.float64)
    node = oracle_mod.NodeRef(map_size=512, levels=3, resolution=0.05, update_dist_thresh=0.05, update_angle_thresh=0.02, laser_transform=T)
    o = oracle_mod.Oracle("hr", 0.05, 512, 512, 3)
    o.set_update_factor_free(0.4)
    o.set_update_factor_occupied(0.9)
    o.proc_set_thresholds(0.05, 0.02)
    last = np.zeros(3, np.float32)
    for t, r in enumerate(scans[:25]):
        pn, cn = node.scan_callback(r, a0, inc, 0.4, 30.0)
        cloud = o.project_laser(r, a0, inc, 0.4, 30.0, 30.0)  # projector_.projectLaser(scan, cloud, 30.0), :273
        pts, origo = o.point_cloud_to_container(cloud, T, node.sqr_min, node.sqr_max, -1.0, 1.0, o.scale_to_map())
        o.proc_update(pts, last, origo=origo)
        po, co = o.proc_last_pose()
        assert np.array_equal(bits(pn), bits(po)) and np.array_equal(bits(cn), bits(co)), ("tf path", t)
        last = po


@pytest.fixture(scope="module")
def capi():
    import torch
    assert torch.cuda.is_available()
    from hector_slam_amd import capi as m
    m.load_library()
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("tf_path", [False, True], ids=["laser-scan path", "tf path (node default)"])
@pytest.mark.parametrize("parity", ["auto", "exact", "fast"])
def test_ros_node_source_runs_unchanged_on_the_mi355_map_representation(capi, oracle_mod, monkeypatch, parity, tf_path):
    """THE drop-in: hector_mapping/src/HectorMappingRos.cpp, unmodified, compiled once against the reference's include tree and
    once against the tree in which only slam_main/MapRepMultiMap.h is ours (+ libhector_mi355.so).  60 raw LaserScan messages
    through scanCallback on both: the library default (and HSM_PARITY=exact) -- every pose, every covariance, the published
    occupancy grid and the log-odds of the node's map bit-identical (round 5: the default takes the reference's summation order
    on the single-scan entry point too); HSM_PARITY=fast -- every pose within 1e-4 m / 1e-4 rad"""
    if not (oracle_mod.available("node") and oracle_mod.available("node_mi355")):
        pytest.skip("oracle/_ref/libhector_node_{ref,mi355}.so not prebuilt (run __graft_entry__.build() where /root/reference exists)")
    monkeypatch.setenv("HSM_PARITY", parity)
    scans, a0, inc = laser_scan_messages(60)
    kw = dict(map_size=512, levels=3, resolution=0.05, update_dist_thresh=0.05, update_angle_thresh=0.02)
    if tf_path:  # use_tf_scan_transformation, the node's default: laser 12 cm ahead of and 30 cm above base_link
        kw["laser_transform"] = np.array([1, 0, 0, 0.12, 0, 1, 0, -0.05, 0, 0, 1, 0.3], np.float64)
    ref = oracle_mod.NodeRef(kind="node", **kw)
    gpu = oracle_mod.NodeRef(kind="node_mi355", **kw)
    worst = 0.0
    for t, r in enumerate(scans):
        pr, cr = ref.scan_callback(r, a0, inc, 0.4, 30.0)
        pg, cg = gpu.scan_callback(r, a0, inc, 0.4, 30.0)
        d = np.abs(pr.astype(np.float64) - pg)
        worst = max(worst, float(d[:2].max()))
        assert d[0] <= 1e-4 and d[1] <= 1e-4 and d[2] <= 1e-4, (parity, t, pr, pg)
        if parity != "fast":
            assert np.array_equal(bits(pr), bits(pg)) and np.array_equal(bits(cr), bits(cg)), (parity, t)
    cells_r, lo_r, ui_r = ref.node_map()
    cells_g, lo_g, ui_g = gpu.node_map()
    assert ui_r == ui_g and (cells_r == 100).sum() > 200
    if parity != "fast":
        assert np.array_equal(cells_r, cells_g) and np.array_equal(bits(lo_r), bits(lo_g))
    else:
        assert (cells_r != cells_g).sum() <= 0.002 * (cells_r != -1).sum()
    print(f"node source on the facade, HSM_PARITY={parity}: worst pose deviation {worst:.2e} m over {len(scans)} scans")
    ref.close()
    gpu.close()


@pytest.mark.gpu
def test_gpu_node_rows_equal_the_reference_node_source(capi, oracle_mod, small_scene):
    """rows f1 / f2 against the reference's ROS node source itself (oracle/node_shim.cpp): the device's LaserScan ingestion,
    PointCloud ingestion, fused projectLaser + ingestion, occupancy export and map metadata equal what the unmodified
    HectorMappingRos member functions produce, bit for bit"""
    if not oracle_mod.available("node"):
        pytest.skip("oracle/_ref/libhector_node_ref.so not prebuilt (run __graft_entry__.build() where /root/reference exists)")
    g = capi.MapRepMultiMap(0.05, 256, 256, 2)
    s = g.getScaleToMap()
    rng = np.random.default_rng(22)
    for lmin, lmax, zmin, zmax in NODE_GATES:
        node = oracle_mod.NodeRef(lmin, lmax, zmin, zmax)
        gates = (np.float32(node.sqr_min), np.float32(node.sqr_max), zmin, zmax)
        for a0, inc, n in SCAN_GEOMS:
            r = synthetic_ranges(rng, n)
            want, _ = node.laser_scan_to_container(r, a0, inc, 0.4, 30.0, s)
            got = g.ingest_laser_scan(r, a0, inc, 0.4, 30.0)
            assert got.shape == want.shape and np.array_equal(bits(got), bits(want)), (n, "laser scan")
            T = rigid_rows(rng)
            for cutoff in (30.0, -1.0):
                want, wo, _cloud = node.project_and_convert(r, a0, inc, 0.1, 30.0, cutoff, T, s)
                got, go = g.ingest_laser_scan_tf(r, a0, inc, 0.1, 30.0, cutoff, T, *gates)
                assert got.shape == want.shape and np.array_equal(bits(got), bits(want)) and np.array_equal(bits(go), bits(wo)), (n, cutoff)
        for n in (1, 10, 1081, 5000):
            c, T = synthetic_cloud(rng, n), rigid_rows(rng)
            want, wo = node.point_cloud_to_container(c, T, s)
            got, go = g.ingest_point_cloud(c, T, *gates)
            assert got.shape == want.shape and np.array_equal(bits(got), bits(want)) and np.array_equal(bits(go), bits(wo)), n
        node.close()
    # occupancy export + metadata of a map the device built
    sc = small_scene
    m = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    m.setUpdateFactorFree(0.4)
    m.setUpdateFactorOccupied(0.9)
    m.build_map(sc.build_poses, sc.build_scans)
    lo, _ = m.download_level(0)
    node = oracle_mod.NodeRef()
    cells, meta = node.publish_map(sc.resolution, lo)
    assert np.array_equal(m.occupancy_grid(0), cells) and (cells == 100).sum() > 50
    ox, oy, res = m.map_metadata(0)
    assert (np.float32(ox), np.float32(oy), np.float32(res)) == (np.float32(meta[0]), np.float32(meta[1]), np.float32(meta[2]))


@pytest.mark.gpu
def test_ingest_laser_scan_bit_exact(capi, oracle_mod, pyramid_scene):
    sc = pyramid_scene
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    o = make_oracle(oracle_mod, "ho", sc, build=False)
    rng = np.random.default_rng(2)
    s = g.getScaleToMap()
    geoms = [(-2.35619449, 0.00436332, 1081), (-1.5707964, 0.017453292, 181), (-3.1415927, 0.00038349519, 16384),
             (0.3, -0.01, 700), (-2.35619449, 0.00436332, 1081)]
    for a0, inc, n in geoms:
        for trial in range(3):
            r = synthetic_ranges(rng, n)
            got = g.ingest_laser_scan(r, a0, inc, 0.4, 30.0)
            ref = o.laser_scan_to_container(r, a0, inc, 0.4, 30.0, s)
            assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), (a0, inc, n, trial)
    # edge cases: empty scan, nothing valid, everything valid, sizes around the 1024-beam chunk
    assert g.ingest_laser_scan(np.zeros(0, np.float32), 0.0, 0.1, 0.4, 30.0).shape == (0, 2)
    assert g.ingest_laser_scan(np.full(500, np.inf, np.float32), 0.0, 0.01, 0.4, 30.0).shape == (0, 2)
    for n in (1, 63, 64, 65, 1023, 1024, 1025, 2049):
        r = rng.uniform(1.0, 20.0, n).astype(np.float32)
        got = g.ingest_laser_scan(r, -1.0, 0.002, 0.4, 30.0)
        ref = o.laser_scan_to_container(r, -1.0, 0.002, 0.4, 30.0, s)
        assert got.shape[0] == n and np.array_equal(bits(got), bits(ref)), n


@pytest.mark.gpu
def test_ingested_scan_drives_match_and_update_identically(capi, oracle_mod, pyramid_scene):
    """ranges -> device container -> matchData / updateByScan == the same calls fed with host endpoints"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    a = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    b = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    for m in (a, b):
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
    ang = synth.beam_angles(1081)
    a0, inc = float(ang[0]), float(np.float32(synth.SCAN_SHAPES[1081][1]))
    rng = np.random.default_rng(3)
    for t in range(12):
        r = sc.world.raycast(sc.build_poses[t], ang).astype(np.float32)
        r = (r + rng.normal(0, 0.01, r.shape)).astype(np.float32)
        pts = a.ingest_laser_scan(r, a0, inc, 0.4, 30.0)
        assert pts.shape[0] > 900
        pa, ca = a.match_ingested(sc.build_poses[t])
        pb, cb = b.matchData(sc.build_poses[t], pts)
        assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(ca), bits(cb))
        a.update_by_ingested(sc.build_poses[t])
        b.updateByScan(pts, sc.build_poses[t])
    for lvl in range(sc.levels):
        la, lb = a.download_level(lvl), b.download_level(lvl)
        assert (la[0] != 0).sum() > 500
        assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1])


@pytest.mark.gpu
def test_ingest_point_cloud_bit_exact(capi, oracle_mod, pyramid_scene):
    """rosPointCloudToDataContainer (+ fused projectLaser) on the device == the restatement, bit for bit"""
    sc = pyramid_scene
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    o = make_oracle(oracle_mod, "ho", sc, build=False)
    rng = np.random.default_rng(21)
    s = g.getScaleToMap()
    gates = (np.float32(0.4 * 0.4), np.float32(30.0 * 30.0), -1.0, 1.0)  # the node's defaults (:98-108)
    for n in (0, 1, 63, 64, 65, 1023, 1024, 1025, 1081, 16384):
        for trial in range(2):
            c, T = synthetic_cloud(rng, n), rigid_rows(rng)
            got, go = g.ingest_point_cloud(c, T, *gates)
            ref, ro = o.point_cloud_to_container(c, T, *gates, s)
            assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), (n, trial)
            assert np.array_equal(bits(go), bits(ro))
            if n >= 1000:
                assert 0.3 * n < ref.shape[0] < n
    for a0, inc, n in [(-2.35619449, 0.00436332, 1081), (-3.1415927, 0.00038349519, 16384), (0.3, -0.01, 700)]:
        for cutoff in (30.0, -1.0, 12.5):
            r, T = synthetic_ranges(rng, n), rigid_rows(rng)
            got, go = g.ingest_laser_scan_tf(r, a0, inc, 0.1, 30.0, cutoff, T, *gates)
            cloud = o.project_laser(r, a0, inc, 0.1, 30.0, cutoff)
            ref, ro = o.point_cloud_to_container(cloud, T, *gates, s)
            assert got.shape == ref.shape and np.array_equal(bits(got), bits(ref)), (n, cutoff)
            assert np.array_equal(bits(go), bits(ro)) and ref.shape[0] > n // 4
    # argument validation: NULL transform / negative count / too many points are refused, nothing is ingested
    import ctypes as C
    lib = capi.load_library()
    cnt = C.c_int(-7)
    c3 = synthetic_cloud(rng, 10)
    assert lib.hsm_ingest_point_cloud(g._h, c3.ctypes.data, 10, None, 0.16, 900.0, -1.0, 1.0, 20.0, None,
                                      C.byref(cnt), None) == -1 and b"bad argument" in lib.hsm_last_error()
    assert lib.hsm_ingest_point_cloud(g._h, c3.ctypes.data, -1, rigid_rows(rng).ctypes.data, 0.16, 900.0, -1.0, 1.0,
                                      20.0, None, C.byref(cnt), None) == -1
    assert lib.hsm_ingest_laser_scan_tf(g._h, None, 5, 0.0, 0.1, 0.1, 30.0, 30.0, rigid_rows(rng).ctypes.data, 0.16,
                                        900.0, -1.0, 1.0, 20.0, None, C.byref(cnt), None) == -1
    assert lib.hsm_ingest_point_cloud(g._h, c3.ctypes.data, 1_048_576, rigid_rows(rng).ctypes.data, 0.16, 900.0, -1.0,
                                      1.0, 20.0, None, C.byref(cnt), None) == -1 and cnt.value == -7
    # alternating entries share the geometry-table cache: the float2 and double2 tables must not be confused
    r = synthetic_ranges(rng, 1081)
    T = rigid_rows(rng)
    for _ in range(2):
        a = g.ingest_laser_scan(r, -2.35619449, 0.00436332, 0.4, 30.0)
        assert np.array_equal(bits(a), bits(o.laser_scan_to_container(r, -2.35619449, 0.00436332, 0.4, 30.0, s)))
        b, _o = g.ingest_laser_scan_tf(r, -2.35619449, 0.00436332, 0.1, 30.0, 30.0, T, *gates)
        cloud = o.project_laser(r, -2.35619449, 0.00436332, 0.1, 30.0, 30.0)
        assert np.array_equal(bits(b), bits(o.point_cloud_to_container(cloud, T, *gates, s)[0]))


@pytest.mark.gpu
def test_ingested_cloud_keeps_its_origo_through_match_and_update(capi, oracle_mod, pyramid_scene):
    """tf path: the container's origo is the laser position (:517); the device-resident container updates the
    map exactly like updateByScan fed with the host endpoints + that origo, and like the oracle"""
    from hector_slam_amd import synth
    sc = pyramid_scene
    a = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    b = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    o = oracle_mod.Oracle("ho", sc.resolution, sc.map_size, sc.map_size, sc.levels)
    for m in (a, b, o):
        (m.setUpdateFactorFree if hasattr(m, "setUpdateFactorFree") else m.set_update_factor_free)(0.4)
        (m.setUpdateFactorOccupied if hasattr(m, "setUpdateFactorOccupied") else m.set_update_factor_occupied)(0.9)
    ang = synth.beam_angles(1081)
    a0, inc = float(ang[0]), float(np.float32(synth.SCAN_SHAPES[1081][1]))
    T = np.array([1, 0, 0, 0.12, 0, 1, 0, -0.05, 0, 0, 1, 0.3], np.float64)  # laser 12 cm ahead of base_link
    gates = (np.float32(0.16), np.float32(900.0), -1.0, 1.0)
    rng = np.random.default_rng(5)
    for t in range(10):
        r = sc.world.raycast(sc.build_poses[t], ang).astype(np.float32)
        r = (r + rng.normal(0, 0.01, r.shape)).astype(np.float32)
        pts, origo = a.ingest_laser_scan_tf(r, a0, inc, 0.1, 30.0, 30.0, T, *gates)
        assert pts.shape[0] > 900 and origo[0] != 0
        pa, ca = a.match_ingested(sc.build_poses[t])
        pb, cb = b.matchData(sc.build_poses[t], pts, origo=origo)
        assert np.array_equal(bits(pa), bits(pb)) and np.array_equal(bits(ca), bits(cb))
        o.match(sc.build_poses[t], pts, origo)  # retains the coarse containers like the reference
        a.update_by_ingested(sc.build_poses[t])
        b.updateByScan(pts, sc.build_poses[t], origo=origo)
        o.update_by_scan(sc.build_poses[t], pts, origo)
        o.on_map_updated()
    for lvl in range(sc.levels):
        la, lb, lo = a.download_level(lvl), b.download_level(lvl), o.download_level(lvl)
        assert (la[0] != 0).sum() > 500
        assert np.array_equal(bits(la[0]), bits(lb[0])) and np.array_equal(la[1], lb[1])
        assert np.array_equal(bits(la[0]), bits(lo[0])) and np.array_equal(la[1], lo[1])


@pytest.mark.gpu
def test_occupancy_grid_bit_exact(capi, oracle_mod, pyramid_scene, small_scene):
    for sc in (pyramid_scene, small_scene):
        g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
        g.setUpdateFactorFree(0.4)
        g.setUpdateFactorOccupied(0.9)
        g.build_map(sc.build_poses[:30], sc.build_scans[:30])
        o = make_oracle(oracle_mod, "ho", sc, build=False)
        o.build_map(sc.build_poses[:30], sc.build_scans[:30])
        for lvl in range(sc.levels):
            got, ref = g.occupancy_grid(lvl), o.occupancy_grid(lvl)
            assert np.array_equal(got, ref)
            assert (ref == 100).sum() > 50 and (ref == 0).sum() > 1000 and (ref == -1).sum() > 1000
    # odd-sized level (cells not a multiple of 4) and a fresh map (all unknown)
    g = capi.MapRepMultiMap(0.1, 250, 250, 2)
    assert (g.occupancy_grid(1) == -1).all() and g.occupancy_grid(1).shape == (125, 125)
    lo = np.random.default_rng(4).normal(0, 1, (125, 125)).astype(np.float32)
    lo[::7, ::3] = 0.0
    g.upload_level(1, lo)
    exp = np.where(lo < 0, 0, np.where(lo > 0, 100, -1)).astype(np.int8)
    assert np.array_equal(g.occupancy_grid(1), exp)


@pytest.mark.gpu
def test_likelihood_states_matches_reference(capi, oracle_mod, pyramid_scene):
    """f3: getLikelihoodForState for 7-sigma-point style hypothesis sets (getCovarianceForPose's pattern,
    OccGridMapUtil.h:106-160) and a 4096-particle cloud; per-beam M is bit-exact, the residual sum differs
    only in summation order (tree vs the reference's 1081-term sequential fp32 chain, whose own rounding is ~sqrt(n)*eps): |dlh| <= 1e-5"""
    sc = pyramid_scene
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    kinds = ["ho"] + (["hr"] if oracle_mod.available("hr") else [])
    orc = {k: make_oracle(oracle_mod, k, sc) for k in kinds}
    rng = np.random.default_rng(9)
    for lvl in range(sc.levels):
        f = np.float32(1.0 / 2 ** lvl)
        for q in range(4):
            pm = orc["ho"].map_coords_pose(lvl, sc.query_truth[q])
            x, y, a = (float(v) for v in pm)
            sig = np.array([[x + 1.5, y, a], [x - 1.5, y, a], [x, y + 1.5, a], [x, y - 1.5, a], [x, y, a + 0.05],
                            [x, y, a - 0.05], [x, y, a]], np.float32)
            cloud = (pm[None, :] + rng.normal(0, [2.0, 2.0, 0.05], (4096, 3))).astype(np.float32)
            far = np.array([[-50.0, 3.0, 0.1], [1e6, 1e6, 0.0]], np.float32)  # (almost) everything out of map
            states = np.concatenate([sig, cloud, far])
            got = g.likelihood_states(lvl, states, sc.query_scans[q])
            for k in kinds:
                ref = orc[k].likelihood_states(lvl, states, sc.query_scans[q] * f)
                assert np.abs(got - ref).max() <= 1e-5, (lvl, q, k, np.abs(got - ref).max())
            assert got[6] > got[:6].max() - 1e-3 and got[-1] == 0.0  # truth is (near) the best sigma point
    assert g.likelihood_states(0, np.zeros((0, 3), np.float32), sc.query_scans[0]).shape == (0,)


def sigma_statistics_f32(pose, lh):
    """the 7-term statistics of getCovarianceForPose (:138-154) in scalar fp32, source order"""
    F = np.float32
    x, y, a = (F(v) for v in pose)
    sp = [[x + F(1.5), y, a], [x - F(1.5), y, a], [x, y + F(1.5), a], [x, y - F(1.5), a], [x, y, a + F(0.05)],
          [x, y, a - F(0.05)], [x, y, a]]
    lh = [F(v) for v in lh]
    inv = F(1) / ((lh[0] + (lh[1] + lh[2])) + ((lh[3] + lh[4]) + (lh[5] + lh[6])))
    mean = [F(0)] * 3
    for i in range(7):
        for r in range(3):
            mean[r] = mean[r] + sp[i][r] * lh[i]
    mean = [m * inv for m in mean]
    cov = np.zeros(9, np.float32)
    for i in range(7):
        d = [sp[i][r] - mean[r] for r in range(3)]
        w = lh[i] * inv
        for c in range(3):
            for r in range(3):
                cov[c * 3 + r] = cov[c * 3 + r] + w * (d[r] * d[c])
    return cov


def test_oracle_pose_covariance_restatement_vs_reference(oracle_mod, small_scene):
    """CPU: the restatement of getResidualForState / getCovarianceForPose / getCovMatrixWorldCoords against the
    reference's own functions (hr): likelihoods and residuals bit-identical, covariances to 2 ulp-ish (Eigen
    may associate scalar * (d d^T) either way), symmetric, positive semi-definite, world = map scaled"""
    if not oracle_mod.available("hr"):
        pytest.skip("oracle/_ref not built")
    sc = small_scene
    a, b = make_oracle(oracle_mod, "ho", sc), make_oracle(oracle_mod, "hr", sc)
    poses = np.stack([a.map_coords_pose(0, sc.query_truth[q]) for q in range(6)]).astype(np.float32)
    poses[3:] += np.float32([0.7, -0.4, 0.02])
    for q in range(3):
        ra, rb = (o.residual_states(0, poses, sc.query_scans[q]) for o in (a, b))
        assert np.array_equal(bits(ra), bits(rb)) and (ra > 0).all()
        (ma, wa, la), (mb, wb, lb) = (o.covariance_for_poses(0, poses, sc.query_scans[q]) for o in (a, b))
        assert np.array_equal(bits(la), bits(lb))
        assert np.allclose(ma, mb, rtol=1e-5, atol=1e-7) and np.allclose(wa, wb, rtol=1e-5, atol=1e-9)
        for i in range(poses.shape[0]):
            assert np.array_equal(bits(ma[i]), bits(sigma_statistics_f32(poses[i], la[i])))
            M = ma[i].reshape(3, 3)
            assert np.array_equal(M, M.T) and np.linalg.eigvalsh(M.astype(np.float64)).min() > -1e-6
            assert 0.3 < M[0, 0] < 2.25 and 0.3 < M[1, 1] < 2.25 and 1e-4 < M[2, 2] < 0.0025
        c = np.float32(sc.resolution)
        assert np.array_equal(bits(wa[:, 0]), bits(ma[:, 0] * (c * c))) and np.array_equal(bits(wa[:, 8]), bits(ma[:, 8]))
        assert np.array_equal(bits(wa[:, 2]), bits(ma[:, 2] * c)) and np.array_equal(bits(wa[:, 6]), bits(wa[:, 2]))


@pytest.mark.gpu
def test_pose_covariance_and_residual_match_reference(capi, oracle_mod, pyramid_scene):
    """f3: getResidualForState and getCovarianceForPose (+ world scaling) for batches of poses on every level.
    The 7 likelihoods differ from the reference only by summation order (<= 1e-5); the 7-term statistics on
    top of them are bit-exact (checked by feeding the device's own likelihoods to a scalar fp32 restatement),
    so the covariances agree to the propagated 1e-5."""
    sc = pyramid_scene
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    kinds = ["ho"] + (["hr"] if oracle_mod.available("hr") else [])
    orc = {k: make_oracle(oracle_mod, k, sc) for k in kinds}
    rng = np.random.default_rng(19)
    for lvl in range(sc.levels):
        f = np.float32(1.0 / 2 ** lvl)
        cell = np.float32(g.level_info(lvl)[2])
        for q in range(3):
            pm = orc["ho"].map_coords_pose(lvl, sc.query_truth[q])
            poses = (pm[None, :] + rng.normal(0, [1.0, 1.0, 0.03], (33, 3))).astype(np.float32)
            poses[0] = pm
            cm, cw, lh = g.covariance_for_poses(lvl, poses, sc.query_scans[q])
            res = g.residual_states(lvl, poses, sc.query_scans[q])
            n = sc.query_scans[q].shape[0]
            for i in range(poses.shape[0]):
                assert np.array_equal(bits(cm[i]), bits(sigma_statistics_f32(poses[i], lh[i]))), (lvl, q, i)
            assert np.array_equal(bits(cw[:, 0]), bits(cm[:, 0] * (cell * cell)))
            assert np.array_equal(bits(cw[:, 5]), bits(cm[:, 5] * cell)) and np.array_equal(bits(cw[:, 7]), bits(cw[:, 5]))
            assert np.array_equal(bits(cw[:, 3]), bits(cw[:, 1])) and np.array_equal(bits(cw[:, 8]), bits(cm[:, 8]))
            assert np.array_equal(bits(lh[:, 6]), bits(g.likelihood_states(lvl, poses, sc.query_scans[q])))
            for k in kinds:
                rm, rw, rl = orc[k].covariance_for_poses(lvl, poses, sc.query_scans[q] * f)
                rr = orc[k].residual_states(lvl, poses, sc.query_scans[q] * f)
                assert np.abs(lh - rl).max() <= 1e-5, (lvl, q, k)
                assert np.abs(res - rr).max() <= 1e-5 * n, (lvl, q, k, np.abs(res - rr).max())
                assert np.abs(cm - rm).max() <= 5e-5 and np.abs(cw - rw).max() <= 5e-5, (lvl, q, k, np.abs(cm - rm).max())
    cm, cw, lh = g.covariance_for_poses(0, np.zeros((0, 3), np.float32), sc.query_scans[0])
    assert cm.shape == (0, 9)
    with pytest.raises(capi.HsmError):
        g.covariance_for_poses(sc.levels, np.zeros((1, 3), np.float32), sc.query_scans[0])


@pytest.mark.gpu
def test_ray_distances_bit_exact(capi, oracle_mod, pyramid_scene):
    """f4: hector_map_tools' getDist on 20k random rays per level: distances and hit coordinates bit-exact
    against the restatement run on the oracle's occupancy grid (integer Bresenham + one fp32 sqrt)"""
    sc = pyramid_scene
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
    g.setUpdateFactorFree(0.4)
    g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    o = make_oracle(oracle_mod, "ho", sc)
    rng = np.random.default_rng(12)
    kinds = ["ho"] + (["hr"] if oracle_mod.available("hr") else [])
    for lvl in range(sc.levels):
        ox, oy, res = g.map_metadata(lvl)
        grid = o.occupancy_grid(lvl)
        n = 20000
        begin = rng.uniform(-9, 9, (n, 2)).astype(np.float32)
        ang = rng.uniform(0, 2 * np.pi, n)
        length = rng.uniform(0.0, 25.0, n)
        end = (begin + np.stack([np.cos(ang), np.sin(ang)], 1) * length[:, None]).astype(np.float32)
        end[:50] = begin[:50]                      # zero-length rays
        begin[50:100] += 100.0                     # begin outside the map
        end[100:150] += 100.0                      # end outside the map
        end[150:200, 1] = begin[150:200, 1]        # axis-aligned
        end[200:250, 0] = begin[200:250, 0]
        dist, hit = g.ray_distances(lvl, begin, end)
        for k in kinds:
            rd, rh = oracle_mod.ray_distances(k, grid, (ox, oy), res, begin, end)
            assert np.array_equal(bits(dist), bits(rd)), (lvl, k, (bits(dist) != bits(rd)).sum())
            has = rd >= 0
            assert np.array_equal(bits(hit[has]), bits(rh[has])) and np.isnan(hit[~has]).all()
        assert 0.2 < (dist >= 0).mean() < 0.95 and (dist[:150] < 0).sum() >= 100
    d0, h0 = g.ray_distances(0, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))
    assert d0.shape == (0,)

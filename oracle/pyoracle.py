"""TEST INFRASTRUCTURE ONLY -- ctypes front-end for the two CPU checkers.

``Oracle("ho")`` drives oracle/build/libhector_oracle.so (plain-C++ restatement),
``Oracle("hr")`` drives oracle/_ref/libhector_ref.so (the unmodified reference
headers compiled through the private Eigen/tf stand-in).  Same API, see
oracle/oracle_api.h.  ``NodeRef`` drives oracle/_ref/libhector_node_ref.so: the reference's ROS
node source (hector_mapping/src/HectorMappingRos.cpp) compiled unmodified through
oracle/stubs_node/ -- the checker of the node-side rows.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SAN = os.environ.get("HSM_ORACLE_SAN") == "1"  # tools/sanitize_cpu.sh: the ASan/UBSan builds (make -C oracle SAN=1)
_LIBS = {"ho": os.path.join(_HERE, "build", *(["san"] if _SAN else []), "libhector_oracle.so"),
         "hr": os.path.join(_HERE, "_ref", *(["san"] if _SAN else []), "libhector_ref.so"),
         # the UNMODIFIED ROS node source (hector_mapping/src/HectorMappingRos.cpp) through oracle/stubs_node/: rows f1 / f2
         "node": os.path.join(_HERE, "_ref", "libhector_node_ref.so"),
         # ... and the same unmodified node source on the drop-in facade + libhector_mi355.so (needs a GPU to construct)
         "node_mi355": os.path.join(_HERE, "_ref", "libhector_node_mi355.so")}
_loaded: dict = {}

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(quiet: bool = True) -> None:
    """make -C oracle: restatement always; _ref only where /root/reference exists."""
    subprocess.run(["make", "-C", _HERE] + (["SAN=1"] if _SAN else []), check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def available(kind: str) -> bool:
    return os.path.exists(_LIBS[kind])


def host_libm_is_fma_variant() -> bool:
    """glibc >= 2.28 dispatches sincosf / expf to its *_fma ifunc variants on x86-64 CPUs with FMA + AVX2; those are the
    variants the product's libm_exact.h restates operation for operation.  On any other host the checker's libm may differ
    from the model in the last bit, and every "bit-exact against the reference" statement would silently compare against
    another function."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                f = line.split()
                return "fma" in f and "avx2" in f
    except OSError:
        pass
    return False


def _load(kind: str):
    if kind in _loaded:
        return _loaded[kind]
    # the checker is only a checker of the bit-exact claims on a host whose libm is the one the model restates
    assert host_libm_is_fma_variant(), ("oracle host without FMA + AVX2: glibc evaluates sincosf / expf unfused here, the "
                                        "bit-exact parity of the device's libm (csrc/libm_exact.h) is UNPINNED on this host")
    if not os.path.exists(_LIBS[kind]):
        if kind == "ho":
            build()
        else:
            raise FileNotFoundError(f"{_LIBS[kind]} missing (built only where /root/reference exists)")
    lib = C.CDLL(_LIBS[kind])
    p = kind + "_"
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    sig = {
        "create": (vp, [f, i, i, C.c_uint, f, f]),
        "destroy": (None, [vp]), "reset": (None, [vp]), "levels": (i, [vp]),
        "scale_to_map": (f, [vp]),
        "set_update_factor_free": (None, [vp, f]), "set_update_factor_occupied": (None, [vp, f]),
        "level_info": (None, [vp, i, C.POINTER(i), C.POINTER(i), C.POINTER(f), C.POINTER(f)]),
        "download_level": (None, [vp, i, _f32p, _i32p]),
        "upload_level": (None, [vp, i, _f32p, _i32p]),
        "map_coords_pose": (None, [vp, i, _f32p, _f32p]),
        "world_coords_pose": (None, [vp, i, _f32p, _f32p]),
        "interp": (None, [vp, i, _f32p, i, _f32p]),
        "hessian_derivs": (None, [vp, i, _f32p, _f32p, i, _f32p, _f32p]),
        "match_level": (None, [vp, i, _f32p, _f32p, i, i, _f32p, _f32p]),
        "match": (None, [vp, _f32p, _f32p, i, _f32p, _f32p, _f32p]),
        "match_many": (None, [vp, i, _f32p, _f32p, _i32p, _f32p]),
        "update_by_scan": (None, [vp, _f32p, _f32p, i, _f32p]),
        "update_by_scan_level": (None, [vp, i, _f32p, _f32p, i, _f32p]),
        "on_map_updated": (None, [vp]),
        "undefined_reads": (C.c_long, [vp]),
        "proc_set_thresholds": (None, [vp, f, f]),
        "proc_update": (None, [vp, _f32p, i, _f32p, _f32p, i]),
        "proc_last_pose": (None, [vp, _f32p, _f32p]),
        "likelihood_states": (None, [vp, i, i, _f32p, _f32p, i, _f32p]),
        "residual_states": (None, [vp, i, i, _f32p, _f32p, i, _f32p]),
        "covariance_for_poses": (None, [vp, i, i, _f32p, _f32p, i, _f32p, _f32p, _f32p]),
        "ray_distances": (None, [np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS"), i, i, f, f, f, i, _f32p, _f32p,
                                 _f32p, _f32p]),
        "occupancy_grid": (None, [vp, i, np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")]),
        "laser_scan_to_container": (i, [_f32p, i, f, f, f, f, f, _f32p]),
        "point_cloud_to_container": (i, [_f32p, i, np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"),
                                         f, f, f, f, f, _f32p, _f32p]),
        "project_laser": (i, [_f32p, i, f, f, f, f, C.c_double, _f32p]),
        "normalize_angle": (f, [f]),
        "libm_sincosf": (None, [i, _f32p, _f32p, _f32p]),
        "libm_expf": (None, [i, _f32p, _f32p, _f32p]),
        "pose_difference_larger_than": (i, [_f32p, _f32p, f, f]),
    }
    ns = {}
    for name, (res, args) in sig.items():
        fn = getattr(lib, p + name)
        fn.restype, fn.argtypes = res, args
        ns[name] = fn
    _loaded[kind] = ns
    return ns


def _pts(pts) -> np.ndarray:
    a = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    return a if a.size else np.zeros((1, 2), np.float32)[:0].reshape(0, 2).copy()


def _v(x, n) -> np.ndarray:
    a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    assert a.size == n
    return a


_ZERO2 = np.zeros(2, np.float32)


def ray_distances(kind, grid, origin_xy, resolution, begin_world, end_world):
    """DistanceMeasurementProvider::getDist over a batch of rays -> (dist[n], hit[n,2] (NaN where no hit))"""
    f = _load(kind)
    g = np.ascontiguousarray(grid, np.int8)
    b = np.ascontiguousarray(begin_world, np.float32).reshape(-1, 2)
    e = np.ascontiguousarray(end_world, np.float32).reshape(-1, 2)
    dist = np.empty(b.shape[0], np.float32)
    hit = np.full((b.shape[0], 2), np.nan, np.float32)
    f["ray_distances"](g, g.shape[1], g.shape[0], origin_xy[0], origin_xy[1], resolution, b.shape[0],
                       b.reshape(-1) if b.size else np.zeros(2, np.float32), e.reshape(-1) if e.size else
                       np.zeros(2, np.float32), dist, hit.reshape(-1) if b.size else np.zeros(2, np.float32))
    return dist, hit


def libm_sincosf(x, kind="ho"):
    """host sinf/cosf (glibc, as the reference links them) of an array"""
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    s, c = np.empty_like(x), np.empty_like(x)
    _load(kind)["libm_sincosf"](x.size, x, s, c)
    return s, c


def libm_expf(x, kind="ho"):
    """host expf and getGridProbability of an array"""
    x = np.ascontiguousarray(x, np.float32).reshape(-1)
    e, p = np.empty_like(x), np.empty_like(x)
    _load(kind)["libm_expf"](x.size, x, e, p)
    return e, p


class NodeRef:
    """The reference's ROS node source compiled unmodified (oracle/node_shim.cpp): its own rosLaserScanToDataContainer,
    rosPointCloudToDataContainer, the projectLaser + conversion step of scanCallback, publishMap and setServiceGetMapData.
    The node reads its gates from parameters: laser_min_dist / laser_max_dist (squared in double, narrowed to float, as
    HectorMappingRos.cpp:95-100 does), laser_z_min_value / laser_z_max_value."""

    def __init__(self, laser_min_dist=0.4, laser_max_dist=30.0, laser_z_min=-1.0, laser_z_max=1.0, map_size=0, levels=1,
                 resolution=0.05, update_dist_thresh=0.4, update_angle_thresh=0.9, factor_free=0.4, factor_occ=0.9, kind="node",
                 laser_transform=None):
        """map_size = 0: a handle for the container conversions / publish_map only; map_size > 0: a whole node whose
        scan_callback runs rosLaserScanToDataContainer + HectorSlamProcessor::update on its own map.  kind "node" = on the
        reference's CPU map representation, "node_mi355" = the same node source on the drop-in facade (GPU).
        laser_transform (12 doubles [R | t], base_link <- laser): the node is created with use_tf_scan_transformation (its
        default) and scan_callback takes the tf path: lookupTransform, projectLaser, rosPointCloudToDataContainer."""
        lib = C.CDLL(_LIBS[kind])
        vp, i, f, d = C.c_void_p, C.c_int, C.c_float, C.c_double
        f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        i8p = np.ctypeslib.ndpointer(dtype=np.int8, flags="C_CONTIGUOUS")
        self.map_size = map_size
        sig = {"hn_create": (vp, [d, d, d, d, i, i, d, d, d, d, d]), "hn_destroy": (None, [vp]),
               "hn_scan_callback": (None, [vp, _f32p, i, f, f, f, f, _f32p, _f32p]),
               "hn_set_laser_transform": (None, [C.c_void_p]),
               "hn_node_map": (i, [vp, i8p, _f32p]),
               "hn_laser_scan_to_container": (i, [vp, _f32p, i, f, f, f, f, f, _f32p, _f32p]),
               "hn_point_cloud_to_container": (i, [vp, _f32p, i, f64p, f, _f32p, _f32p]),
               "hn_project_and_convert": (i, [vp, _f32p, i, f, f, f, f, d, f64p, f, _f32p, _f32p, _f32p, C.POINTER(i)]),
               "hn_publish_map": (None, [vp, f, i, i, f, f, _f32p, i8p, f64p])}
        self.f = {}
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
            self.f[name] = fn
        T = None if laser_transform is None else np.ascontiguousarray(laser_transform, np.float64).reshape(12)
        self.f["hn_set_laser_transform"](None if T is None else T.ctypes.data)  # read by the constructor, per library
        self.h = self.f["hn_create"](laser_min_dist, laser_max_dist, laser_z_min, laser_z_max, map_size, levels, resolution,
                                     update_dist_thresh, update_angle_thresh, factor_free, factor_occ)
        # what the node makes of its two distance parameters (HectorMappingRos.cpp:96,99): the gates the other checkers take
        self.sqr_min = float(np.float32(laser_min_dist * laser_min_dist))
        self.sqr_max = float(np.float32(laser_max_dist * laser_max_dist))

    def close(self):
        if getattr(self, "h", None):
            self.f["hn_destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def scan_callback(self, ranges, angle_min, angle_increment, range_min, range_max):
        """HectorMappingRos::scanCallback on one LaserScan -> (pose [3], covariance [9] column major) of the processor"""
        r = np.ascontiguousarray(ranges, np.float32).reshape(-1)
        pose, cov = np.empty(3, np.float32), np.empty(9, np.float32)
        self.f["hn_scan_callback"](self.h, r if r.size else np.zeros(1, np.float32), r.size, angle_min, angle_increment, range_min,
                                   range_max, pose, cov)
        return pose, cov

    def node_map(self):
        """publishMap of the node's own level-0 map -> (int8 cells [S,S], log-odds [S,S], update index)"""
        S = self.map_size
        cells, lo = np.empty((S, S), np.int8), np.empty((S, S), np.float32)
        ui = self.f["hn_node_map"](self.h, cells.reshape(-1), lo.reshape(-1))
        return cells, lo, ui

    def laser_scan_to_container(self, ranges, angle_min, angle_increment, range_min, range_max, scale_to_map):
        r = np.ascontiguousarray(ranges, np.float32).reshape(-1)
        out = np.empty(2 * max(r.size, 1), np.float32)
        origo = np.empty(2, np.float32)
        m = self.f["hn_laser_scan_to_container"](self.h, r if r.size else np.zeros(1, np.float32), r.size, angle_min,
                                                 angle_increment, range_min, range_max, scale_to_map, out, origo)
        return out[:2 * m].reshape(m, 2).copy(), origo

    def point_cloud_to_container(self, pts_xyz, tf_rows, scale_to_map):
        p = np.ascontiguousarray(pts_xyz, np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(tf_rows, np.float64).reshape(12)
        out = np.empty(2 * max(p.shape[0], 1), np.float32)
        origo = np.empty(2, np.float32)
        m = self.f["hn_point_cloud_to_container"](self.h, p.reshape(-1) if p.size else np.zeros(3, np.float32), p.shape[0], T,
                                                  scale_to_map, out, origo)
        return out[:2 * m].reshape(m, 2).copy(), origo

    def project_and_convert(self, ranges, angle_min, angle_increment, range_min, range_max, range_cutoff, tf_rows, scale_to_map):
        """scanCallback's default ingestion: projectLaser (laser_geometry: restated stand-in) + rosPointCloudToDataContainer
        -> (endpoints [m,2], origo [2], projected cloud [k,3])"""
        r = np.ascontiguousarray(ranges, np.float32).reshape(-1)
        T = np.ascontiguousarray(tf_rows, np.float64).reshape(12)
        out = np.empty(2 * max(r.size, 1), np.float32)
        cloud = np.empty(3 * max(r.size, 1), np.float32)
        origo = np.empty(2, np.float32)
        k = C.c_int()
        m = self.f["hn_project_and_convert"](self.h, r if r.size else np.zeros(1, np.float32), r.size, angle_min, angle_increment,
                                             range_min, range_max, range_cutoff, T, scale_to_map, out, origo, cloud, C.byref(k))
        return out[:2 * m].reshape(m, 2).copy(), origo, cloud[:3 * k.value].reshape(k.value, 3).copy()

    def publish_map(self, resolution, logodds, start=(0.5, 0.5)):
        """publishMap's cells + setServiceGetMapData's metadata for a level-0 grid with these log-odds
        -> (int8 [sy,sx], (origin_x, origin_y, resolution, width, height))"""
        lo = np.ascontiguousarray(logodds, np.float32)
        sy, sx = lo.shape
        out = np.empty((sy, sx), np.int8)
        meta = np.empty(5, np.float64)
        self.f["hn_publish_map"](self.h, resolution, sx, sy, start[0], start[1], lo.reshape(-1), out.reshape(-1), meta)
        return out, tuple(float(v) for v in meta)


class Oracle:
    """One HectorSlamProcessor-equivalent on the CPU (kind 'ho' = restatement, 'hr' = reference)."""

    def __init__(self, kind: str, resolution: float, size_x: int, size_y: int, levels: int,
                 start=(0.5, 0.5)):
        self.kind = kind
        self.f = _load(kind)
        self.h = self.f["create"](resolution, size_x, size_y, levels, start[0], start[1])
        self.n_levels = levels

    def close(self):
        if self.h:
            self.f["destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- map representation interface -------------------------------------
    def reset(self): self.f["reset"](self.h)
    def levels(self) -> int: return self.f["levels"](self.h)
    def scale_to_map(self) -> float: return self.f["scale_to_map"](self.h)
    def set_update_factor_free(self, v): self.f["set_update_factor_free"](self.h, v)
    def set_update_factor_occupied(self, v): self.f["set_update_factor_occupied"](self.h, v)
    def on_map_updated(self): self.f["on_map_updated"](self.h)

    def undefined_reads(self) -> int:
        """"ho": map reads with a NaN coordinate so far (where the reference would index the grid with (int)NaN and crash);
        "hr": -1"""
        return int(self.f["undefined_reads"](self.h))

    def level_info(self, level):
        sx, sy, cell, scale = C.c_int(), C.c_int(), C.c_float(), C.c_float()
        self.f["level_info"](self.h, level, C.byref(sx), C.byref(sy), C.byref(cell), C.byref(scale))
        return sx.value, sy.value, cell.value, scale.value

    def download_level(self, level):
        sx, sy, _, _ = self.level_info(level)
        lo = np.empty(sx * sy, np.float32)
        ui = np.empty(sx * sy, np.int32)
        self.f["download_level"](self.h, level, lo, ui)
        return lo.reshape(sy, sx), ui.reshape(sy, sx)

    def upload_level(self, level, logodds, update_index):
        self.f["upload_level"](self.h, level, np.ascontiguousarray(logodds, np.float32).reshape(-1),
                               np.ascontiguousarray(update_index, np.int32).reshape(-1))

    def map_coords_pose(self, level, world):
        out = np.empty(3, np.float32)
        self.f["map_coords_pose"](self.h, level, _v(world, 3), out)
        return out

    def world_coords_pose(self, level, mp):
        out = np.empty(3, np.float32)
        self.f["world_coords_pose"](self.h, level, _v(mp, 3), out)
        return out

    def interp(self, level, coords):
        c = _pts(coords)
        out = np.empty((c.shape[0], 3), np.float32)
        self.f["interp"](self.h, level, c.reshape(-1), c.shape[0], out.reshape(-1))
        return out

    def hessian_derivs(self, level, pose_map, pts_level):
        p = _pts(pts_level)
        H = np.empty(9, np.float32)
        d = np.empty(3, np.float32)
        self.f["hessian_derivs"](self.h, level, _v(pose_map, 3), p.reshape(-1), p.shape[0], H, d)
        return H.reshape(3, 3).T.copy(), d  # column-major -> numpy [r, c]

    def match_level(self, level, begin_world, pts_level, max_iter, cov=None):
        p = _pts(pts_level)
        out = np.empty(3, np.float32)
        c = np.zeros(9, np.float32) if cov is None else _v(cov, 9).copy()
        self.f["match_level"](self.h, level, _v(begin_world, 3), p.reshape(-1), p.shape[0], max_iter, out, c)
        return out, c

    def match(self, begin_world, pts, origo=_ZERO2, cov=None):
        p = _pts(pts)
        out = np.empty(3, np.float32)
        c = np.zeros(9, np.float32) if cov is None else _v(cov, 9).copy()
        self.f["match"](self.h, _v(begin_world, 3), p.reshape(-1), p.shape[0], _v(origo, 2), out, c)
        return out, c

    def match_many(self, begin_world, pts, offsets):
        """len(offsets)-1 matchData calls in one C loop (CPU-baseline timing helper)."""
        b = np.ascontiguousarray(begin_world, np.float32).reshape(-1, 3)
        out = np.empty_like(b)
        self.f["match_many"](self.h, b.shape[0], b.reshape(-1), np.ascontiguousarray(pts, np.float32).reshape(-1),
                             np.ascontiguousarray(offsets, np.int32), out.reshape(-1))
        return out

    def update_by_scan(self, pose_world, pts, origo=_ZERO2):
        p = _pts(pts)
        self.f["update_by_scan"](self.h, _v(pose_world, 3), p.reshape(-1), p.shape[0], _v(origo, 2))

    def update_by_scan_level(self, level, pose_world, pts_level, origo_level=_ZERO2):
        p = _pts(pts_level)
        self.f["update_by_scan_level"](self.h, level, _v(pose_world, 3), p.reshape(-1), p.shape[0],
                                       _v(origo_level, 2))

    # ---- processor ------------------------------------------------------------
    def proc_set_thresholds(self, d, a): self.f["proc_set_thresholds"](self.h, d, a)

    def proc_update(self, pts, hint_world, origo=_ZERO2, map_without_matching=False):
        p = _pts(pts)
        self.f["proc_update"](self.h, p.reshape(-1), p.shape[0], _v(origo, 2), _v(hint_world, 3),
                              1 if map_without_matching else 0)

    def proc_last_pose(self):
        pose = np.empty(3, np.float32)
        cov = np.empty(9, np.float32)
        self.f["proc_last_pose"](self.h, pose, cov)
        return pose, cov

    def likelihood_states(self, level, states_map, pts_level):
        st = np.ascontiguousarray(states_map, np.float32).reshape(-1, 3)
        p = _pts(pts_level)
        out = np.empty(st.shape[0], np.float32)
        self.f["likelihood_states"](self.h, level, st.shape[0], st.reshape(-1), p.reshape(-1) if p.size else
                                    np.zeros(2, np.float32), p.shape[0], out)
        return out

    def residual_states(self, level, states_map, pts_level):
        st = np.ascontiguousarray(states_map, np.float32).reshape(-1, 3)
        p = _pts(pts_level)
        out = np.empty(st.shape[0], np.float32)
        self.f["residual_states"](self.h, level, st.shape[0], st.reshape(-1), p.reshape(-1) if p.size else
                                  np.zeros(2, np.float32), p.shape[0], out)
        return out

    def covariance_for_poses(self, level, poses_map, pts_level):
        """-> (cov_map [B,9], cov_world [B,9], likelihoods [B,7]); column major"""
        st = np.ascontiguousarray(poses_map, np.float32).reshape(-1, 3)
        p = _pts(pts_level)
        B = st.shape[0]
        cm, cw, lh = np.zeros((B, 9), np.float32), np.zeros((B, 9), np.float32), np.zeros((B, 7), np.float32)
        self.f["covariance_for_poses"](self.h, level, B, st.reshape(-1), p.reshape(-1) if p.size else
                                       np.zeros(2, np.float32), p.shape[0], cm.reshape(-1), cw.reshape(-1),
                                       lh.reshape(-1))
        return cm, cw, lh

    def occupancy_grid(self, level):
        sx, sy, _, _ = self.level_info(level)
        out = np.empty((sy, sx), np.int8)
        self.f["occupancy_grid"](self.h, level, out)
        return out

    def laser_scan_to_container(self, ranges, angle_min, angle_increment, range_min, range_max, scale_to_map):
        r = np.ascontiguousarray(ranges, np.float32)
        out = np.empty(2 * r.size, np.float32)
        m = self.f["laser_scan_to_container"](r, r.size, angle_min, angle_increment, range_min, range_max,
                                              scale_to_map, out)
        return out[:2 * m].reshape(m, 2).copy()

    def point_cloud_to_container(self, pts_xyz, tf_rows, sqr_min, sqr_max, z_min, z_max, scale_to_map):
        """-> (endpoints [m,2], origo [2])"""
        p = np.ascontiguousarray(pts_xyz, np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(tf_rows, np.float64).reshape(12)
        out = np.empty(2 * max(p.shape[0], 1), np.float32)
        origo = np.empty(2, np.float32)
        m = self.f["point_cloud_to_container"](p.reshape(-1), p.shape[0], T, sqr_min, sqr_max, z_min, z_max,
                                               scale_to_map, out, origo)
        return out[:2 * m].reshape(m, 2).copy(), origo

    def project_laser(self, ranges, angle_min, angle_increment, range_min, range_max, range_cutoff):
        r = np.ascontiguousarray(ranges, np.float32)
        out = np.empty(3 * max(r.size, 1), np.float32)
        m = self.f["project_laser"](r, r.size, angle_min, angle_increment, range_min, range_max, range_cutoff, out)
        return out[:3 * m].reshape(m, 3).copy()

    def normalize_angle(self, a) -> float:
        return self.f["normalize_angle"](float(a))

    def pose_difference_larger_than(self, p1, p2, d, a) -> bool:
        return bool(self.f["pose_difference_larger_than"](_v(p1, 3), _v(p2, 3), d, a))

    # ---- convenience ----------------------------------------------------------
    def build_map(self, poses, scans, origo=_ZERO2):
        """Map from ground-truth poses: match-free updates on every level (the retained
        coarse containers are refreshed through match's setFrom, as in the processor)."""
        for pose, pts in zip(poses, scans):
            # proc_update(map_without_matching) would leave coarse levels stale (row a12);
            # drive each level explicitly with the level-scaled container instead.
            for lvl in range(self.n_levels):
                f = np.float32(1.0 / 2.0 ** lvl)
                self.update_by_scan_level(lvl, pose, (np.asarray(pts, np.float32) * f),
                                          np.asarray(origo, np.float32) * f)
            self.on_map_updated()

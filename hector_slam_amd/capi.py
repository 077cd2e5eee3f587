"""ctypes binding of the C ABI (include/hector_mi355/capi.h) + a thin Python mirror of the
reference's map-representation interface.

``MapRepMultiMap`` keeps the reference's method names and argument meaning
(hector_slam_lib/slam_main/MapRepMultiMap.h, MapRepresentationInterface.h:38-62) so the
parity tests read like calls into the reference.  Everything computes on the GPU through
``libhector_mi355.so``; if the library is missing, or no HIP device is present, construction
raises -- there is no CPU fallback in this package.

torch is imported first on purpose: PyTorch-ROCm bundles its own ``libamdhip64.so`` with the
same SONAME as /opt/rocm's; loading torch first makes the dynamic loader bind this library
to that single HIP runtime, so torch device pointers / streams and ours interoperate.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:  # plumbing only (device memory, streams, torch.distributed); see module docstring
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is always present in the target image
    torch = None

from . import build as _build

HSM_OK = 0
LAYOUT_AUTO, LAYOUT_QUAD, LAYOUT_PLANE = 0, 1, 2
PARITY_FAST, PARITY_EXACT, PARITY_RELAXED, PARITY_AUTO = 0, 1, 2, 3
GATHER_AUTO, GATHER_PEER, GATHER_RCCL, GATHER_DIRECT = 0, 1, 2, 3
ORDER_GIVEN, ORDER_MORTON, ORDER_AUTO = 0, 1, 2
EXCHANGE_HANDLE_BYTES = 64

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class HsmOpts(C.Structure):
    _fields_ = [("device", C.c_int), ("layout", C.c_int), ("waves_per_scan", C.c_int)]


class HsmError(RuntimeError):
    pass


# every symbol include/hector_mi355/capi.h declares: name -> (restype, argtypes)
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SIGNATURES = {
    "hsm_create": (_i, [_f, _i, _i, C.c_uint, _f, _f, C.POINTER(HsmOpts), C.POINTER(_vp)]),
    "hsm_destroy": (None, [_vp]),
    "hsm_reset": (_i, [_vp]),
    "hsm_levels": (_i, [_vp]),
    "hsm_scale_to_map": (_f, [_vp]),
    "hsm_set_update_factor_free": (_i, [_vp, _f]),
    "hsm_set_update_factor_occupied": (_i, [_vp, _f]),
    "hsm_on_map_updated": (_i, [_vp]),
    "hsm_set_parity": (_i, [_vp, _i]),
    "hsm_parity": (_i, [_vp]),
    "hsm_last_launch_parity": (_i, [_vp]),
    "hsm_set_batch_order": (_i, [_vp, _i]),
    "hsm_set_batch_order_refresh": (_i, [_vp, _i]),
    "hsm_batch_order": (_i, [_vp]),
    "hsm_last_launch_sorted": (_i, [_vp]),
    "hsm_match": (_i, [_vp, _f32p, _vp, _i, _f32p, _f32p, _f32p]),
    "hsm_match_trace": (_i, [_vp, _f32p, _vp, _i, _f32p, _f32p, _f32p, _f32p, _i, C.POINTER(_i)]),
    "hsm_match_batch_device": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "hsm_match_batch": (_i, [_vp, _i, _f32p, _vp, _vp, _i, _f32p, _vp]),
    "hsm_update_by_scan": (_i, [_vp, _f32p, _vp, _i, _f32p]),
    "hsm_update_by_scan_level": (_i, [_vp, _i, _f32p, _vp, _i, _f32p]),
    "hsm_ingest_laser_scan": (_i, [_vp, _vp, _i, _f, _f, _f, _f, _f, _vp, C.POINTER(_i)]),
    "hsm_ingest_point_cloud": (_i, [_vp, _vp, _i, _vp, _f, _f, _f, _f, _f, _vp, C.POINTER(_i), _vp]),
    "hsm_ingest_laser_scan_tf": (_i, [_vp, _vp, _i, _f, _f, _f, _f, C.c_double, _vp, _f, _f, _f, _f, _f, _vp,
                                      C.POINTER(_i), _vp]),
    "hsm_synchronize": (_i, [_vp]),
    "hsm_match_ingested": (_i, [_vp, _f32p, _f32p, _f32p]),
    "hsm_update_by_ingested": (_i, [_vp, _f32p]),
    "hsm_occupancy_grid": (_i, [_vp, _i, _vp]),
    "hsm_ray_distances": (_i, [_vp, _i, _f, _f, _f, _i, _f32p, _f32p, _f32p, _f32p]),
    "hsm_likelihood_states": (_i, [_vp, _i, _i, _f32p, _vp, _i, _f32p]),
    "hsm_residual_states": (_i, [_vp, _i, _i, _f32p, _vp, _i, _f32p]),
    "hsm_covariance_for_poses": (_i, [_vp, _i, _i, _f32p, _vp, _i, _f32p, _f32p, _f32p]),
    "hsm_group_create": (_i, [_f, _i, _i, C.c_uint, _f, _f, _i32p, _i, C.POINTER(_vp)]),
    "hsm_group_destroy": (None, [_vp]),
    "hsm_group_size": (_i, [_vp]),
    "hsm_group_member": (_vp, [_vp, _i]),
    "hsm_group_set_update_factors": (_i, [_vp, _f, _f]),
    "hsm_group_process_scan": (_i, [_vp, _f32p, _vp, _i, _f32p, _i, _f32p, _f32p]),
    "hsm_group_match_batch": (_i, [_vp, _i, _f32p, _vp, _vp, _i, _f32p, _vp]),
    "hsm_group_match_batch_device": (_i, [_vp, _i32p, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "hsm_group_synchronize": (_i, [_vp]),
    "hsm_group_debug_force_p2p": (_i, [_vp, _i]),
    "hsm_shard_bounds": (_i, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "hsm_group_set_gather": (_i, [_vp, _i]),
    "hsm_group_gather_mode": (_i, [_vp]),
    "hsm_group_gather_note": (C.c_char_p, [_vp]),
    "hsm_group_gathered": (_vp, [_vp, _i, _i]),
    "hsm_retain_scan": (_i, [_vp, _vp, _i, _f32p]),
    "hsm_exchange_create": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "hsm_exchange_destroy": (None, [_vp]),
    "hsm_exchange_handle": (_i, [_vp, _vp]),
    "hsm_exchange_connect": (_i, [_vp, _vp]),
    "hsm_exchange_connect_local": (_i, [_vp, C.POINTER(_vp)]),
    "hsm_exchange_post": (_i, [_vp, _vp, _i, _i, _vp]),
    "hsm_exchange_wait": (_i, [_vp, _vp, _vp]),
    "hsm_exchange_post_wait": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "hsm_match_batch_device_gather": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "hsm_exchange_epochs": (_i, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "hsm_exchange_status": (_i, [_vp]),
    "hsm_exchange_memory_kind": (C.c_char_p, [_vp]),
    "hsm_level_info": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f), C.POINTER(_f)]),
    "hsm_map_coords_pose": (_i, [_vp, _i, _f32p, _f32p]),
    "hsm_world_coords_pose": (_i, [_vp, _i, _f32p, _f32p]),
    "hsm_update_index": (_i, [_vp, _i]),
    "hsm_download_level": (_i, [_vp, _i, _vp, _vp]),
    "hsm_upload_level": (_i, [_vp, _i, _vp, _vp]),
    "hsm_download_rows": (_i, [_vp, _i, _i, _i, _f32p]),
    "hsm_download_cells": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _i]),
    "hsm_last_update_bbox": (_i, [_vp, _i, _i32p]),
    "hsm_take_dirty_bbox": (_i, [_vp, _i, _i32p]),
    "hsm_download_prob": (_i, [_vp, _i, _f32p]),
    "hsm_hessian_derivs": (_i, [_vp, _i, _f32p, _vp, _i, _f32p, _f32p]),
    "hsm_eval_beams": (_i, [_vp, _i, _f32p, _vp, _i, _vp]),
    "hsm_match_level": (_i, [_vp, _i, _f32p, _vp, _i, _i, _f32p, _f32p]),
    "hsm_debug_set_update_serial": (_i, [_vp, _i, C.c_uint]),
    "hsm_debug_set_coop_barrier": (_i, [_vp, C.c_uint]),
    "hsm_debug_set_coop_mute": (_i, [_vp, _i]),
    "hsm_debug_coop_fallbacks": (_i, [_vp]),
    "hsm_debug_spec_stats": (_i, [_vp, _i, _vp]),
    "hsm_debug_marks_nonzero": (_i, [_vp, _i, _vp]),
    "hsm_debug_sincos": (_i, [_vp, _i, _f32p, _f32p, _f32p]),
    "hsm_debug_expf": (_i, [_vp, _i, _f32p, _f32p, _f32p]),
    "hsm_device_info": (_i, [_vp, _i32p]),
    "hsm_set_clock_probe": (_i, [_vp, _vp]),
    "hsm_gn_iterations_per_match": (_i, [_vp]),
    "hsm_last_launch_kernel": (C.c_char_p, [_vp]),
    "hsm_last_launch_config": (_i, [_vp, _i32p]),
    "hsm_last_error": (C.c_char_p, []),
    "hsm_version": (C.c_char_p, []),
}

_lib = None


def library_path() -> str:
    return _build.LIB


def load_library(build_if_missing: bool = True):
    """dlopen libhector_mi355.so (building it with hipcc when stale) and type every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build_native() if build_if_missing else _build.LIB
    if os.environ.get("HSM_LIB"):  # kernel A/B experiments: an alternative build of the same library
        path = os.environ["HSM_LIB"]
    if not os.path.exists(path):
        raise HsmError(f"{path} not found and could not be built; hector_slam_amd has no CPU fallback")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch: fail loudly
        fn.restype, fn.argtypes = res, args
    # the node's two per-scan calls a second time with raw-address prototypes: numpy's ndpointer argument check costs
    # ~2 us per array argument, which is a third of a 40 us call (lib["name"] creates a separate function object)
    lib._hsm_match_raw = lib["hsm_match"]
    lib._hsm_match_raw.restype, lib._hsm_match_raw.argtypes = _i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]
    lib._hsm_update_raw = lib["hsm_update_by_scan"]
    lib._hsm_update_raw.restype, lib._hsm_update_raw.argtypes = _i, [_vp, _vp, _vp, _i, _vp]
    _lib = lib
    return lib


def shard_bounds(total: int, rank: int, world: int) -> tuple[int, int]:
    """[begin, end) of shard `rank` of `total` scans over `world` replicas: the library's ONE partitioning rule, asked of the
    library itself (hsm_shard_bounds).  sharding.shard_bounds is the same closed form in Python -- it must not need a HIP
    toolchain or a built library for an index computation; tests/test_sharding_cpu.py holds the two equal."""
    b, e = C.c_int(0), C.c_int(0)
    _check(load_library().hsm_shard_bounds(int(total), int(rank), int(world), C.byref(b), C.byref(e)), "hsm_shard_bounds")
    return b.value, e.value


def _check(rc: int, what: str):
    if rc != HSM_OK:
        raise HsmError(f"{what} failed ({rc}): {load_library().hsm_last_error().decode()}")


def _addr(a) -> int:
    """address of a numpy array's first element (cheaper than a.ctypes.data)"""
    return a.__array_interface__["data"][0]


def _is_f32c(x) -> bool:
    return type(x) is np.ndarray and x.dtype == np.float32 and x.flags.c_contiguous


def _pts(pts):
    if _is_f32c(pts) and pts.ndim == 2 and pts.shape[1] == 2:
        a = pts
    else:
        a = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 2)
    return a, (_addr(a) if a.size else None), a.shape[0]


def _v(x, n):
    if _is_f32c(x) and x.ndim == 1:
        a = x
    else:
        a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    if a.size != n:
        raise ValueError(f"expected {n} floats")
    return a


def _fill(dst, x, n):
    """dst[:] = the n floats of x (ValueError on any other size, as _v)"""
    if type(x) is not np.ndarray:
        x = np.asarray(x, dtype=np.float32)
    if x.size != n:
        raise ValueError(f"expected {n} floats")
    dst[:] = x.reshape(-1)


class _SmallArgs:
    __slots__ = ("buf", "pose", "origo", "out", "cov", "a_pose", "a_origo", "a_out", "a_cov", "lock")

    def __init__(self):
        import threading
        self.buf = np.zeros(3 + 2 + 3 + 9 + 3, np.float32)
        self.pose, self.origo, self.out, self.cov = self.buf[0:3], self.buf[3:5], self.buf[5:8], self.buf[8:17]
        base = _addr(self.buf)
        self.a_pose, self.a_origo, self.a_out, self.a_cov = base, base + 12, base + 20, base + 32
        self.lock = threading.Lock()


_ZERO2 = np.zeros(2, np.float32)


class MapRepMultiMap:
    """GPU-resident multi-resolution map representation (reference: MapRepMultiMap.h:45-173)."""

    def __init__(self, mapResolution: float, mapSizeX: int, mapSizeY: int, numDepth: int,
                 startCoords=(0.5, 0.5), device: int = -1, layout: int = LAYOUT_AUTO,
                 waves_per_scan: int = 0, parity: int | None = None):
        self._lib = load_library()
        self._h = _vp()
        opts = HsmOpts(device, layout, waves_per_scan)
        _check(self._lib.hsm_create(mapResolution, mapSizeX, mapSizeY, numDepth, startCoords[0],
                                    startCoords[1], C.byref(opts), C.byref(self._h)), "hsm_create")
        self._iobuf = _SmallArgs()  # created here, not on first use: two threads' first calls must meet at ONE lock
        if parity is not None:
            self.set_parity(parity)

    def _io(self):
        """small-argument block of the two per-scan calls: one persistent float32 array whose addresses are taken ONCE
        (extracting an array's address costs ~1 us, a 3-float copy 0.3 us); the lock keeps concurrent Python callers
        of one map apart (ctypes releases the GIL during the call)"""
        io = getattr(self, "_iobuf", None)
        if io is None:
            io = self._iobuf = _SmallArgs()
        return io

    def set_parity(self, mode: int):
        """PARITY_FAST (tree summation) / PARITY_EXACT (the reference's beam-order fp32 chains: bit-identical poses)"""
        _check(self._lib.hsm_set_parity(self._h, mode), "hsm_set_parity")

    def parity(self) -> int:
        return self._lib.hsm_parity(self._h)

    def set_batch_order(self, order: int):
        """ORDER_AUTO (default) / ORDER_GIVEN / ORDER_MORTON: how a batch is laid out on the device (hsm_set_batch_order)"""
        _check(self._lib.hsm_set_batch_order(self._h, order), "hsm_set_batch_order")

    def set_batch_order_refresh(self, launches: int):
        """a stream's permutation serves that many launches of the same batch size before it is computed again (default 16)"""
        _check(self._lib.hsm_set_batch_order_refresh(self._h, launches), "hsm_set_batch_order_refresh")

    def batch_order(self) -> int:
        return self._lib.hsm_batch_order(self._h)

    def last_launch_sorted(self) -> bool:
        return bool(self._lib.hsm_last_launch_sorted(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hsm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- MapRepresentationInterface -------------------------------------------------
    def reset(self): _check(self._lib.hsm_reset(self._h), "hsm_reset")
    def getScaleToMap(self) -> float: return self._lib.hsm_scale_to_map(self._h)
    def getMapLevels(self) -> int: return self._lib.hsm_levels(self._h)
    def onMapUpdated(self): _check(self._lib.hsm_on_map_updated(self._h), "hsm_on_map_updated")

    def setUpdateFactorFree(self, v: float):
        _check(self._lib.hsm_set_update_factor_free(self._h, v), "hsm_set_update_factor_free")

    def setUpdateFactorOccupied(self, v: float):
        _check(self._lib.hsm_set_update_factor_occupied(self._h, v), "hsm_set_update_factor_occupied")

    def matchData(self, beginEstimateWorld, dataContainer, covMatrix=None, origo=_ZERO2):
        """-> (newEstimateWorld[3], covMatrix[9] column-major); cov is in/out like the reference."""
        a, p, n = _pts(dataContainer)
        io = self._io()
        with io.lock:
            _fill(io.pose, beginEstimateWorld, 3)
            _fill(io.origo, origo, 2)
            if covMatrix is None:
                io.cov[:] = 0.0
            else:
                _fill(io.cov, covMatrix, 9)
            rc = self._lib._hsm_match_raw(self._h, io.a_pose, p, n, io.a_origo, io.a_out, io.a_cov)
            out, cov = io.out.copy(), io.cov.copy()
        if rc != HSM_OK:
            _check(rc, "hsm_match")
        return out, cov

    def updateByScan(self, dataContainer, robotPoseWorld, origo=_ZERO2):
        a, p, n = _pts(dataContainer)
        io = self._io()
        with io.lock:
            _fill(io.pose, robotPoseWorld, 3)
            _fill(io.origo, origo, 2)
            rc = self._lib._hsm_update_raw(self._h, io.a_pose, p, n, io.a_origo)
        if rc != HSM_OK:
            _check(rc, "hsm_update_by_scan")

    def synchronize(self):
        """wait for queued device work (updateByScan returns once its kernels are queued)"""
        _check(self._lib.hsm_synchronize(self._h), "hsm_synchronize")

    # ---- GridMap accessors (host mirror support) ---------------------------------------
    def level_info(self, level: int):
        sx, sy, cell, scale = _i(), _i(), _f(), _f()
        _check(self._lib.hsm_level_info(self._h, level, C.byref(sx), C.byref(sy), C.byref(cell),
                                        C.byref(scale)), "hsm_level_info")
        return sx.value, sy.value, cell.value, scale.value

    def getMapCoordsPose(self, level, world):
        out = np.empty(3, np.float32)
        _check(self._lib.hsm_map_coords_pose(self._h, level, _v(world, 3), out), "hsm_map_coords_pose")
        return out

    def getWorldCoordsPose(self, level, mp):
        out = np.empty(3, np.float32)
        _check(self._lib.hsm_world_coords_pose(self._h, level, _v(mp, 3), out), "hsm_world_coords_pose")
        return out

    def getUpdateIndex(self, level=0) -> int:
        return self._lib.hsm_update_index(self._h, level)

    def download_level(self, level):
        sx, sy, _, _ = self.level_info(level)
        lo = np.empty((sy, sx), np.float32)
        ui = np.empty((sy, sx), np.int32)
        _check(self._lib.hsm_download_level(self._h, level, lo.ctypes.data, ui.ctypes.data), "hsm_download_level")
        return lo, ui

    def upload_level(self, level, logodds, update_index=None):
        lo = np.ascontiguousarray(logodds, np.float32)
        ui = None if update_index is None else np.ascontiguousarray(update_index, np.int32)
        _check(self._lib.hsm_upload_level(self._h, level, lo.ctypes.data, None if ui is None else ui.ctypes.data),
               "hsm_upload_level")

    def download_rows(self, level, y0, y1):
        sx, _, _, _ = self.level_info(level)
        out = np.empty((max(y1 - y0, 0), sx), np.float32)
        _check(self._lib.hsm_download_rows(self._h, level, y0, y1, out.reshape(-1) if out.size else
                                           np.zeros(1, np.float32)), "hsm_download_rows")
        return out

    def last_update_bbox(self, level):
        bb = np.empty(4, np.int32)
        _check(self._lib.hsm_last_update_bbox(self._h, level, bb), "hsm_last_update_bbox")
        return bb

    def take_dirty_bbox(self, level):
        bb = np.empty(4, np.int32)
        _check(self._lib.hsm_take_dirty_bbox(self._h, level, bb), "hsm_take_dirty_bbox")
        return bb

    def download_prob(self, level):
        sx, sy, _, _ = self.level_info(level)
        out = np.empty((sy, sx), np.float32)
        _check(self._lib.hsm_download_prob(self._h, level, out.reshape(-1)), "hsm_download_prob")
        return out

    # ---- rows next to the path (ROS-node side) -------------------------------------
    def ingest_laser_scan(self, ranges, angle_min, angle_increment, range_min, range_max, scale_to_map=None):
        """rosLaserScanToDataContainer on the device; returns the (n_valid, 2) endpoints (host copy)."""
        r = np.ascontiguousarray(ranges, np.float32).reshape(-1)
        out = np.empty((r.size, 2), np.float32)
        m = _i()
        s = self.getScaleToMap() if scale_to_map is None else scale_to_map
        _check(self._lib.hsm_ingest_laser_scan(self._h, r.ctypes.data if r.size else None, r.size, angle_min,
                                               angle_increment, range_min, range_max, s, out.ctypes.data,
                                               C.byref(m)), "hsm_ingest_laser_scan")
        return out[:m.value].copy()

    def ingest_point_cloud(self, pts_xyz, tf_rows, sqr_laser_min_dist, sqr_laser_max_dist, laser_z_min,
                           laser_z_max, scale_to_map=None):
        """rosPointCloudToDataContainer on the device; tf_rows = laser->base transform, 12 doubles [R | t]
        row major.  Returns (endpoints (m, 2), origo (2,))."""
        p = np.ascontiguousarray(pts_xyz, np.float32).reshape(-1, 3)
        T = np.ascontiguousarray(tf_rows, np.float64).reshape(12)
        out = np.empty((max(p.shape[0], 1), 2), np.float32)
        origo = np.empty(2, np.float32)
        m = _i()
        s = self.getScaleToMap() if scale_to_map is None else scale_to_map
        _check(self._lib.hsm_ingest_point_cloud(self._h, p.ctypes.data if p.size else None, p.shape[0],
                                                T.ctypes.data, sqr_laser_min_dist, sqr_laser_max_dist, laser_z_min,
                                                laser_z_max, s, out.ctypes.data, C.byref(m), origo.ctypes.data),
               "hsm_ingest_point_cloud")
        return out[:m.value].copy(), origo

    def ingest_laser_scan_tf(self, ranges, angle_min, angle_increment, range_min, range_max, range_cutoff, tf_rows,
                             sqr_laser_min_dist, sqr_laser_max_dist, laser_z_min, laser_z_max, scale_to_map=None):
        """projectLaser + rosPointCloudToDataContainer fused on the device (the node's default path)"""
        r = np.ascontiguousarray(ranges, np.float32).reshape(-1)
        T = np.ascontiguousarray(tf_rows, np.float64).reshape(12)
        out = np.empty((max(r.size, 1), 2), np.float32)
        origo = np.empty(2, np.float32)
        m = _i()
        s = self.getScaleToMap() if scale_to_map is None else scale_to_map
        _check(self._lib.hsm_ingest_laser_scan_tf(self._h, r.ctypes.data if r.size else None, r.size, angle_min,
                                                  angle_increment, range_min, range_max, range_cutoff, T.ctypes.data,
                                                  sqr_laser_min_dist, sqr_laser_max_dist, laser_z_min, laser_z_max,
                                                  s, out.ctypes.data, C.byref(m), origo.ctypes.data),
               "hsm_ingest_laser_scan_tf")
        return out[:m.value].copy(), origo

    def match_ingested(self, beginEstimateWorld, covMatrix=None):
        out = np.empty(3, np.float32)
        cov = np.zeros(9, np.float32) if covMatrix is None else _v(covMatrix, 9).copy()
        _check(self._lib.hsm_match_ingested(self._h, _v(beginEstimateWorld, 3), out, cov), "hsm_match_ingested")
        return out, cov

    def update_by_ingested(self, robotPoseWorld):
        _check(self._lib.hsm_update_by_ingested(self._h, _v(robotPoseWorld, 3)), "hsm_update_by_ingested")

    def likelihood_states(self, level, states_map, pts):
        """OccGridMapUtil::getLikelihoodForState for a batch of map-frame states; pts in level-0 units"""
        st = np.ascontiguousarray(states_map, np.float32).reshape(-1, 3)
        a, p, n = _pts(pts)
        out = np.empty(st.shape[0], np.float32)
        _check(self._lib.hsm_likelihood_states(self._h, level, st.shape[0], st.reshape(-1), p, n, out),
               "hsm_likelihood_states")
        return out

    def residual_states(self, level, states_map, pts):
        """OccGridMapUtil::getResidualForState for a batch of map-frame states"""
        st = np.ascontiguousarray(states_map, np.float32).reshape(-1, 3)
        a, p, n = _pts(pts)
        out = np.empty(st.shape[0], np.float32)
        _check(self._lib.hsm_residual_states(self._h, level, st.shape[0], st.reshape(-1), p, n, out),
               "hsm_residual_states")
        return out

    def covariance_for_poses(self, level, poses_map, pts):
        """OccGridMapUtil::getCovarianceForPose + getCovMatrixWorldCoords for a batch of map-frame poses
        -> (cov_map [B,9], cov_world [B,9], likelihoods [B,7]); matrices column major"""
        st = np.ascontiguousarray(poses_map, np.float32).reshape(-1, 3)
        a, p, n = _pts(pts)
        B = st.shape[0]
        cm, cw, lh = (np.zeros((B, 9), np.float32), np.zeros((B, 9), np.float32), np.zeros((B, 7), np.float32))
        _check(self._lib.hsm_covariance_for_poses(self._h, level, B, st.reshape(-1), p, n, cm.reshape(-1),
                                                  cw.reshape(-1), lh.reshape(-1)), "hsm_covariance_for_poses")
        return cm, cw, lh

    def map_metadata(self, level=0):
        """(origin_x, origin_y, resolution) as HectorMappingRos::setServiceGetMapData publishes them (:546-553)"""
        sx, sy, cell, _ = self.level_info(level)
        w = self.getWorldCoordsPose(level, np.zeros(3, np.float32))  # getWorldCoords(Vector2f::Zero())
        half = np.float32(cell) * np.float32(0.5)
        return float(np.float32(w[0]) - half), float(np.float32(w[1]) - half), float(cell)

    def ray_distances(self, level, begin_world, end_world, origin_xy=None, resolution=None):
        """DistanceMeasurementProvider::getDist for a batch of rays -> (dist[n], hit[n,2]; NaN where no hit)"""
        b = np.ascontiguousarray(begin_world, np.float32).reshape(-1, 2)
        e = np.ascontiguousarray(end_world, np.float32).reshape(-1, 2)
        ox, oy, res = self.map_metadata(level)
        if origin_xy is not None:
            ox, oy = origin_xy
        if resolution is not None:
            res = resolution
        dist = np.empty(b.shape[0], np.float32)
        hit = np.full((b.shape[0], 2), np.nan, np.float32)
        if b.shape[0]:
            _check(self._lib.hsm_ray_distances(self._h, level, ox, oy, res, b.shape[0], b.reshape(-1), e.reshape(-1),
                                               dist, hit.reshape(-1)), "hsm_ray_distances")
        return dist, hit

    def occupancy_grid(self, level=0):
        """publishMap's int8 grid: -1 unknown, 0 free, 100 occupied"""
        sx, sy, _, _ = self.level_info(level)
        out = np.empty((sy, sx), np.int8)
        _check(self._lib.hsm_occupancy_grid(self._h, level, out.ctypes.data), "hsm_occupancy_grid")
        return out

    # ---- batched extension ---------------------------------------------------------------
    def match_batch(self, begin_world, pts, offsets=None, want_cov=True):
        """Host arrays in/out.  ``offsets`` None = every hypothesis uses the same scan ``pts``."""
        b = np.ascontiguousarray(begin_world, np.float32).reshape(-1, 3)
        a, p, n = _pts(pts)
        out = np.empty_like(b)
        cov = np.zeros((b.shape[0], 9), np.float32) if want_cov else None
        offs = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
        _check(self._lib.hsm_match_batch(self._h, b.shape[0], b.reshape(-1), p,
                                         None if offs is None else offs.ctypes.data,
                                         n if offs is None else 0, out.reshape(-1),
                                         None if cov is None else cov.ctypes.data), "hsm_match_batch")
        return out, cov

    def match_batch_device(self, batch, d_begin, d_pts, d_offsets, shared_n, d_out_pose, d_out_cov, stream=0):
        """Raw device pointers (ints), asynchronous on ``stream`` (a hipStream_t value)."""
        _check(self._lib.hsm_match_batch_device(self._h, batch, d_begin, d_pts, d_offsets or None, shared_n,
                                                d_out_pose, d_out_cov or None, stream or None),
               "hsm_match_batch_device")

    def match_batch_device_gather(self, batch, d_begin, d_pts, d_offsets, shared_n, d_out_pose, d_out_cov, exchange, first_row, lag,
                                  d_out_all, stream=0):
        """match_batch_device + one step of `exchange` (a PoseExchange) in one call; the launch carries the exchange where it can"""
        _check(self._lib.hsm_match_batch_device_gather(self._h, batch, d_begin, d_pts, d_offsets, shared_n, d_out_pose, d_out_cov or None,
                                                       exchange._h, int(first_row), int(lag), d_out_all or None, stream),
               "hsm_match_batch_device_gather")

    def device_info(self):
        a = np.empty(4, np.int32)
        _check(self._lib.hsm_device_info(self._h, a), "hsm_device_info")
        return {"device": int(a[0]), "compute_units": int(a[1]), "clock_khz": int(a[2]), "memory_clock_khz": int(a[3])}

    def set_clock_probe(self, d_stamps4: int):
        """device pointer (int) to four uint64 words, or 0: see hsm_set_clock_probe"""
        _check(self._lib.hsm_set_clock_probe(self._h, d_stamps4 or None), "hsm_set_clock_probe")

    def gn_iterations_per_match(self) -> int:
        return self._lib.hsm_gn_iterations_per_match(self._h)

    def last_launch_config(self):
        cfg = np.empty(5, np.int32)
        _check(self._lib.hsm_last_launch_config(self._h, cfg), "hsm_last_launch_config")
        return {"layout": {1: "quad", 2: "plane"}.get(int(cfg[0]), "?"), "waves_per_scan": int(cfg[1]),
                "block": int(cfg[2]), "grid": int(cfg[3]), "beams_per_lane_in_vgprs": max(int(cfg[4]), 0),
                "texel_cache": bool(cfg[4] < 0), "beams_per_lane": abs(int(cfg[4])),
                "kernel": (self._lib.hsm_last_launch_kernel(self._h) or b"").decode(),
                "parity": {PARITY_EXACT: "exact", PARITY_RELAXED: "relaxed", PARITY_AUTO: "auto"}.get(self.parity(), "fast"),
                "parity_effective": {PARITY_EXACT: "exact", PARITY_RELAXED: "relaxed"}.get(
                    self._lib.hsm_last_launch_parity(self._h), "fast")}

    # ---- parity / debug ---------------------------------------------------------------------
    def hessian_derivs(self, level, pose_map, pts_level):
        a, p, n = _pts(pts_level)
        H = np.empty(9, np.float32)
        d = np.empty(3, np.float32)
        _check(self._lib.hsm_hessian_derivs(self._h, level, _v(pose_map, 3), p, n, H, d), "hsm_hessian_derivs")
        return H.reshape(3, 3).T.copy(), d

    def eval_beams(self, level, pose_map, pts_level):
        a, p, n = _pts(pts_level)
        out = np.empty((n, 4), np.float32)
        _check(self._lib.hsm_eval_beams(self._h, level, _v(pose_map, 3), p, n, out.ctypes.data if n else None),
               "hsm_eval_beams")
        return out

    def debug_sincos(self, x):
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        s, c = np.empty_like(x), np.empty_like(x)
        _check(self._lib.hsm_debug_sincos(self._h, x.size, x, s, c), "hsm_debug_sincos")
        return s, c

    def debug_expf(self, x):
        """device expf(x) and getGridProbability(x)"""
        x = np.ascontiguousarray(x, np.float32).reshape(-1)
        e, p = np.empty_like(x), np.empty_like(x)
        _check(self._lib.hsm_debug_expf(self._h, x.size, x, e, p), "hsm_debug_expf")
        return e, p

    def debug_marks_nonzero(self, level):
        """(non-zero words of the dense update's byte map, non-zero words of the end-cell bitmap): both 0 between updates"""
        out = np.zeros(2, np.uint64)
        _check(self._lib.hsm_debug_marks_nonzero(self._h, level, out.ctypes.data), "hsm_debug_marks_nonzero")
        return int(out[0]), int(out[1])

    def debug_set_coop_barrier(self, value):
        _check(self._lib.hsm_debug_set_coop_barrier(self._h, int(value) & 0xffffffff), "hsm_debug_set_coop_barrier")

    def debug_set_coop_mute(self, block_plus_one):
        _check(self._lib.hsm_debug_set_coop_mute(self._h, int(block_plus_one)), "hsm_debug_set_coop_mute")

    def debug_spec_stats(self, enable=True):
        """(boundaries, candidate == carry, shifted, re-run) of the speculative-carry matcher's stitching passes since the last call"""
        out = (C.c_ulonglong * 4)()
        _check(self._lib.hsm_debug_spec_stats(self._h, 1 if enable else 0, C.cast(out, _vp)), "hsm_debug_spec_stats")
        return tuple(int(x) for x in out)

    def debug_coop_fallbacks(self):
        return int(self._lib.hsm_debug_coop_fallbacks(self._h))

    def match_level(self, level, begin_world, pts_level, max_iter, cov=None):
        a, p, n = _pts(pts_level)
        out = np.empty(3, np.float32)
        c = np.zeros(9, np.float32) if cov is None else _v(cov, 9).copy()
        _check(self._lib.hsm_match_level(self._h, level, _v(begin_world, 3), p, n, max_iter, out, c),
               "hsm_match_level")
        return out, c

    def update_by_scan_level(self, level, pose_world, pts_level, origo_level=_ZERO2):
        a, p, n = _pts(pts_level)
        _check(self._lib.hsm_update_by_scan_level(self._h, level, _v(pose_world, 3), p, n, _v(origo_level, 2)),
               "hsm_update_by_scan_level")

    def build_map(self, poses, scans, origo=_ZERO2):
        """Ground-truth-posed map building on every level (same recipe as the oracle's build_map)."""
        for pose, pts in zip(poses, scans):
            for lvl in range(self.getMapLevels()):
                f = np.float32(1.0 / 2.0 ** lvl)
                self.update_by_scan_level(lvl, pose, np.asarray(pts, np.float32) * f,
                                          np.asarray(origo, np.float32) * f)
            self.onMapUpdated()


class MapRepGroup:
    """single-process multi-GPU group (hsm_group_*): one pyramid replica per listed device"""

    def __init__(self, mapResolution, mapSizeX, mapSizeY, numDepth, devices, startCoords=(0.5, 0.5)):
        self._lib = load_library()
        self._g = _vp()
        dev = np.ascontiguousarray(devices, np.int32)
        _check(self._lib.hsm_group_create(mapResolution, mapSizeX, mapSizeY, numDepth, startCoords[0], startCoords[1],
                                          dev, dev.size, C.byref(self._g)), "hsm_group_create")

    def close(self):
        if getattr(self, "_g", None):
            self._lib.hsm_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return self._lib.hsm_group_size(self._g)

    def member(self, i):
        """borrowed view of replica i (do not close it)"""
        m = MapRepMultiMap.__new__(MapRepMultiMap)
        m._lib = self._lib
        m._h = _vp(self._lib.hsm_group_member(self._g, i))
        m.close = lambda: None
        return m

    def set_update_factors(self, free, occ):
        _check(self._lib.hsm_group_set_update_factors(self._g, free, occ), "hsm_group_set_update_factors")

    def process_scan(self, hint_world, pts, origo=_ZERO2, do_update=True, cov=None):
        a, p, n = _pts(pts)
        out = np.empty(3, np.float32)
        c = np.zeros(9, np.float32) if cov is None else _v(cov, 9).copy()
        _check(self._lib.hsm_group_process_scan(self._g, _v(hint_world, 3), p, n, _v(origo, 2), 1 if do_update else 0,
                                                out, c), "hsm_group_process_scan")
        return out, c

    def match_batch_device(self, counts, d_begin, d_pts, d_offsets, shared_n, root, d_out_pose_all, d_out_cov_all=0):
        """device-resident shards: per-replica lists of raw device pointers (ints); asynchronous (synchronize() waits)"""
        R = self.size()
        cnt = np.ascontiguousarray(counts, np.int32)
        arr = lambda v: (C.c_void_p * R)(*[C.c_void_p(int(x) if x else None) for x in v])
        b, p = arr(d_begin), arr(d_pts)
        o = arr(d_offsets) if d_offsets is not None else None
        _check(self._lib.hsm_group_match_batch_device(self._g, cnt, b, p, o, shared_n, root, d_out_pose_all,
                                                      d_out_cov_all or None), "hsm_group_match_batch_device")

    def synchronize(self):
        _check(self._lib.hsm_group_synchronize(self._g), "hsm_group_synchronize")

    def set_gather(self, mode: int):
        """GATHER_AUTO / GATHER_DIRECT / GATHER_RCCL / GATHER_PEER for match_batch_device (DIRECT or RCCL asked for explicitly raise
        if unavailable)"""
        _check(self._lib.hsm_group_set_gather(self._g, mode), "hsm_group_set_gather")

    def debug_force_p2p(self, on: bool):
        """test hook: the RCCL gather sends every shard, the root's too, through grouped ncclSend / ncclRecv"""
        _check(self._lib.hsm_group_debug_force_p2p(self._g, 1 if on else 0), "hsm_group_debug_force_p2p")

    def gather_mode(self):
        """("direct" | "rccl" | "peer", note): what match_batch_device gathers with (takes the decision if it is still open)"""
        m = self._lib.hsm_group_gather_mode(self._g)
        return {GATHER_PEER: "peer", GATHER_RCCL: "rccl", GATHER_DIRECT: "direct"}.get(m, "undecided"), self._lib.hsm_group_gather_note(self._g).decode()

    def gathered(self, replica, want_cov=False):
        """device pointer (int, 0 = none) of the all-gathered poses / Hessians replica `replica` holds after a direct / RCCL gather"""
        return int(self._lib.hsm_group_gathered(self._g, replica, 1 if want_cov else 0) or 0)

    def match_batch(self, begin_world, pts, offsets=None, want_cov=True):
        b = np.ascontiguousarray(begin_world, np.float32).reshape(-1, 3)
        a, p, n = _pts(pts)
        out = np.empty_like(b)
        cov = np.zeros((b.shape[0], 9), np.float32) if want_cov else None
        offs = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
        _check(self._lib.hsm_group_match_batch(self._g, b.shape[0], b.reshape(-1), p,
                                               None if offs is None else offs.ctypes.data, n if offs is None else 0,
                                               out.reshape(-1), None if cov is None else cov.ctypes.data),
               "hsm_group_match_batch")
        return out, cov


class PoseExchange:
    """hsm_exchange_*: this rank's end of the device-side gather of sharded result rows (capi.h; protocol in
    csrc/pose_exchange.h).  `handle()` is what the other processes need (64 bytes, any channel); `connect(handles)` maps their
    mailboxes; inside ONE process `connect_local(list of PoseExchange in rank order)` uses peer access instead."""

    def __init__(self, rank: int, world: int, total_rows: int, cols: int = 3, depth: int = 4, device: int = -1):
        self._lib = load_library()
        h = _vp()
        _check(self._lib.hsm_exchange_create(int(device), int(rank), int(world), int(total_rows), int(cols), int(depth), C.byref(h)),
               "hsm_exchange_create")
        self._h = h
        self.rank, self.world, self.total_rows, self.cols, self.depth = rank, world, total_rows, cols, depth

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hsm_exchange_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def handle(self) -> bytes:
        buf = C.create_string_buffer(EXCHANGE_HANDLE_BYTES)
        _check(self._lib.hsm_exchange_handle(self._h, C.cast(buf, _vp)), "hsm_exchange_handle")
        return buf.raw

    def connect(self, handles):
        """handles: the world x 64-byte IPC handles in rank order (bytes, or a list of bytes)"""
        blob = b"".join(handles) if not isinstance(handles, (bytes, bytearray)) else bytes(handles)
        if len(blob) != self.world * EXCHANGE_HANDLE_BYTES:
            raise ValueError(f"expected {self.world} handles of {EXCHANGE_HANDLE_BYTES} bytes")
        buf = C.create_string_buffer(blob, len(blob))
        _check(self._lib.hsm_exchange_connect(self._h, C.cast(buf, _vp)), "hsm_exchange_connect")

    def connect_local(self, ranks):
        arr = (_vp * self.world)(*[r._h for r in ranks])
        _check(self._lib.hsm_exchange_connect_local(self._h, arr), "hsm_exchange_connect_local")

    def post(self, d_rows: int, first_row: int, n_rows: int, stream: int = 0):
        _check(self._lib.hsm_exchange_post(self._h, d_rows, int(first_row), int(n_rows), stream), "hsm_exchange_post")

    def wait(self, d_out_all: int, stream: int = 0):
        _check(self._lib.hsm_exchange_wait(self._h, d_out_all, stream), "hsm_exchange_wait")

    def post_wait(self, d_rows: int, first_row: int, n_rows: int, lag: int, d_out_all: int, stream: int = 0):
        _check(self._lib.hsm_exchange_post_wait(self._h, d_rows, int(first_row), int(n_rows), int(lag), d_out_all, stream),
               "hsm_exchange_post_wait")

    def epochs(self):
        p, w = C.c_ulonglong(0), C.c_ulonglong(0)
        _check(self._lib.hsm_exchange_epochs(self._h, C.byref(p), C.byref(w)), "hsm_exchange_epochs")
        return p.value, w.value

    def check(self):
        """raises HsmError if a wait timed out (call after synchronising the stream)"""
        _check(self._lib.hsm_exchange_status(self._h), "hsm_exchange_status")

    def memory_kind(self) -> str:
        return self._lib.hsm_exchange_memory_kind(self._h).decode()


def pose_difference_larger_than(p1, p2, dist_thresh, ang_thresh) -> bool:
    """util::poseDifferenceLargerThan (hector_slam_lib/util/UtilFunctions.h:73-92), fp32."""
    p1 = np.asarray(p1, np.float32)
    p2 = np.asarray(p2, np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        d = p1[:2] - p2[:2]
        if np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1])) > np.float32(dist_thresh):
            return True
        ang = np.float32(p1[2] - p2[2])
        if ang > np.pi:
            ang = np.float32(np.float64(ang) - np.pi * 2.0)
        elif ang < -np.pi:
            ang = np.float32(np.float64(ang) + np.pi * 2.0)
        return bool(abs(ang) > np.float32(ang_thresh))


class HectorSlamProcessor:
    """Glue of HectorSlamProcessor::update (slam_main/HectorSlamProcessor.h:71-124) over the
    GPU map representation: match -> threshold test -> updateByScan -> onMapUpdated."""

    def __init__(self, mapResolution, mapSizeX, mapSizeY, startCoords, multi_res_size, **kw):
        self.mapRep = MapRepMultiMap(mapResolution, mapSizeX, mapSizeY, multi_res_size, startCoords, **kw)
        self.lastScanMatchCov = np.zeros(9, np.float32)
        self.reset()
        self.paramMinDistanceDiffForMapUpdate = np.float32(0.4)
        self.paramMinAngleDiffForMapUpdate = np.float32(0.13)

    def reset(self):
        fmax = np.finfo(np.float32).max
        self.lastMapUpdatePose = np.array([fmax, fmax, fmax], np.float32)
        self.lastScanMatchPose = np.zeros(3, np.float32)
        self.mapRep.reset()

    def setMapUpdateMinDistDiff(self, v): self.paramMinDistanceDiffForMapUpdate = np.float32(v)
    def setMapUpdateMinAngleDiff(self, v): self.paramMinAngleDiffForMapUpdate = np.float32(v)
    def setUpdateFactorFree(self, v): self.mapRep.setUpdateFactorFree(v)
    def setUpdateFactorOccupied(self, v): self.mapRep.setUpdateFactorOccupied(v)
    def getLastScanMatchPose(self): return self.lastScanMatchPose
    def getLastScanMatchCovariance(self): return self.lastScanMatchCov

    def update(self, dataContainer, poseHintWorld, map_without_matching=False, origo=_ZERO2):
        if not map_without_matching:
            new_pose, self.lastScanMatchCov = self.mapRep.matchData(poseHintWorld, dataContainer,
                                                                    self.lastScanMatchCov, origo)
        else:
            new_pose = np.asarray(poseHintWorld, np.float32).copy()
        self.lastScanMatchPose = new_pose
        if pose_difference_larger_than(new_pose, self.lastMapUpdatePose, self.paramMinDistanceDiffForMapUpdate,
                                       self.paramMinAngleDiffForMapUpdate) or map_without_matching:
            self.mapRep.updateByScan(dataContainer, new_pose, origo)
            self.mapRep.onMapUpdated()
            self.lastMapUpdatePose = new_pose.copy()

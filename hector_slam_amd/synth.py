"""Deterministic synthetic worlds, laser scans and trajectories.

The reference ships no bag files or fixtures, so every input used by the tests
and by bench.py is generated here from seeds: an axis-aligned room with box
obstacles, analytic ray casting, Gaussian range noise, and the same range
filtering / range->endpoint conversion the ROS node applies before the scan
reaches the matcher (hector_mapping/src/HectorMappingRos.cpp:483-507,
``rosLaserScanToDataContainer``: keep ``range_min < r < range_max - 0.1``,
endpoint = (cos(a) r s, sin(a) r s) with s = scaleToMap, all fp32).

Pure numpy, no GPU, no oracle: this module only produces inputs.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

# beam fans named in BASELINE.json configs
SCAN_SHAPES = {
    181: (-math.pi / 2, math.pi / 180.0),                    # 180 deg / 1 deg
    1081: (-math.radians(135.0), math.radians(0.25)),        # Hokuyo UTM-30LX 270 deg / 0.25 deg
    16384: (-math.pi, 2.0 * math.pi / 16384.0),              # dense multi-echo / 3D-projected
}


@dataclasses.dataclass
class World:
    """Room + obstacles as a list of wall segments, world frame, metres."""
    segments: np.ndarray  # (M, 4) float64: x0, y0, x1, y1
    width: float
    height: float

    @staticmethod
    def make(width: float = 40.0, height: float = 30.0, n_boxes: int = 12, seed: int = 1234,
             box_min: float = 0.5, box_max: float = 3.0, keep_clear: float = 2.5) -> "World":
        rng = np.random.default_rng(seed)
        hw, hh = width / 2.0, height / 2.0
        segs = [(-hw, -hh, hw, -hh), (hw, -hh, hw, hh), (hw, hh, -hw, hh), (-hw, hh, -hw, -hh)]
        n = 0
        while n < n_boxes:
            bw, bh = rng.uniform(box_min, box_max, size=2) * (width / 40.0)
            cx = rng.uniform(-hw + bw, hw - bw)
            cy = rng.uniform(-hh + bh, hh - bh)
            # keep the trajectory corridor (an ellipse at 55% of the room) free
            if abs(math.hypot(cx / (0.55 * hw), cy / (0.55 * hh)) - 1.0) < keep_clear / min(hw, hh):
                continue
            x0, x1, y0, y1 = cx - bw / 2, cx + bw / 2, cy - bh / 2, cy + bh / 2
            segs += [(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)]
            n += 1
        return World(np.asarray(segs, dtype=np.float64), width, height)

    def raycast(self, pose, angles: np.ndarray) -> np.ndarray:
        """Exact ranges (float64) for beams at robot-frame ``angles`` from ``pose``=(x,y,theta)."""
        x, y, th = (float(v) for v in pose)
        a = angles.astype(np.float64) + th
        dx, dy = np.cos(a)[:, None], np.sin(a)[:, None]
        sx0, sy0, sx1, sy1 = (self.segments[:, i][None, :] for i in range(4))
        ex, ey = sx1 - sx0, sy1 - sy0
        den = dx * ey - dy * ex
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((sx0 - x) * ey - (sy0 - y) * ex) / den
            u = ((sx0 - x) * dy - (sy0 - y) * dx) / den
        ok = (np.abs(den) > 1e-12) & (t > 1e-9) & (u >= 0.0) & (u <= 1.0)
        t = np.where(ok, t, np.inf)
        return t.min(axis=1)


_ANGLE_CACHE: dict = {}


def beam_angles(n_beams: int) -> np.ndarray:
    """the fan's beam angles (cached per fan: every scan of a bench batch shares them; callers do not modify the array)"""
    if n_beams in _ANGLE_CACHE:
        return _ANGLE_CACHE[n_beams]
    a0, inc = SCAN_SHAPES[n_beams] if n_beams in SCAN_SHAPES else (-math.pi, 2.0 * math.pi / n_beams)
    # the node accumulates ``angle += angle_increment`` in fp32 (HectorMappingRos.cpp:491,505)
    out = np.empty(n_beams, dtype=np.float32)
    ang = np.float32(a0)
    inc32 = np.float32(inc)
    for i in range(n_beams):
        out[i] = ang
        ang = np.float32(ang + inc32)
    out.setflags(write=False)
    _ANGLE_CACHE[n_beams] = out
    return out


def scan_to_points(ranges: np.ndarray, angles: np.ndarray, scale_to_map: float,
                   range_min: float = 0.4, range_max: float = 30.0) -> np.ndarray:
    """LaserScan ranges -> DataContainer endpoints, fp32, robot frame, level-0 cell units."""
    r = ranges.astype(np.float32)
    keep = (r > np.float32(range_min)) & (r < np.float32(range_max) - np.float32(0.1))
    r = r[keep] * np.float32(scale_to_map)
    a = angles[keep].astype(np.float32)
    pts = np.empty((r.shape[0], 2), dtype=np.float32)
    pts[:, 0] = np.cos(a).astype(np.float32) * r
    pts[:, 1] = np.sin(a).astype(np.float32) * r
    return pts


def make_scan(world: World, pose, n_beams: int, scale_to_map: float, rng: np.random.Generator | None,
              noise_sigma: float = 0.01, range_max: float = 30.0, pad_to_full: bool = False) -> np.ndarray:
    """One scan taken at ground-truth ``pose``; returns (n_valid, 2) fp32 endpoints.

    With ``pad_to_full`` invalid returns are clamped into range instead of dropped so
    every scan has exactly ``n_beams`` endpoints (fixed-shape batches for the bench).
    """
    ang = beam_angles(n_beams)
    r = world.raycast(pose, ang)
    if rng is not None and noise_sigma > 0:
        r = r + rng.normal(0.0, noise_sigma, size=r.shape)
    if pad_to_full:
        r = np.clip(r, 0.45, range_max - 0.2)
    return scan_to_points(r, ang, scale_to_map, range_max=range_max)


def loop_trajectory(world: World, n_poses: int, frac: float = 0.55, phase: float = 0.0) -> np.ndarray:
    """Elliptic loop inside the room, heading tangent to the path; (n, 3) float64."""
    t = phase + np.linspace(0.0, 2.0 * math.pi, n_poses, endpoint=False)
    x = frac * world.width / 2.0 * np.cos(t)
    y = frac * world.height / 2.0 * np.sin(t)
    th = np.arctan2(frac * world.height / 2.0 * np.cos(t), -frac * world.width / 2.0 * np.sin(t))
    return np.stack([x, y, th], axis=1)


def perturb_poses(poses: np.ndarray, rng: np.random.Generator, d_xy: float = 0.15,
                  d_th: float = 0.05) -> np.ndarray:
    """Initial estimates = truth + U[-d_xy, d_xy] m, U[-d_th, d_th] rad (inside the GN basin)."""
    out = poses.astype(np.float64).copy()
    out[:, :2] += rng.uniform(-d_xy, d_xy, size=(poses.shape[0], 2))
    out[:, 2] += rng.uniform(-d_th, d_th, size=poses.shape[0])
    return out.astype(np.float32)


@dataclasses.dataclass
class Scene:
    """Everything needed to exercise the path: map-building scans and query scans."""
    world: World
    resolution: float
    map_size: int
    levels: int
    n_beams: int
    build_poses: np.ndarray          # (T, 3) float32 ground-truth poses used to build the map
    build_scans: list                # T arrays (n_i, 2) float32
    query_truth: np.ndarray          # (B, 3) float32
    query_init: np.ndarray           # (B, 3) float32 perturbed start estimates
    query_scans: list                # B arrays (n_i, 2) float32

    @property
    def scale_to_map(self) -> float:
        return float(np.float32(1.0) / np.float32(self.resolution))


def make_scene(n_beams: int = 1081, map_size: int = 2048, levels: int = 3, resolution: float = 0.05,
               n_build: int = 200, n_query: int = 64, room=(40.0, 30.0), seed: int = 1234,
               pad_to_full: bool = False, range_max: float = 30.0) -> Scene:
    world = World.make(room[0], room[1], seed=seed)
    s = float(np.float32(1.0) / np.float32(resolution))
    rng_noise = np.random.default_rng(seed + 1)
    rng_init = np.random.default_rng(seed + 2)
    build_poses = loop_trajectory(world, n_build).astype(np.float32)
    build_scans = [make_scan(world, p, n_beams, s, rng_noise, range_max=range_max) for p in build_poses]
    query_truth = loop_trajectory(world, n_query, phase=0.37 * 2 * math.pi / max(n_build, 1)).astype(np.float32)
    query_scans = [make_scan(world, p, n_beams, s, rng_noise, pad_to_full=pad_to_full, range_max=range_max)
                   for p in query_truth]
    query_init = perturb_poses(query_truth, rng_init)
    return Scene(world, resolution, map_size, levels, n_beams, build_poses, build_scans,
                 query_truth, query_init, query_scans)


def pack_scans(scans) -> tuple[np.ndarray, np.ndarray]:
    """CSR packing used by the batched C ABI: (total, 2) fp32 points + (B+1,) int32 offsets."""
    offs = np.zeros(len(scans) + 1, dtype=np.int32)
    for i, s in enumerate(scans):
        offs[i + 1] = offs[i] + s.shape[0]
    pts = np.concatenate(scans, axis=0).astype(np.float32) if scans else np.zeros((0, 2), np.float32)
    return np.ascontiguousarray(pts), offs

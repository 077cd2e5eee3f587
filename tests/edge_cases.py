"""Inputs for the map-edge tests of updateByScan (OccGridMapBase.h:121-260): dense fans whose end points lie on and just
beyond the four map borders, begin cells next to / off every border.  Pure numpy; shared by the CPU pin (restatement ==
reference headers, tests/test_oracle_vs_reference.py) and the GPU tests (tests/test_gpu_dense_edges.py)."""
import numpy as np


def world_pose_of_cell(res, sx, sy, cx, cy, th):
    """world pose whose level-0 map coordinates are (cx, cy): map = world / res + size * 0.5 (GridMapBase.h:265-280)"""
    return np.array([(cx - sx * 0.5) * res, (cy - sy * 0.5) * res, th], np.float32)


def border_fan(rng, sx, sy, cx, cy, th, n):
    """n end points (robot frame, level-0 cell units) of a 360 degree fan from cell (cx, cy): every ray is cut where it meets a
    rectangle that lies d cells outside the map, d drawn per beam from {-3 .. 6} -- so a third of the beams end on / just
    inside the border cells and the rest 0.5 .. 6 cells outside one of the four borders"""
    a = np.linspace(-np.pi, np.pi, n, endpoint=False) + rng.uniform(0, 1e-3)
    d = rng.choice(np.array([-3.0, -1.2, -0.4, 0.6, 1.5, 2.5, 4.0, 6.0]), size=n)
    dx, dy = np.cos(a), np.sin(a)
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(dx > 0, (sx - 1 + d - cx) / dx, np.where(dx < 0, (-d - cx) / dx, np.inf))
        ty = np.where(dy > 0, (sy - 1 + d - cy) / dy, np.where(dy < 0, (-d - cy) / dy, np.inf))
    t = np.minimum(np.where(tx > 0, tx, np.inf), np.where(ty > 0, ty, np.inf))
    t = np.where(np.isfinite(t), t, 3.0)
    # a few short beams and a few that end in the begin cell (skipped by :158)
    short = rng.random(n) < 0.05
    t = np.where(short, rng.uniform(0.0, 8.0, n), t)
    ex, ey = cx + t * dx, cy + t * dy
    c, s = np.cos(-th), np.sin(-th)
    rx, ry = ex - cx, ey - cy
    return np.stack([c * rx - s * ry, s * rx + c * ry], 1).astype(np.float32)


def begin_cells(sx, sy):
    """robot positions (level-0 cells) that matter: centre, next to every border and corner, just off the map on the low side
    (dropped at level 0, valid on coarser levels) and on the high side (dropped everywhere)"""
    return [(sx * 0.5, sy * 0.5), (1.2, sy * 0.4), (sx - 1.6, sy * 0.6), (sx * 0.3, 0.7), (sx * 0.7, sy - 1.4),
            (0.2, 0.3), (sx - 1.2, sy - 1.1), (-1.2, sy * 0.5), (sx * 0.5, -2.4), (-2.0, -2.0), (sx + 0.7, sy * 0.5),
            (3.0, sy - 0.6)]


def probe_coords(sx, sy, rng):
    """map coordinates whose 2x2 footprint runs along the four borders (incl. the last sampled row / column, sx - 2 / sy - 2)"""
    xs = np.arange(0, sx - 1, dtype=np.float64)
    ys = np.arange(0, sy - 1, dtype=np.float64)
    c = [np.stack([xs + rng.uniform(0, 1, xs.size), np.full(xs.size, y) + rng.uniform(0, 1, xs.size)], 1)
         for y in (0.0, 1.0, sy - 3.0, sy - 2.0)]
    c += [np.stack([np.full(ys.size, x) + rng.uniform(0, 1, ys.size), ys + rng.uniform(0, 1, ys.size)], 1)
          for x in (0.0, 1.0, sx - 3.0, sx - 2.0)]
    c = np.concatenate(c).astype(np.float32)
    return np.ascontiguousarray(np.clip(c, 0.0, [sx - 2.0, sy - 2.0]).astype(np.float32))


GEOMETRIES = [  # (sx, sy, levels, beams per scan): widths 64 / 128 / 192 / 256 / 512, heights with sy % 4 in {0, 1, 2, 3}
    (512, 512, 3, 6000),
    (256, 250, 3, 4096),   # 250 / 125 / 62 rows: every level's last block row straddles the map's end
    (128, 127, 2, 16384),  # 128 / 64 columns, 127 / 63 rows
    (192, 190, 1, 4100),   # dense form on a 192-column level
    (192, 190, 2, 4100),   # level 1 is 96 columns wide: the whole batch takes the keyed form by itself
    (64, 50, 1, 4096),     # one block column
    (64, 67, 1, 5000),
]

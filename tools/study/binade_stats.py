#!/usr/bin/env python
"""Feasibility statistics for the binade-speculative parallel form of the reference's fp32 chains (round-4 verdict item 1(ii)):
along the nine sequential sums of getCompleteHessianDerivs over a dense scan, how often does the running sum change binade
(exponent), how are those changes distributed over 64-beam rows, and how many round-to-even ties occur -- CPU only, numpy,
on the oracle's map and per-beam terms."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from hector_slam_amd import synth
from oracle import pyoracle

def products(o, level, pose_map, pts):
    f = np.float32
    ex, ey, th = (f(v) for v in pose_map)
    s, c = f(math.sin(th)), f(math.cos(th))
    px, py = pts[:, 0].astype(f), pts[:, 1].astype(f)
    rx = c * px - s * py
    ry = s * px + c * py
    coords = np.stack([ex + rx, ey + ry], 1).astype(f)
    mg = o.interp(level, coords)  # (M, gx, gy)
    M, gx, gy = mg[:, 0], mg[:, 1], mg[:, 2]
    fun = f(1.0) - M
    rd = (-ry) * gx + rx * gy
    return np.stack([gx * fun, gy * fun, rd * fun, gx * gx, gy * gy, rd * rd, gx * gy, gx * rd, gy * rd], 1).astype(f)

def chain_stats(x, row=64):
    n = len(x)
    s = np.float32(0.0)
    exps = np.empty(n, np.int32)
    ties = 0
    for i in range(n):
        t = np.float64(s) + np.float64(x[i])
        s2 = np.float32(t)
        if s2 != t:
            # tie: exact sum is halfway between two floats
            lo = np.nextafter(s2, np.float32(-np.inf)); hi = np.nextafter(s2, np.float32(np.inf))
            if abs(t - np.float64(s2)) == abs(np.float64(lo) - np.float64(s2)) / 2 or abs(t - np.float64(s2)) == abs(np.float64(hi) - np.float64(s2)) / 2:
                ties += 1
        s = s2
        exps[i] = -1000 if s == 0 else int(math.frexp(float(s))[1]) + (10000 if s < 0 else 0)
    ch = np.flatnonzero(exps[1:] != exps[:-1]) + 1
    rows = np.bincount(ch // row, minlength=(n + row - 1) // row)
    return {"changes": int(len(ch)), "rows_with_change": int((rows > 0).sum()), "rows_gt2": int((rows > 2).sum()), "ties": ties, "final": float(s)}

def main():
    n_beams = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    size = 2048
    sc = synth.make_scene(n_beams=n_beams, map_size=size, levels=1, resolution=0.05, n_build=30, n_query=3, room=(40.0, 30.0), seed=31)
    pyoracle.build()
    o = pyoracle.Oracle("ho", sc.resolution, size, size, 1)
    o.set_update_factor_free(0.4); o.set_update_factor_occupied(0.9)
    o.build_map(sc.build_poses, sc.build_scans)
    names = ["dTr0", "dTr1", "dTr2", "H00", "H11", "H22", "H01", "H02", "H12"]
    for q in range(2):
        pts = sc.query_scans[q]
        for label, pose_w in (("start", sc.query_init[q]), ("converged", o.match(sc.query_init[q], pts)[0])):
            pm = o.map_coords_pose(0, pose_w)
            pr = products(o, 0, pm, pts)
            if "--dump" in sys.argv:
                np.ascontiguousarray(pr.T).tofile(f"/tmp/products_{n_beams}_{q}_{label}.bin")
                print("dumped", pr.shape)
                continue
            nrows = (len(pts) + 63) // 64
            tot_rows = np.zeros(nrows, bool)
            print(f"scan {q} ({len(pts)} beams, {nrows} rows) at {label} pose")
            allrows = set()
            for k, nm in enumerate(names):
                st = chain_stats(pr[:, k])
                print(f"  {nm}: exponent changes {st['changes']:5d}  rows with a change {st['rows_with_change']:4d}  rows with > 2 {st['rows_gt2']:4d}  ties {st['ties']:5d}  final {st['final']:.4g}")

if __name__ == "__main__":
    main()

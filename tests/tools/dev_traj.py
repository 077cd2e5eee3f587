#!/usr/bin/env python
"""debug: per-step deviation GPU vs oracle in the config5 loop"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hector_slam_amd import synth, capi
from oracle import pyoracle
pyoracle.build()
beams, size, res, levels = 16384, 8192, 0.05, 3
roomw = float(sys.argv[2]) if len(sys.argv) > 2 else 320.0
room, rmax = (roomw, roomw * 0.75), roomw * 0.75
T = 20
world = synth.World.make(room[0], room[1], seed=1234)
rng_noise = np.random.default_rng(1235)
sfac = float(np.float32(1.0) / np.float32(res))
poses = synth.loop_trajectory(world, int(sys.argv[3]) if len(sys.argv) > 3 else 1400)[: T + 1].astype(np.float32)
scans = [synth.make_scan(world, p, beams, sfac, rng_noise, range_max=rmax) for p in poses]
m = capi.MapRepMultiMap(res, size, size, levels)
m.setUpdateFactorFree(0.4); m.setUpdateFactorOccupied(0.9)
o = pyoracle.Oracle("ho", res, size, size, levels)
o.set_update_factor_free(0.4); o.set_update_factor_occupied(0.9)
NI = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for k in range(NI + 1):
    m.matchData(poses[k], scans[k]); m.updateByScan(scans[k], poses[k])
    o.match(poses[k], scans[k]); o.update_by_scan(poses[k], scans[k]); o.on_map_updated()
pg = po = poses[NI]
for t in range(NI + 1, T + 1):
    for lvl in range(levels):
        a, b = m.download_level(lvl), o.download_level(lvl)
        nd = (a[0].view(np.uint32) != b[0].view(np.uint32)).sum()
        if nd: print("  map differs before step", t, "level", lvl, nd)
    d = poses[t] - poses[t - 1]
    pg, cg = m.matchData(pg + d, scans[t])
    po, co = o.match(po + d, scans[t])
    print(t, "gpu-truth", pg - poses[t], "cpu-truth", po - poses[t], "gpu-cpu", pg - po)
    for lvl in range(levels - 1, -1, -1):
        f = np.float32(1.0 / 2 ** lvl)
        hint = (po + d) if lvl == levels - 1 else hint
    # per-level conditioning of the reference at its own start estimate
    for lvl in range(levels):
        f = np.float32(1.0 / 2 ** lvl)
        H, _ = o.hessian_derivs(lvl, o.map_coords_pose(lvl, po), scans[t] * f)
        print("     lvl", lvl, "cond(H) at cpu result = %.3g" % np.linalg.cond(H.astype(np.float64)))
    m.updateByScan(scans[t], pg); o.update_by_scan(po, scans[t]); o.on_map_updated()

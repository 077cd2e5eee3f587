#!/usr/bin/env python
"""debug: one GN evaluation / step per level, GPU vs oracle, dense scan"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hector_slam_amd import synth, capi
from oracle import pyoracle
pyoracle.build()
beams, size, res, levels = 16384, 8192, 0.05, 3
room = (160.0, 120.0); rmax = 120.0
world = synth.World.make(room[0], room[1], seed=1234)
rng_noise = np.random.default_rng(1235)
sfac = float(np.float32(1.0) / np.float32(res))
T = 10
poses = synth.loop_trajectory(world, 700)[: T + 1].astype(np.float32)
scans = [synth.make_scan(world, p, beams, sfac, rng_noise, range_max=rmax) for p in poses]
m = capi.MapRepMultiMap(res, size, size, levels)
m.setUpdateFactorFree(0.4); m.setUpdateFactorOccupied(0.9)
o = pyoracle.Oracle("ho", res, size, size, levels)
o.set_update_factor_free(0.4); o.set_update_factor_occupied(0.9)
for k in range(9):
    m.matchData(poses[k], scans[k]); m.updateByScan(scans[k], poses[k])
    o.match(poses[k], scans[k]); o.update_by_scan(poses[k], scans[k]); o.on_map_updated()
for lvl in range(levels):
    a, b = m.download_level(lvl), o.download_level(lvl)
    print("level", lvl, "map diff cells", (a[0].view(np.uint32) != b[0].view(np.uint32)).sum())
for lvl in range(levels):
    lo, _ = m.download_level(lvl)
    pr = m.download_prob(lvl)
    odds = np.exp(lo.astype(np.float64)).astype(np.float32)
    exp_p = odds / (odds + np.float32(1.0))
    bad = pr.view(np.uint32) != exp_p.view(np.uint32)
    print("level", lvl, "prob plane mismatches", bad.sum(), "of touched", (lo != 0).sum())
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("   bad bbox x", xs.min(), xs.max(), "y", ys.min(), ys.max(), "bbox", m.last_update_bbox(lvl))
t = 9
hint = poses[t]
pts0 = scans[t]
pm0 = o.map_coords_pose(0, hint)
got = m.eval_beams(0, pm0, pts0)
s_, c_ = np.float32(np.sin(np.float64(pm0[2]))), np.float32(np.cos(np.float64(pm0[2])))
tx = pm0[0] + (c_ * pts0[:, 0] + (-s_) * pts0[:, 1]); ty = pm0[1] + (s_ * pts0[:, 0] + c_ * pts0[:, 1])
ref = o.interp(0, np.stack([tx, ty], 1).astype(np.float32))
badb = (got[:, :3].view(np.uint32) != ref.view(np.uint32)).any(1)
print("per-beam mismatches level 0:", badb.sum(), "of", pts0.shape[0], "first bad idx", np.nonzero(badb)[0][:10])
if badb.any():
    i = np.nonzero(badb)[0][0]
    print("   beam", i, "coords", tx[i], ty[i], "gpu", got[i], "ref", ref[i])
for lvl in range(levels - 1, -1, -1):
    f = np.float32(1.0 / 2 ** lvl)
    pts = scans[t] * f
    pm = o.map_coords_pose(lvl, hint)
    Hg, dg = m.hessian_derivs(lvl, pm, pts)
    Ho, do = o.hessian_derivs(lvl, pm, pts)
    print("lvl", lvl, "H rel dev", np.abs(Hg - Ho).max() / np.abs(Ho).max(), "dTr", dg, do)
    for it in (0, 1, 3, 5):
        pg, _ = m.match_level(lvl, hint, pts, it)
        po, _ = o.match_level(lvl, hint, pts, it)
        print("   iters", it + 1, "gpu", pg - poses[t], "cpu", po - poses[t])
pg, _ = m.matchData(hint, scans[t]); po, _ = o.match(hint, scans[t])
print("full gpu", pg - poses[t], "cpu", po - poses[t], m.last_launch_config())
for n in (1081, 4096, 8192, 12000):
    sub = scans[t][np.linspace(0, scans[t].shape[0] - 1, n).astype(int)]
    pg, _ = m.matchData(hint, sub); po, _ = o.match(hint, sub)
    print("n", n, "gpu", pg - poses[t], "cpu", po - poses[t], m.last_launch_config())

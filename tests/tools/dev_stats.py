#!/usr/bin/env python
"""Deviation statistics GPU vs oracle on a config (debug tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hector_slam_amd import synth, capi
from oracle import pyoracle
pyoracle.build()
res = float(sys.argv[1]); size = int(sys.argv[2]); nq = int(sys.argv[3])
room = float(sys.argv[4]) if len(sys.argv) > 4 else 40.0
rmax = float(sys.argv[5]) if len(sys.argv) > 5 else 30.0
sc = synth.make_scene(n_beams=1081, map_size=size, levels=3, resolution=res, n_build=100, n_query=nq, room=(room, room * 0.75), seed=77, range_max=rmax)
g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9); g.build_map(sc.build_poses, sc.build_scans)
o = pyoracle.Oracle("ho", sc.resolution, sc.map_size, sc.map_size, sc.levels)
o.set_update_factor_free(0.4); o.set_update_factor_occupied(0.9); o.build_map(sc.build_poses, sc.build_scans)
pts, offs = synth.pack_scans(sc.query_scans)
pose, cov = g.match_batch(sc.query_init, pts, offs)
po = o.match_many(sc.query_init, pts, offs)
d = np.abs(pose.astype(np.float64) - po)
dxy = d[:, :2].max(1)
print("frac within 1e-4:", (dxy <= 1e-4).mean(), "median", np.median(dxy), "p90", np.percentile(dxy, 90), "p99", np.percentile(dxy, 99), "max", dxy.max())
# stability of the reference itself: continue from its own result
po2 = o.match_many(po, pts, offs)
mv = np.abs(po2.astype(np.float64) - po)[:, :2].max(1)
print("oracle self-move: median", np.median(mv), "p90", np.percentile(mv, 90), "max", mv.max())
bad = dxy > 1e-4
print("bad:", bad.sum(), "self-move of bad: median", np.median(mv[bad]) if bad.any() else None, "self-move of good: median", np.median(mv[~bad]))
dth = np.abs(pose[:, 2].astype(np.float64) - po[:, 2])
print("dtheta: frac within 1e-4", (dth <= 1e-4).mean(), "max", dth.max())
dc = np.abs(cov - np.array([o.match(sc.query_init[q], sc.query_scans[q])[1] for q in range(32)])).max(1) / np.abs(cov[:32]).max(1)
print("cov rel dev first 32:", np.sort(dc)[-5:])
err = np.abs(po.astype(np.float64) - sc.query_truth)[:, :2].max(1)
print("oracle err vs truth median", np.median(err), "p99", np.percentile(err, 99))

#!/usr/bin/env python
"""Long-run soak (one-off evidence, not part of the suite): N match+update steps around the room, the same poses fed
to the GPU path and to the oracle, so the maps must stay BIT-IDENTICAL throughout -- through the natural wrap of the
16-bit key generation (65535 updates) and the saturation of the log-odds clamp.  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from hector_slam_amd import synth, capi
from oracle import pyoracle
pyoracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
sc = synth.make_scene(n_beams=1081, map_size=1024, levels=3, resolution=0.05, n_build=400, n_query=4, room=(40.0, 30.0), seed=5)
g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
o = pyoracle.Oracle("ho", sc.resolution, sc.map_size, sc.map_size, sc.levels)
for m in (g.setUpdateFactorFree, o.set_update_factor_free): m(0.4)
for m in (g.setUpdateFactorOccupied, o.set_update_factor_occupied): m(0.9)
worst = 0.0; ident = 0; tg = tc = 0.0
for t in range(N):
    k = t % 400
    a = time.perf_counter()
    pg, _ = g.matchData(sc.build_poses[k], sc.build_scans[k]); g.updateByScan(sc.build_scans[k], sc.build_poses[k])
    b = time.perf_counter()
    po, _ = o.match(sc.build_poses[k], sc.build_scans[k]); o.update_by_scan(sc.build_poses[k], sc.build_scans[k]); o.on_map_updated()
    c = time.perf_counter()
    tg += b - a; tc += c - b
    worst = max(worst, float(np.abs(pg.astype(np.float64) - po)[:2].max()))
    ident += int(np.array_equal(pg.view(np.uint32), po.view(np.uint32)))
maps_equal = True; sat = 0
for lvl in range(sc.levels):
    a_, b_ = g.download_level(lvl), o.download_level(lvl)
    maps_equal &= bool(np.array_equal(a_[0].view(np.uint32), b_[0].view(np.uint32)) and np.array_equal(a_[1], b_[1]))
    sat += int((b_[0] >= 50.0).sum())
print(json.dumps({"steps": N, "maps_bit_identical": maps_equal, "cells_at_clamp": sat, "worst_pose_dev_m": worst,
                  "bit_identical_pose_fraction": ident / N, "gpu_ms_per_step": tg / N * 1e3, "cpu_ms_per_step": tc / N * 1e3}))

#!/usr/bin/env python
"""host-call time of ONE scan through hsm_match in the default (reference-order) mode, by waves per scan.
usage: tools/study/single_scan_wps_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hector_slam_amd import capi, synth
for n_beams, size in ((1081, 2048), (720, 1024), (360, 1024)):
    sc = synth.make_scene(n_beams=n_beams, map_size=size, levels=3, resolution=0.05, n_build=20, n_query=2, room=(40.0, 30.0), seed=31)
    ref = None
    for wps in (0, 1, 2, 4, 8, 16):
        kw = {"waves_per_scan": wps} if wps else {}
        g = capi.MapRepMultiMap(sc.resolution, size, size, 3, **kw)
        g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
        g.build_map(sc.build_poses, sc.build_scans)
        for k in range(50):
            p, c = g.matchData(sc.query_init[0], sc.query_scans[0])
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for k in range(200):
                p, c = g.matchData(sc.query_init[0], sc.query_scans[0])
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        bits = np.asarray(p, np.float32).view(np.uint32).tolist() + np.asarray(c, np.float32).view(np.uint32).ravel().tolist()
        ref = ref or bits
        cfg = g.last_launch_config()
        print(n_beams, len(sc.query_scans[0]), "waves_per_scan", wps or "default", cfg["kernel"], cfg["block"], "host call us: median %.1f min %.1f" % (sorted(ts)[2], min(ts)),
              "same bits" if bits == ref else "DIFFERENT BITS", flush=True)
        g.close()

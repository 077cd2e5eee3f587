#!/usr/bin/env python
"""s_memtime stamps of the reference-order single-scan matcher's phases (gn_match_kernel<.., EXACT>, -DHSM_TEAM_TIMELINE variant):
per GN step {pose -> products staged, chain, totals -> solved}, summed over the steps of a match.
usage: HSM_LIB=.../libhector_mi355_teamtl.so tools/study/team_phase_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hector_slam_amd import capi, synth
for n_beams, size in ((1081, 2048), (1081, 1024), (360, 1024)):
    sc = synth.make_scene(n_beams=n_beams, map_size=size, levels=3, resolution=0.05, n_build=20, n_query=2, room=(40.0, 30.0), seed=31)
    g = capi.MapRepMultiMap(sc.resolution, size, size, 3)
    g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    probe = torch.zeros(64, dtype=torch.int64, device="cuda")
    g.set_clock_probe(probe.data_ptr())
    for k in range(20):
        g.matchData(sc.query_init[0], sc.query_scans[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200):
        g.matchData(sc.query_init[0], sc.query_scans[0])
    dt = (time.perf_counter() - t0) / 200
    t = probe.cpu().numpy()
    print(n_beams, size, g.last_launch_config()["kernel"], g.last_launch_config().get("block"), "host call %.1f us" % (dt * 1e6), "steps", int(t[11]),
          "cycles summed: pose->staged %d, chain %d, totals->solved %d" % (t[8], t[9], t[10]),
          "first step start -> last step end: %d cycles, %.2f us, clock %.3f GHz" % (t[14] - t[12], (t[15] - t[13]) / 100.0, (t[14] - t[12]) / max(1, t[15] - t[13]) * 0.1),
          "\n   kernel entry -> first step %.2f us, last step end -> exit stamp %.2f us, entry -> exit %.2f us" % ((t[13] - t[4]) / 100.0, (t[5] - t[15]) / 100.0, (t[5] - t[4]) / 100.0),
          "\n   pose->staged per step:", [int(x) for x in t[16:16 + int(t[11])]], "\n   chain:", [int(x) for x in t[32:32 + int(t[11])]],
          "\n   totals->solved (+ level change):", [int(x) for x in t[48:48 + int(t[11])]])

#!/usr/bin/env python
"""PCIe-inclusive rate of the host-buffer batch entry (hsm_match_batch): B=4096 x 1081-beam scans from pageable host
memory, results back to the host -- the figure DESIGN.md quotes next to the HBM-resident headline."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from hector_slam_amd import capi
bp, bs, truth, init_l0, init_pyr, pts, offs = bench.make_inputs(0, 4096)[:7]
m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, 1)
m.setUpdateFactorFree(0.4); m.setUpdateFactorOccupied(0.9)
m.build_map(bp, bs)
m.match_batch(init_l0, pts, offs)
t = []
for _ in range(15):
    a = time.perf_counter(); pose, cov = m.match_batch(init_l0, pts, offs); t.append(time.perf_counter() - a)
med = float(np.median(t))
print({"host_call_ms": med * 1e3, "GN_it_per_s": 4096 * 6 / med, "bytes_in_MB": pts.nbytes / 1e6, "effective_GBps": pts.nbytes / med / 1e9})

#!/usr/bin/env python
"""start / end (100 MHz wall clock) of EVERY workgroup of the headline launch (-DHSM_XTIMELINE_WG variant build): how long a
workgroup lives, when the first and the last one end, per XCD.  usage: HSM_LIB=<variant.so> tools/study/exact_wg_times.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from hector_slam_amd import capi
B = 4096
build_poses, build_scans, truth, init, init_pyr, pts, offs, _ = bench.make_inputs(0, B)
m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, 1)
m.setUpdateFactorFree(0.4); m.setUpdateFactorOccupied(0.9)
m.build_map(build_poses, build_scans)
dev = torch.device("cuda", 0)
d_init, d_pts, d_offs = (torch.from_numpy(x).to(dev) for x in (init, pts, offs))
d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
probe = torch.zeros(1024 + 4 * 1024 + 8, dtype=torch.int64, device=dev)
m.set_clock_probe(probe.data_ptr())
for rep in range(30):
    m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, d_pose.data_ptr(), 0, 0)
torch.cuda.synchronize()
t = probe.cpu().numpy()[1024:1024 + 4096].reshape(1024, 4).astype(np.int64)
st, en, idw = t[:, 0], t[:, 1], t[:, 2]
xcc = (idw >> 32) & 0xf
t0 = st.min()
life = (en - st) * 10.0 / 1e3  # us
print(m.last_launch_config().get("kernel"))
print("starts: first..last %.2f us; ends: first %.2f, median %.2f, last %.2f us after the first start" % ((st.max() - t0) * 0.01, (en.min() - t0) * 0.01, (np.median(en) - t0) * 0.01, (en.max() - t0) * 0.01))
print("lifetime of a workgroup: min %.2f median %.2f p90 %.2f max %.2f us" % (life.min(), np.median(life), np.percentile(life, 90), life.max()))
for x in range(8):
    s = xcc == x
    if s.any():
        print("XCC %d: %4d workgroups, blocks %4d..%4d, end median %.2f max %.2f, lifetime median %.2f" % (x, s.sum(), np.flatnonzero(s).min(), np.flatnonzero(s).max(), (np.median(en[s]) - t0) * 0.01, (en[s].max() - t0) * 0.01, np.median(life[s])))
order = np.argsort(en)
print("last 8 workgroups to end:", [(int(b), int(xcc[b]), round((en[b] - t0) * 0.01, 2), round(life[b], 2)) for b in order[-8:]])

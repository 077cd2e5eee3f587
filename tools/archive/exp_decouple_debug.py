#!/usr/bin/env python
"""Where the wavefronts of the decoupled exact-order kernel wait (variant built with -DHSM_EXPERIMENTS -DHSM_XDEBUG
-DHSM_XDECOUPLE=1: the covariance output carries shader-clock cycle counts).  HSM_LIB=<variant> python tools/exp_decouple_debug.py"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch
from hector_slam_amd import capi

B = 4096
build_poses, build_scans, truth, init, init_pyr, pts, offs = bench.make_inputs(0, B)
m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, 1)
m.setUpdateFactorFree(0.4); m.setUpdateFactorOccupied(0.9)
m.build_map(build_poses, build_scans)
dev = torch.device("cuda", 0)
d_init = torch.from_numpy(init).to(dev); d_pts = torch.from_numpy(pts).to(dev); d_offs = torch.from_numpy(offs).to(dev)
d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev); d_cov = torch.zeros((B, 9), dtype=torch.float32, device=dev)
for _ in range(300):
    m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, d_pose.data_ptr(), d_cov.data_ptr(), 0)
m.synchronize()
c = d_cov.cpu().numpy()
names = ["total", "buffer_wait", "job_ready_wait", "end_of_step_wait", "in_jobs"]
out = {n: {"mean": float(c[:, i].mean()), "p10": float(np.percentile(c[:, i], 10)), "p90": float(np.percentile(c[:, i], 90)), "max": float(c[:, i].max())} for i, n in enumerate(names)}
sel = np.r_[0:100, 1024:B]
for i, n in enumerate(names):
    out[n] = {"mean": float(c[sel, i].mean()), "p10": float(np.percentile(c[sel, i], 10)), "p90": float(np.percentile(c[sel, i], 90)), "max": float(c[sel, i].max())}
out["config"] = m.last_launch_config()
flat = c.reshape(-1)
tl = {}
for d in range(4):
    t = flat[1024 * (d + 1):1024 * (d + 2)]
    tl[f"wg{256 * d}"] = {"rows": [[int(x) for x in t[w * 102:(w + 1) * 102]] for w in range(4)],
                           "job_arrive": [int(x) for x in t[408:510]], "job_start": [int(x) for x in t[510:612]], "job_end": [int(x) for x in t[612:714]],
                           "t0_low24": [int(x) for x in t[800:804]]}
if len(sys.argv) > 1:
    json.dump(tl, open(sys.argv[1], "w"))
print(json.dumps(out))

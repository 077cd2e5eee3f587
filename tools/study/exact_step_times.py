#!/usr/bin/env python
"""cycles per GN step of workgroup 0's first wavefront in the headline kernel (-DHSM_XTIMELINE_STEPS variant build):
usage: HSM_LIB=<variant.so> tools/study/exact_step_times.py"""
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
probe = torch.zeros(4 * 31 * 4 + 40, dtype=torch.int64, device=dev)
m.set_clock_probe(probe.data_ptr())
acc = []
for rep in range(40):
    m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, d_pose.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    t = probe.cpu().numpy()[4 * 31 * 4: 4 * 31 * 4 + 7].astype(np.int64)
    if rep >= 10:
        acc.append(np.diff(t))
a = np.array(acc)
print(m.last_launch_config().get("kernel"), "cycles per GN step (median of 30 launches):", [int(x) for x in np.median(a, 0)], "sum", int(np.median(a.sum(1))),
      "per round (17) in steps 3-6:", [int(x / 17) for x in np.median(a, 0)[2:]])

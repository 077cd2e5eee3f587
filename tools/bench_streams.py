#!/usr/bin/env python
"""Headline workload issued round-robin on S caller-owned streams (independent batches overlap: the tail of one launch
-- waves that finish early leave their slots empty -- is filled by the next launch).  Prints one JSON line per S."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import torch
    from hector_slam_amd import capi
    dev = torch.device("cuda", 0)
    B = 4096
    bp, bs, truth, init, init_pyr, pts, offs = bench.make_inputs(0, B)[:7]
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, levels, device=0)
    m.setUpdateFactorFree(0.4)
    m.setUpdateFactorOccupied(0.9)
    m.build_map(bp, bs)
    d_init = torch.from_numpy(init if levels == 1 else init_pyr).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    its = m.gn_iterations_per_match()
    steps = 240
    ref = None
    for S in (1, 2, 3, 4):
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        poses = [torch.zeros((B, 3), dtype=torch.float32, device=dev) for _ in range(S)]
        covs = [torch.zeros((B, 9), dtype=torch.float32, device=dev) for _ in range(S)]

        def step(k):
            s = k % S
            m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, poses[s].data_ptr(),
                                 covs[s].data_ptr(), streams[s].cuda_stream)
        for k in range(12):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p = poses[0].cpu().numpy()
        if ref is None:
            ref = p
        print(json.dumps({"streams": S, "levels": levels, "us_per_step": dt / steps * 1e6, "Mit_per_s": B * its * steps / dt / 1e6,
                          "bit_identical_to_single_stream": bool((p.view(np.uint32) == ref.view(np.uint32)).all())}))


if __name__ == "__main__":
    main()

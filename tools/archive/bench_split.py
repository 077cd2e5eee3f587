#!/usr/bin/env python
"""One 4096-scan batch split into Q sub-batches launched on Q streams, joined before the next batch starts (no overlap
BETWEEN batches): does staggering inside one batch recover the tail?  Prints one JSON line per Q."""
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
    bp, bs, truth, init, init_pyr, pts, offs = bench.make_inputs(0, B)
    m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, 1, device=0)
    m.setUpdateFactorFree(0.4)
    m.setUpdateFactorOccupied(0.9)
    m.build_map(bp, bs)
    d_init = torch.from_numpy(init).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    its = m.gn_iterations_per_match()
    steps = 200
    for Q in (1, 2, 4, 8):
        streams = [torch.cuda.Stream(device=dev) for _ in range(Q)]
        pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        sub = B // Q
        d_offs = [torch.from_numpy((offs[q * sub:(q + 1) * sub + 1] - offs[q * sub]).astype(np.int32)).to(dev) for q in range(Q)]
        join = [torch.cuda.Event() for _ in range(Q)]
        start = torch.cuda.Event()

        def step():
            start.record(streams[0])
            for q in range(Q):
                if q:
                    streams[q].wait_event(start)
                m.match_batch_device(sub, d_init.data_ptr() + q * sub * 12, d_pts.data_ptr() + int(offs[q * sub]) * 8,
                                     d_offs[q].data_ptr(), bench.N_BEAMS, pose.data_ptr() + q * sub * 12, 0, streams[q].cuda_stream)
                if q:
                    join[q].record(streams[q])
            for q in range(1, Q):
                streams[0].wait_event(join[q])
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"sub_batches": Q, "us_per_batch": dt / steps * 1e6, "Mit_per_s": B * its * steps / dt / 1e6,
                          "cfg": m.last_launch_config()}))


if __name__ == "__main__":
    main()

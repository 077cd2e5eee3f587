#!/usr/bin/env python
"""Does the order of the scans in a batch matter?  The bench batches follow the trajectory (neighbours in the batch are neighbours
in the map); here the same batch runs in that order and randomly permuted: default (reference-order) mode and fast mode, on the
2048^2 headline map and the 4096^2 pyramid.  usage: tools/study/batch_order_locality.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hector_slam_amd import capi, synth
B, N = 4096, 1081
dev = torch.device("cuda", 0)
s = torch.cuda.current_stream().cuda_stream
for name, kw in (("2048^2, 1 level", dict(map_size=2048, levels=1, room=(40.0, 30.0))),
                 ("4096^2 pyramid", dict(map_size=4096, levels=3, room=(160.0, 120.0), range_max=120.0))):
    sc = synth.make_scene(n_beams=N, resolution=0.05, n_build=100, n_query=B, seed=77, pad_to_full=True, **kw)
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, device=0)
    g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    rng = np.random.default_rng(1)
    res = {}
    rp = rng.permutation(B)
    for order_name, perm, order in (("trajectory order", np.arange(B), capi.ORDER_GIVEN), ("random order", rp, capi.ORDER_GIVEN),
                                    ("random order, HSM_ORDER_MORTON", rp, capi.ORDER_MORTON), ("trajectory order, HSM_ORDER_MORTON", np.arange(B), capi.ORDER_MORTON),
                                    ("random order, HSM_ORDER_AUTO (the default)", rp, capi.ORDER_AUTO), ("trajectory order, HSM_ORDER_AUTO (the default)", np.arange(B), capi.ORDER_AUTO),
                                    ("trajectory order again", np.arange(B), capi.ORDER_GIVEN)):
        g.set_batch_order(order)
        scans = [sc.query_scans[i] for i in perm]
        pts, offs = synth.pack_scans(scans)
        d_pts, d_offs, d_init = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(sc.query_init[perm]).to(dev)
        d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        for mode in (capi.PARITY_AUTO, capi.PARITY_FAST):
            g.set_parity(mode)
            f = lambda: g.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, d_pose.data_ptr(), 0, s)
            for _ in range(600): f()
            torch.cuda.synchronize()
            ts = []
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(200): f()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 200 * 1e6)
            pose = d_pose.cpu().numpy()
            back = np.empty_like(pose); back[perm] = pose
            key = (mode,)
            same = "" if key not in res else (" poses bit-identical to the first order" if np.array_equal(res[key].view(np.uint32), back.view(np.uint32)) else " POSES DIFFER")
            res.setdefault(key, back)
            print(name, "|", order_name, "|", "default" if mode == capi.PARITY_AUTO else "fast", "| %.1f us per launch" % sorted(ts)[1], g.last_launch_config()["kernel"], "(sorted)" if g.last_launch_sorted() else "", same, flush=True)
    g.close()

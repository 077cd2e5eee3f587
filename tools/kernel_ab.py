#!/usr/bin/env python
"""Within-process interleaved A/B of gn_match_kernel variants on the bench workload
(B=4096 x 1081 beams, 2048^2 map).  Variants = layout x waves-per-scan x register-resident on/off.
Prints one JSON line per variant (median / min kernel ms over the rounds) to stdout."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--levels", type=int, default=1)
    ap.add_argument("--variants", default="quad:1:1,quad:1:0,plane:1:1,plane:1:0,quad:2:1,quad:4:1,quad:2:0")
    args = ap.parse_args()
    import torch
    from hector_slam_amd import capi
    dev = torch.device("cuda", 0)
    B = args.batch
    bp, bs, truth, init_l0, init_pyr, pts, offs = bench.make_inputs(0, B)[:7]
    init = init_l0 if args.levels == 1 else init_pyr
    d_init = torch.from_numpy(init).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    stream = torch.cuda.current_stream()
    variants = []
    for spec in args.variants.split(","):
        layout, wps, regs = spec.split(":")[:3]
        os.environ["HSM_BPL"] = regs
        os.environ["HSM_EXPERIMENT"] = spec.split(":")[3] if spec.count(":") >= 3 else "0"
        m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, args.levels, device=0,
                                layout=capi.LAYOUT_QUAD if layout == "quad" else capi.LAYOUT_PLANE,
                                waves_per_scan=int(wps))
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
        m.build_map(bp, bs)
        pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        cov = torch.zeros((B, 9), dtype=torch.float32, device=dev)
        variants.append({"spec": spec, "exp": os.environ["HSM_EXPERIMENT"], "m": m, "pose": pose, "cov": cov, "ms": []})
    os.environ.pop("HSM_BPL", None)

    def launch(v):
        os.environ["HSM_EXPERIMENT"] = v["exp"]
        v["m"].match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS,
                                  v["pose"].data_ptr(), v["cov"].data_ptr(), stream.cuda_stream)

    for v in variants:  # warm
        launch(v)
        launch(v)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for v in variants:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            launch(v)
            b.record(stream)
            torch.cuda.synchronize()
            v["ms"].append(a.elapsed_time(b))
    ref = variants[0]["pose"].cpu().numpy()
    its = variants[0]["m"].gn_iterations_per_match()
    for v in variants:
        ms = np.array(v["ms"])
        p = v["pose"].cpu().numpy()
        print(json.dumps({"variant": v["spec"], "cfg": v["m"].last_launch_config(),
                          "ms_median": float(np.median(ms)), "ms_min": float(ms.min()),
                          "Mit_per_s_median": B * its / float(np.median(ms)) / 1e3,
                          "max_dev_vs_first": float(np.abs(p - ref).max()),
                          "med_err_vs_truth": float(np.median(np.abs(p[:, :2] - truth[:, :2])))}))


if __name__ == "__main__":
    main()

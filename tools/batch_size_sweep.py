#!/usr/bin/env python
"""Default-mode (reference-order) batched matcher over batch sizes: which launch shape serves which batch?  One map per variant
(HSM_* launch knobs are read when a context is created), the first b scans of the bench batch for every b, kernel time from
events around 20 launches, and the poses of every variant compared bit for bit with the first variant's.
  python tools/batch_size_sweep.py --levels 1 --variants "default;HSM_WPS=1;HSM_WPS=1,HSM_EXACT_CHAIN_WAVE=1" """
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", type=int, default=1)
    ap.add_argument("--sizes", default="16,64,128,256,512,768,1024,1536,2048,3072,3584,4096")
    ap.add_argument("--variants", default="default;HSM_WPS=1;HSM_WPS=1,HSM_EXACT_CHAIN_WAVE=1")
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--workload", default="config3", help="config3 (the headline scene) or config4 (the 4096^2 pyramid's scene)")
    ap.add_argument("--thin", type=int, default=1, help="every k-th beam of every scan only: the 13- / 9- / 5-row instantiations")
    ap.add_argument("--stretch", type=int, default=1, help="every scan k times as long (its beams repeated with a 1 cm offset): scans beyond the 17 register rows")
    args = ap.parse_args()
    import torch
    from hector_slam_amd import capi
    dev = torch.device("cuda", 0)
    sizes = [int(s) for s in args.sizes.split(",")]
    B = max(sizes)
    res, size, n_beams = bench.RESOLUTION, bench.MAP_SIZE, bench.N_BEAMS
    if args.workload == "config3":
        bp, bs, truth, init_l0, init_pyr, pts, offs = bench.make_inputs(0, B)[:7]
        init = init_l0 if args.levels == 1 else init_pyr
    else:  # as bench.extra_workload builds it
        import math
        from hector_slam_amd import synth
        n_beams, size, res, room, rmax, _, _ = bench.WORKLOADS[args.workload]
        sfac = float(np.float32(1.0) / np.float32(res))
        world = synth.World.make(room[0], room[1], seed=1234)
        rng_noise = np.random.default_rng(1235)
        bp = synth.loop_trajectory(world, 100).astype(np.float32)
        bs = [synth.make_scan(world, p, n_beams, sfac, rng_noise, range_max=rmax) for p in bp]
        rng = np.random.default_rng(1236)
        base = synth.loop_trajectory(world, B, phase=rng.uniform(0, 2 * math.pi)).astype(np.float64)
        base[:, :2] += rng.uniform(-0.5, 0.5, size=(B, 2)) * (room[0] / 40.0)
        base[:, 2] += rng.uniform(-0.3, 0.3, size=B)
        truth = base.astype(np.float32)
        rng_q = np.random.default_rng(1237)
        scans = [synth.make_scan(world, p, n_beams, sfac, rng_q, pad_to_full=True, range_max=rmax) for p in truth]
        init = synth.perturb_poses(truth, np.random.default_rng(1239), 0.15 if args.levels > 1 else 0.04, 0.05 if args.levels > 1 else 0.01)
        pts, offs = synth.pack_scans(scans)
    if args.thin > 1:
        o = np.asarray(offs, np.int64)
        parts = [pts[o[q]:o[q + 1]][::args.thin] for q in range(B)]
        pts = np.ascontiguousarray(np.concatenate(parts), np.float32)
        offs = np.concatenate([[0], np.cumsum([p.shape[0] for p in parts])]).astype(np.int32)
        n_beams = int(max(p.shape[0] for p in parts))
    if args.stretch > 1:  # offsets[] form: scan q = its own beams, then copies shifted by 1 cm, 2 cm, ...
        o = np.asarray(offs, np.int64)
        parts = [np.concatenate([pts[o[q]:o[q + 1]] + np.float32(0.2 * k) for k in range(args.stretch)]) for q in range(B)]
        pts = np.ascontiguousarray(np.concatenate(parts), np.float32)
        offs = np.concatenate([[0], np.cumsum([p.shape[0] for p in parts])]).astype(np.int32)
        n_beams = int(max(p.shape[0] for p in parts))  # the sizing hint of the CSR form: the longest scan
    d_init = torch.from_numpy(init).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    stream = torch.cuda.current_stream()
    variants = []
    for spec in args.variants.split(";"):
        env = dict(kv.split("=") for kv in spec.split(",") if "=" in kv)
        os.environ.update(env)
        m = capi.MapRepMultiMap(res, size, size, args.levels, device=0)
        for k in env:
            os.environ.pop(k)
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
        m.build_map(bp, bs)
        variants.append((spec, m))
    pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    cov = torch.zeros((B, 9), dtype=torch.float32, device=dev)
    for b in sizes:
        ref = None
        row = {"workload": args.workload, "levels": args.levels, "batch": b}
        for spec, m in variants:
            def launch():
                m.match_batch_device(b, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), n_beams,
                                     pose.data_ptr(), cov.data_ptr(), stream.cuda_stream)
            pose.zero_()
            for _ in range(3):
                launch()
            torch.cuda.synchronize()
            ms = []
            for _ in range(args.launches):
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                launch()
                e.record(stream)
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(e))
            p = pose[:b].cpu().numpy().copy()
            c = cov[:b].cpu().numpy().copy()
            if ref is None:
                ref = (p, c)
            cfg = m.last_launch_config()
            row[spec] = {"us": round(float(np.median(ms)) * 1e3, 2), "kernel": cfg.get("kernel"), "block": cfg.get("block"), "grid": cfg.get("grid"),
                         "bit_identical_to_first": bool(np.array_equal(p.view(np.uint32), ref[0].view(np.uint32)) and
                                                        np.array_equal(c.view(np.uint32), ref[1].view(np.uint32)))}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

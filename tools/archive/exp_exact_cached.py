#!/usr/bin/env python
"""A/B of the exact-order batch kernels on the bench workloads (GPU box):
    python tools/exp_exact_cached.py [config3 config3pyr config4] [--variants "HSM_EXACT_CACHED=0;HSM_EXACT_CACHED=1"]
For every workload: the fast mode, then every variant (env settings applied before the context is created) in
HSM_PARITY_EXACT -- kernel time (HIP events, back-to-back launches) and whether all poses / covariances equal the first
variant's bit for bit (the first variant should be the round-2 form, which the full-size tests pin to the reference).
One JSON line per (workload, variant)."""
import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def build(name, env):
    import torch
    from hector_slam_amd import capi, synth
    beams, size, res, room, rmax, levels, batch = bench.WORKLOADS[name]
    sfac = float(np.float32(1.0) / np.float32(res))
    world = synth.World.make(room[0], room[1], seed=1234)
    rng_noise = np.random.default_rng(1235)
    build_poses = synth.loop_trajectory(world, 100).astype(np.float32)
    build_scans = [synth.make_scan(world, p, beams, sfac, rng_noise, range_max=rmax) for p in build_poses]
    rng = np.random.default_rng(1236)
    base = synth.loop_trajectory(world, batch, phase=rng.uniform(0, 2 * math.pi)).astype(np.float64)
    base[:, :2] += rng.uniform(-0.5, 0.5, size=(batch, 2)) * (room[0] / 40.0)
    base[:, 2] += rng.uniform(-0.3, 0.3, size=batch)
    truth = base.astype(np.float32)
    rng_q = np.random.default_rng(1237)
    scans = [synth.make_scan(world, p, beams, sfac, rng_q, pad_to_full=True, range_max=rmax) for p in truth]
    init = synth.perturb_poses(truth, np.random.default_rng(1239), 0.15 if levels > 1 else 0.04, 0.05 if levels > 1 else 0.01)
    pts, offs = synth.pack_scans(scans)
    return dict(beams=beams, size=size, res=res, levels=levels, batch=batch, build_poses=build_poses,
                build_scans=build_scans, init=init, pts=pts, offs=offs)


def context(w, env):
    from hector_slam_amd import capi
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        m = capi.MapRepMultiMap(w["res"], w["size"], w["size"], w["levels"])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    m.setUpdateFactorFree(0.4)
    m.setUpdateFactorOccupied(0.9)
    m.build_map(w["build_poses"], w["build_scans"])
    return m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["config3", "config3pyr", "config4"])
    ap.add_argument("--variants", default="HSM_EXACT_CACHED=0;HSM_EXACT_CACHED=1")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=0, help="scans per launch (default: the workload's)")
    args = ap.parse_args()
    import torch
    from hector_slam_amd import capi
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    variants = [dict(kv.split("=") for kv in v.split(",") if kv) for v in args.variants.split(";")]
    for name in args.workloads:
        w = build(name, {})
        B = args.batch or w["batch"]
        w["init"] = w["init"][:B]
        w["offs"] = w["offs"][:B + 1]
        d_init = torch.from_numpy(w["init"]).to(dev)
        d_pts = torch.from_numpy(w["pts"]).to(dev)
        d_offs = torch.from_numpy(w["offs"]).to(dev)
        d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
        d_cov = torch.zeros((B, 9), dtype=torch.float32, device=dev)
        its = 6 + 4 * (w["levels"] - 1)

        def timed(m, steps):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3):
                m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), w["beams"],
                                     d_pose.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            ev0.record(stream)
            for _ in range(steps):
                m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), w["beams"],
                                     d_pose.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
            ev1.record(stream)
            torch.cuda.synchronize()
            return ev0.elapsed_time(ev1) / steps

        ref = None
        for vi, env in enumerate([None] + variants):
            m = context(w, env or {})
            if env is not None:
                m.set_parity(capi.PARITY_RELAXED if env.get("MODE") == "relaxed" else capi.PARITY_EXACT)
            else:
                m.set_parity(capi.PARITY_FAST)
            d_pose.zero_()
            d_cov.zero_()
            ms = timed(m, args.steps)
            pose, cov = d_pose.cpu().numpy().copy(), d_cov.cpu().numpy().copy()
            rec = {"workload": name, "variant": "fast" if env is None else env, "kernel_us": ms * 1e3,
                   "M_it_per_s": B * its / ms / 1e3, "cfg": m.last_launch_config()}
            if env is not None:
                if ref is None:
                    ref = (pose, cov)
                rec["pose_identical_to_first_variant"] = float((pose.view(np.uint32) == ref[0].view(np.uint32)).all(1).mean())
                rec["cov_identical_to_first_variant"] = float((cov.view(np.uint32) == ref[1].view(np.uint32)).all(1).mean())
            else:
                fast_pose = pose
            if env is not None and vi == 1:
                dd = np.abs(fast_pose.astype(np.float64) - pose)
                rec["fast_within_1e-4"] = float(((dd[:, :2].max(1) <= 1e-4) & (dd[:, 2] <= 1e-4)).mean())
            if env is not None and vi > 1:  # against the first variant (exact mode = the reference)
                dd = np.abs(ref[0].astype(np.float64) - pose)
                dth = np.abs((dd[:, 2] + np.pi) % (2 * np.pi) - np.pi)
                rec["within_1e-4_of_first"] = float(((dd[:, :2].max(1) <= 1e-4) & (dth <= 1e-4)).mean())
                rec["max_dxy_m_vs_first"] = float(dd[:, :2].max())
            print(json.dumps(rec), flush=True)
            del m


if __name__ == "__main__":
    main()

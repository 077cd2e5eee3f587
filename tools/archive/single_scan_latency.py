#!/usr/bin/env python
"""hsm_match host-call latency (default mode: reference order) by scan length, for launch-knob variants given as
"K=V,K=V;..." (read when a context is created).  Poses of every variant compared bit for bit with the first variant's.
(Round 5 ran it with HSM_EXACT_SINGLE_CW=0/1 -- single scans through the chain-wavefront batch kernel, a knob of that experiment's
build only: profiles/r05/single_scan_latency_chain_wavefront_form.jsonl, README 9.  Pass --variants with knobs that exist.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="default;HSM_PARITY=fast")
    ap.add_argument("--calls", type=int, default=300)
    args = ap.parse_args()
    from hector_slam_amd import capi, synth
    cases = [(181, 256, 1), (360, 1024, 3), (720, 1024, 3), (1081, 1024, 3), (1081, 2048, 1)]
    for beams, size, levels in cases:
        res = 0.05
        ext = size * res
        world = synth.World.make(ext * 0.8, ext * 0.6, n_boxes=4, seed=beams, keep_clear=0.5)
        poses = synth.loop_trajectory(world, 40, frac=0.25).astype(np.float32)
        rng = np.random.default_rng(1)
        s = float(np.float32(1.0) / np.float32(res))
        scans = [synth.make_scan(world, p, beams, s, rng, range_max=min(30.0, ext)) for p in poses]
        row = {"beams": beams, "map": size, "levels": levels, "n": int(scans[0].shape[0])}
        ref = None
        for spec in args.variants.split(";"):
            env = dict(kv.split("=") for kv in spec.split(",") if "=" in kv)
            os.environ.update(env)
            m = capi.MapRepMultiMap(res, size, size, levels, (0.5, 0.5))
            for k in env:
                os.environ.pop(k)
            m.setUpdateFactorFree(0.4)
            m.setUpdateFactorOccupied(0.9)
            for p, sc in zip(poses[:30], scans[:30]):
                m.updateByScan(sc, p)
            m.synchronize()
            q = poses[30].copy()
            q[:2] += np.float32(0.05)
            q[2] += np.float32(0.02)
            out = []
            for _ in range(20):
                out = m.matchData(q, scans[30])
            dts = []
            for _ in range(args.calls):
                t0 = time.perf_counter()
                out = m.matchData(q, scans[30])
                dts.append(time.perf_counter() - t0)
            pose = np.concatenate([np.asarray(out[0], np.float32), np.asarray(out[1], np.float32).ravel()])
            if ref is None:
                ref = pose
            cfg = m.last_launch_config()
            row[spec] = {"us_median": round(float(np.median(dts)) * 1e6, 2), "us_p10": round(float(np.percentile(dts, 10)) * 1e6, 2),
                         "kernel": cfg.get("kernel"), "block": cfg.get("block"),
                         "bit_identical_to_first": bool(np.array_equal(pose.view(np.uint32), ref.view(np.uint32)))}
            m.close()
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()

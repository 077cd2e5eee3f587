#!/usr/bin/env python
"""The ROS-node case of the parity sweep (round-3 verdict, item 2): the reference's UNCHANGED HectorSlamProcessor::update
loop (tests/cpp/slam_driver.cpp) over N >= 5000 scans, once on the reference's CPU map representation
(oracle/_ref/slam_driver_ref) and once on the drop-in facade in the library's DEFAULT parity mode
(oracle/_ref/slam_driver_mi355) -- each side free-running: own matched pose into its map update (the node's default
0.4 m / 0.9 rad thresholds) and, plus the odometry delta, into its next start estimate.  Reports the pose deviation per
step (max, fraction within 1e-4 m / 1e-4 rad, the step of the worst) and how many map cells differ at the end.

usage: node_loop_parity.py [N=5000] [--parity auto|fast|exact] [--min-dist 0.4] [--min-ang 0.9]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def opt(name, default):
    return type(default)(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def main():
    from hector_slam_amd import synth
    import test_facade_dropin as fd
    pos = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
    N = int(pos[0]) if pos else 5000
    parity = opt("--parity", "auto")
    min_dist, min_ang = opt("--min-dist", 0.4), opt("--min-ang", 0.9)
    T = 400
    sc = synth.make_scene(n_beams=1081, map_size=1024, levels=3, resolution=0.05, n_build=T, n_query=1, room=(40.0, 30.0), seed=5)
    # N steps = the 400-pose loop driven round and round
    sc.build_poses = np.concatenate([sc.build_poses] * (N // T + 1))[:N]
    sc.build_scans = (sc.build_scans * (N // T + 1))[:N]
    # consecutive loops: the odometry delta across the seam is the first pose minus the last one of the loop
    with tempfile.TemporaryDirectory(prefix="hsm_node_", dir="/tmp") as d:
        scen = os.path.join(d, "s.bin")
        fd.write_scenario(scen, sc, N, hooks=0, min_dist=min_dist, min_ang=min_ang, origo=(0.0, 0.0), mwm_at=())
        t0 = time.perf_counter()
        fd.run(fd.REF_BIN, scen, os.path.join(d, "ref.bin"))
        t1 = time.perf_counter()
        env = dict(os.environ, HSM_PARITY=parity)
        r = subprocess.run([fd.GPU_BIN, scen, os.path.join(d, "gpu.bin")], capture_output=True, text=True, timeout=1200, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        t2 = time.perf_counter()
        a, b = fd.read_output(os.path.join(d, "ref.bin"), N), fd.read_output(os.path.join(d, "gpu.bin"), N)
    dxy = np.abs(a["pose"][:, :2].astype(np.float64) - b["pose"][:, :2]).max(1)
    dth = np.abs((a["pose"][:, 2].astype(np.float64) - b["pose"][:, 2] + np.pi) % (2 * np.pi) - np.pi)
    within = (dxy <= 1e-4) & (dth <= 1e-4)
    cells = [{"level": i, "touched": int((x["val"] != 0).sum()),
              "differ": int((x["val"].view(np.uint32) != y["val"].view(np.uint32)).sum())} for i, (x, y) in enumerate(zip(a["grids"], b["grids"]))]
    print(json.dumps({"steps": N, "facade_parity_mode": parity, "map_update_thresholds": [min_dist, min_ang],
                      "poses_bit_identical_frac": float((a["pose"].view(np.uint32) == b["pose"].view(np.uint32)).all(1).mean()),
                      "within_1e-4_frac": float(within.mean()), "max_dxy_m": float(dxy.max()), "max_dtheta_rad": float(dth.max()),
                      "step_of_max": int(dxy.argmax()), "p99_dxy_m": float(np.percentile(dxy, 99)),
                      "first_step_beyond_1e-4": int(np.argmax(~within)) if (~within).any() else None,
                      "map_cells": cells, "ref_loop_s": round(t1 - t0, 2), "gpu_loop_s": round(t2 - t1, 2)}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-round timeline of ONE workgroup of the headline kernel (round-5 verdict item 4): run the headline batch on a -DHSM_XTIMELINE
build (HSM_LIB=...), read the s_memtime stamps workgroup 0 left for the last GN step -- per wavefront and row: arrival at the
round's barrier, release, end of the chain job -- and print, per round, how long every wavefront worked before the barrier and how
long it was parked there.  usage: HSM_LIB=<variant.so> tools/study/exact_timeline.py [out.json]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from hector_slam_amd import capi


def main():
    B = 4096
    build_poses, build_scans, truth, init, init_pyr, pts, offs, _ = bench.make_inputs(0, B)
    m = capi.MapRepMultiMap(bench.RESOLUTION, bench.MAP_SIZE, bench.MAP_SIZE, 1)
    m.setUpdateFactorFree(0.4); m.setUpdateFactorOccupied(0.9)
    m.build_map(build_poses, build_scans)
    dev = torch.device("cuda", 0)
    d_init, d_pts, d_offs = (torch.from_numpy(x).to(dev) for x in (init, pts, offs))
    d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    BPL, NW = 17, 4
    probe = torch.zeros(NW * BPL * 4 + 8, dtype=torch.int64, device=dev)
    m.set_clock_probe(probe.data_ptr())
    for _ in range(30):
        m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), bench.N_BEAMS, d_pose.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    t = probe.cpu().numpy()[: NW * BPL * 4].reshape(NW, BPL, 4).astype(np.int64)
    print(m.last_launch_config().get("kernel"))
    rec = {"rows": []}
    t0 = t[:, 0, 1].min()
    print("row | per wave: work before the barrier / parked at it / job (cycles) | round length (release to release)")
    prev_rel = None
    for k in range(BPL):
        arr, rel, job = t[:, k, 0], t[:, k, 1], t[:, k, 2]
        line = []
        for w in range(NW):
            start = t[w, k - 1, 2] if (k > 0 and t[w, k - 1, 2] >= t[w, k - 1, 1]) else (t[w, k - 1, 1] if k > 0 else arr[w])
            work = arr[w] - start if k > 0 else 0
            line.append((int(work), int(rel[w] - arr[w]), int(job[w] - rel[w]) if job[w] >= rel[w] else 0))
        rl = int(rel.max() - prev_rel) if prev_rel is not None else 0
        prev_rel = rel.max()
        rec["rows"].append({"row": k, "waves": line, "round_cycles": rl})
        print(f"{k:3d} | " + "  ".join(f"{a:5d}/{b:5d}/{c:4d}" for a, b, c in line) + f" | {rl}")
    rounds = [r["round_cycles"] for r in rec["rows"][1:]]
    rec["mean_round_cycles"] = float(np.mean(rounds))
    print("mean round", rec["mean_round_cycles"], "cycles; mean parked per wave and round",
          float(np.mean([w[1] for r in rec["rows"][1:] for w in r["waves"]])))
    if len(sys.argv) > 1:
        json.dump(rec, open(sys.argv[1], "w"))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Experiment: the 4096-scan headline batch as TWO concurrent launches on two streams -- 3072 scans in the chain-wavefront form (three
workgroups of five wavefronts per CU, 80 VGPRs) + 1024 scans in the four-producer form (one workgroup per CU, 128 VGPRs), which fit on a
CU together -- against the one launch of four four-producer workgroups per CU.  usage: tools/study/mixed_forms_two_streams.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from hector_slam_amd import capi, synth
B, N = 4096, 1081
dev = torch.device("cuda", 0)
sc = synth.make_scene(n_beams=N, resolution=0.05, n_build=60, n_query=B, seed=77, pad_to_full=True, map_size=2048, levels=1, room=(40.0, 30.0))


def ctx(wps=0, **env):
    for k, v in env.items():
        os.environ[k] = v
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels, device=0, **({"waves_per_scan": wps} if wps else {}))
    for k in env:
        os.environ.pop(k)
    g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
    g.build_map(sc.build_poses, sc.build_scans)
    return g


one, a, b = ctx(), ctx(), ctx(wps=1, HSM_EXACT_CHAIN_WAVE="0")
pts, offs = synth.pack_scans(sc.query_scans)
d_pts, d_offs, d_init = torch.from_numpy(pts).to(dev), torch.from_numpy(offs).to(dev), torch.from_numpy(sc.query_init).to(dev)
d_pose, d_pose2 = torch.zeros((B, 3), dtype=torch.float32, device=dev), torch.zeros((B, 3), dtype=torch.float32, device=dev)
s0, s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()


def single():
    one.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, d_pose.data_ptr(), 0, s0.cuda_stream)


def split(na):
    def f():
        a.match_batch_device(na, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N, d_pose2.data_ptr(), 0, s1.cuda_stream)
        b.match_batch_device(B - na, d_init.data_ptr() + 12 * na, d_pts.data_ptr(), d_offs.data_ptr() + 4 * na, N, d_pose2.data_ptr() + 12 * na, 0, s2.cuda_stream)
    return f


def timed(fn, n=300):
    for _ in range(600): fn()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e6)
    return sorted(ts)[1]


print("one launch, 4096 scans: %.2f us" % timed(single), one.last_launch_config()["kernel"], flush=True)
for na in (3072, 2048, 3584):
    t = timed(split(na))
    same = np.array_equal(d_pose.cpu().numpy().view(np.uint32), d_pose2.cpu().numpy().view(np.uint32))
    print("two streams, %d scans (%s, block %d) + %d scans (%s, block %d): %.2f us per pair; poses %s" % (
        na, a.last_launch_config()["kernel"], a.last_launch_config()["block"], B - na, b.last_launch_config()["kernel"], b.last_launch_config()["block"], t,
        "bit-identical to the one launch" if same else "DIFFER"), flush=True)
print("one launch again: %.2f us" % timed(single), flush=True)

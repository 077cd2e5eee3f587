#!/usr/bin/env python
"""bench.py -- scan-match Gauss-Newton iterations/sec on MI355X (BASELINE.json metric).

A "step" is one batched matchData over B = 4096 independent 1081-beam scans per GPU on a 2048^2
map (BASELINE.json configs[2]: "batch=4096 concurrent 1081-beam scans, 2048^2 map, 1 GPU"), i.e.
B x 6 Gauss-Newton iterations (1 + 5, ScanMatcher.h:74,94-97) in ONE kernel launch, with scans,
start poses and the map already resident in HBM.  With --gpus N every rank holds a replica of
the map and its own 4096 scans (weak scaling); the step ends with the single RCCL all-gather of
the poses.  Rank 0 prints ONE JSON line (see the driver contract in the task description).

Extra evidence in the same line:
  roofline      dominant kernel (gn_match_kernel) timed with HIP events on its own stream;
                achieved = algorithmic bytes per launch ((24*N + 60) B per GN iteration,
                SURVEY.md 8(d)) / mean kernel time; peak = 8 TB/s HBM3E
  cpu_baseline  the reference CPU matcher (oracle/_ref, else the oracle port) on the SAME map and
                scans, single thread (the reference is single threaded), bounded sample, plus the
                GPU-vs-CPU pose deviation on that sample (parity evidence, tolerance 1e-4)
  pyramid       the same batch through the full 3-level 2048/1024/512 schedule (14 iterations)
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_BEAMS = 1081
MAP_SIZE = 2048
RESOLUTION = 0.05
BATCH_PER_GPU = 4096
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md


def algorithmic_bytes_per_iteration(n_beams: int) -> int:
    return 24 * n_beams + 60  # 8 B endpoint + 4 x 4 B samples per beam; 12 B pose in + 48 B H,dTr out


def make_inputs(rank: int, batch: int, n_build: int = 200):
    """Deterministic world, map-building scans and this rank's query batch (distinct per rank)."""
    from hector_slam_amd import synth
    world = synth.World.make(40.0, 30.0, seed=1234)
    s = float(np.float32(1.0) / np.float32(RESOLUTION))
    rng_noise = np.random.default_rng(1235)
    build_poses = synth.loop_trajectory(world, n_build).astype(np.float32)
    build_scans = [synth.make_scan(world, p, N_BEAMS, s, rng_noise) for p in build_poses]
    # query poses: spread along the loop with lateral jitter; every scan padded to exactly 1081 beams
    rng = np.random.default_rng(1236 + 7919 * rank)
    base = synth.loop_trajectory(world, batch, phase=rng.uniform(0, 2 * math.pi)).astype(np.float64)
    base[:, :2] += rng.uniform(-0.5, 0.5, size=(batch, 2))
    base[:, 2] += rng.uniform(-0.3, 0.3, size=batch)
    truth = base.astype(np.float32)
    rng_q = np.random.default_rng(1237 + 7919 * rank)
    scans = [synth.make_scan(world, p, N_BEAMS, s, rng_q, pad_to_full=True) for p in truth]
    # start estimates: the single-level (level-0-only) run has no coarse levels to pull a far start
    # in, so its hypotheses stay within ~1 cell (0.04 m, 0.01 rad) of the truth; the 3-level pyramid
    # run uses SURVEY.md 8(d)'s +-0.15 m / +-0.05 rad.  Both converge on the CPU reference, which
    # makes the GPU-vs-CPU pose deviation a meaningful parity figure over the whole sample.
    init_l0 = synth.perturb_poses(truth, np.random.default_rng(1238 + 7919 * rank), 0.04, 0.01)
    init_pyr = synth.perturb_poses(truth, np.random.default_rng(1239 + 7919 * rank), 0.15, 0.05)
    pts, offs = synth.pack_scans(scans)
    assert pts.shape[0] == batch * N_BEAMS
    return build_poses, build_scans, truth, init_l0, init_pyr, pts, offs


def cpu_baseline(build_poses, build_scans, init, pts, offs, gpu_pose, levels: int, budget_s: float = 12.0,
                 n_par: int = 512):
    """Reference CPU path on the same map + scans, one thread, bounded by ``budget_s`` of matching."""
    from oracle import pyoracle
    pyoracle.build()
    kind = "hr" if pyoracle.available("hr") else "ho"
    o = pyoracle.Oracle(kind, RESOLUTION, MAP_SIZE, MAP_SIZE, levels)
    o.set_update_factor_free(0.4)
    o.set_update_factor_occupied(0.9)
    o.build_map(build_poses, build_scans)
    B = init.shape[0]
    its_per_match = 6 + 4 * (levels - 1)
    # warm pass (populates the reference's probability cache, its steady state) + parity sample
    n_par = min(B, n_par)
    cpu_pose = o.match_many(init[:n_par], pts, offs[:n_par + 1])
    d = np.abs(cpu_pose.astype(np.float64) - gpu_pose[:n_par].astype(np.float64))
    dth = np.abs((d[:, 2] + np.pi) % (2 * np.pi) - np.pi)
    t0 = time.perf_counter()
    done = 0
    while budget_s > 0:  # whole passes over the batch, each one C loop of B matchData calls
        o.match_many(init, pts, offs)
        done += B
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    par = {"parity_sample": n_par, "max_abs_dxy_m": float(d[:, :2].max()), "max_abs_dtheta_rad": float(dth.max()),
           "median_abs_dxy_m": float(np.median(d[:, :2])), "tolerance": "1e-4 m / 1e-4 rad"}
    if budget_s <= 0:
        return par
    return {
        "value": done * its_per_match / dt, "unit": "GN it/s", "cores": 1,
        "kind": "reference" if kind == "hr" else "port",
        "sample": f"{done} matchData calls ({done * its_per_match} GN iterations, {dt:.1f} s) over the same "
                  f"{B} scans + map, warm probability cache, single thread; "
                  + ("unmodified reference headers via private Eigen stand-in" if kind == "hr"
                     else "plain-C++ restatement of the reference"),
        "host_cpu": model, "host_logical_cores": os.cpu_count(), **par,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="scans per GPU")
    ap.add_argument("--levels", type=int, default=1, help="pyramid levels of the headline run")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-pyramid", action="store_true", help="skip the 3-level extra run")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from hector_slam_amd import capi, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    B = args.batch
    build_poses, build_scans, truth, init, init_pyr, pts, offs = make_inputs(rank, B)

    def build_matcher(levels):
        m = capi.MapRepMultiMap(RESOLUTION, MAP_SIZE, MAP_SIZE, levels, device=local_rank)
        m.setUpdateFactorFree(0.4)
        m.setUpdateFactorOccupied(0.9)
        m.build_map(build_poses, build_scans)  # the product's own updateByScan kernels
        return m

    stream = torch.cuda.current_stream()
    d_init_l0 = torch.from_numpy(init).to(dev)
    d_init_pyr = torch.from_numpy(init_pyr).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    d_cov = torch.zeros((B, 9), dtype=torch.float32, device=dev)
    total = B * world

    def run(matcher, d_init, steps, warmup, gather=True):
        its = matcher.gn_iterations_per_match()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

        def step(ev=None):
            if ev:
                ev[0].record(stream)
            matcher.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N_BEAMS,
                                       d_pose.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
            if ev:
                ev[1].record(stream)
            if world > 1 and gather:
                return sharding.all_gather_rows(d_pose, total)
            return d_pose

        for _ in range(warmup):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(evs[k])
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        return dt, kern_ms, its

    matcher = build_matcher(args.levels)
    dt, kern_ms, its = run(matcher, d_init_l0 if args.levels == 1 else d_init_pyr, args.steps, args.warmup)
    gpu_pose = d_pose.cpu().numpy()
    cfg = matcher.last_launch_config()
    value = total * its * args.steps / dt
    bytes_per_launch = algorithmic_bytes_per_iteration(N_BEAMS) * its * B
    achieved = bytes_per_launch / (kern_ms * 1e-3)

    out = {
        "metric": "scan-match GN iterations/sec (1081-beam, 2048^2 map)",
        "value": value, "unit": "GN it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[2]: batch={B}/GPU concurrent {N_BEAMS}-beam scans (distinct pose+scan "
                               f"pairs), {MAP_SIZE}^2 map, {args.levels}-level matchData = {its} GN it/scan",
                   "batch_per_gpu": B, "global_batch": total, "beams": N_BEAMS, "map": MAP_SIZE,
                   "levels": args.levels, "gn_iterations_per_scan": its, "parallelism": f"dp{world}",
                   "kernel": cfg},
        "matchdata_per_s": total * args.steps / dt,
        "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK, "traffic": None,
                     "kernel": "gn_match_kernel", "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "frac_of_measured_copy_bw_6.29TBps": achieved / 6.29e12},
    }
    conv = np.abs(gpu_pose.astype(np.float64) - truth.astype(np.float64))
    out["convergence"] = {"median_abs_err_xy_m": float(np.median(conv[:, :2])),
                          "median_abs_err_theta_rad": float(np.median(conv[:, 2]))}

    if not args.no_pyramid and args.levels == 1:
        m3 = build_matcher(3)
        steps3 = max(3, args.steps // 3)
        dt3, k3, its3 = run(m3, d_init_pyr, steps3, 2, gather=True)
        out["pyramid"] = {"levels": 3, "gn_iterations_per_scan": its3,
                          "value": total * its3 * steps3 / dt3, "unit": "GN it/s",
                          "matchdata_per_s": total * steps3 / dt3, "kernel_ms": k3,
                          "start_error": "+-0.15 m, +-0.05 rad"}
        if rank == 0 and world == 1 and not args.no_cpu:  # parity only (no timing) for the pyramid
            out["pyramid"]["parity_vs_cpu"] = cpu_baseline(build_poses, build_scans, init_pyr, pts, offs,
                                                           d_pose.cpu().numpy(), 3, budget_s=0.0, n_par=256)
        m3.close()

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(build_poses, build_scans, init if args.levels == 1 else init_pyr, pts,
                                           offs, gpu_pose, args.levels)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()

"""the reference CPU matcher timed on the GPU box's host cores (oracle/_ref, else the oracle port): only ever the baseline beside
the measurement, never the thing measured."""
from __future__ import annotations

import os
import time

import numpy as np

from .common import MAP_SIZE, RESOLUTION


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
    # cold after an update: onMapUpdated() bumps the generation of the reference's probability cache
    # (GridMapCacheArray.h:69-72), so the first matchData after every map update pays exp() + a divide per touched cell
    n_cold = min(B, 256) if budget_s > 0 else 0
    tc0 = time.perf_counter()
    for q in range(n_cold):
        o.on_map_updated()
        o.match(init[q], pts[offs[q]:offs[q + 1]])
    dt_cold = max(time.perf_counter() - tc0, 1e-9)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    same = (cpu_pose.view(np.uint32) == np.ascontiguousarray(gpu_pose[:n_par], np.float32).view(np.uint32)).all(1)
    par = {"parity_sample": n_par, "bit_identical_pose_fraction": float(same.mean()),
           "max_abs_dxy_m": float(d[:, :2].max()), "max_abs_dtheta_rad": float(dth.max()),
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
        "cold_after_update": {"value": n_cold * its_per_match / dt_cold, "unit": "GN it/s",
                              "sample": f"{n_cold} matchData calls, each right after onMapUpdated() (probability cache invalidated), {dt_cold:.2f} s"},
        "host_cpu": model, "host_logical_cores": os.cpu_count(), **par,
    }


def cpu_baseline_all_cores(build_poses, build_scans, init, pts, offs, levels: int, budget_s: float = 4.0,
                           max_threads: int = 64):
    """The same reference matcher on T host threads, each with its OWN map + matcher state (the reference has no
    threading of its own: one ROS callback, hector_mapping/src/main.cpp:40), scans split contiguously.  An
    aggregate-throughput yardstick for the GPU/CPU ratio, reported next to the single-thread baseline."""
    import threading
    from oracle import pyoracle
    kind = "hr" if pyoracle.available("hr") else "ho"
    T = max(1, min(max_threads, (os.cpu_count() or 2) // 2))
    B = init.shape[0]
    its_per_match = 6 + 4 * (levels - 1)
    bounds = [(B * t // T, B * (t + 1) // T) for t in range(T)]
    oracles = [None] * T

    def prepare(t):
        o = pyoracle.Oracle(kind, RESOLUTION, MAP_SIZE, MAP_SIZE, levels)
        o.set_update_factor_free(0.4)
        o.set_update_factor_occupied(0.9)
        o.build_map(build_poses, build_scans)
        b, e = bounds[t]
        o.match_many(init[b:e], pts, offs[b:e + 1])  # warm the probability cache
        oracles[t] = o

    th = [threading.Thread(target=prepare, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    done = [0] * T
    stop = time.perf_counter() + budget_s

    def work(t):
        b, e = bounds[t]
        while time.perf_counter() < stop:
            oracles[t].match_many(init[b:e], pts, offs[b:e + 1])  # ctypes releases the GIL during the C loop
            done[t] += e - b

    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    return {"value": sum(done) * its_per_match / dt, "unit": "GN it/s", "cores": T,
            "kind": "reference" if kind == "hr" else "port",
            "sample": f"{sum(done)} matchData calls in {dt:.1f} s on {T} threads, one private map + matcher per thread"}


def config1_plumbing(capi):
    """BASELINE configs[0]: single 181-beam synthetic scan, 256x256 single-resolution map, 5 GN iterations on the reference CPU
    path (plumbing) -- timed on the host, and the same call through the C ABI in HSM_PARITY_EXACT compared bit for bit."""
    from hector_slam_amd import synth
    from oracle import pyoracle
    pyoracle.build()
    kind = "hr" if pyoracle.available("hr") else "ho"
    sc = synth.make_scene(n_beams=181, map_size=256, levels=1, resolution=0.1, n_build=40, n_query=8, room=(20.0, 15.0), seed=4321)
    o = pyoracle.Oracle(kind, sc.resolution, sc.map_size, sc.map_size, 1)
    g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, 1, parity=capi.PARITY_EXACT)
    for x in (o.set_update_factor_free, g.setUpdateFactorFree):
        x(0.4)
    for x in (o.set_update_factor_occupied, g.setUpdateFactorOccupied):
        x(0.9)
    o.build_map(sc.build_poses, sc.build_scans)
    g.build_map(sc.build_poses, sc.build_scans)
    same = True
    for q in range(8):
        po, co = o.match_level(0, sc.query_init[q], sc.query_scans[q], 5)
        pg, cg = g.match_level(0, sc.query_init[q], sc.query_scans[q], 5)
        same &= bool((po.view(np.uint32) == pg.view(np.uint32)).all() and (co.view(np.uint32) == cg.view(np.uint32)).all())
    n = 2000
    t0 = time.perf_counter()
    for k in range(n):
        o.match_level(0, sc.query_init[k % 8], sc.query_scans[k % 8], 5)
    dt = time.perf_counter() - t0
    lat = []
    for k in range(200):
        a = time.perf_counter()
        g.match_level(0, sc.query_init[k % 8], sc.query_scans[k % 8], 5)
        lat.append(time.perf_counter() - a)
    g.close()
    return {"workload": "configs[0]: single 181-beam scan, 256^2 single-resolution map, 5 GN iterations (+ the unconditional first step)",
            "cpu_reference": {"kind": "reference" if kind == "hr" else "port", "us_per_match": dt / n * 1e6, "gn_it_per_s": 6 * n / dt, "cores": 1},
            "mi355x_host_call_us": float(np.median(lat[20:])) * 1e6,
            "exact_mode_pose_and_cov_bit_identical": same}

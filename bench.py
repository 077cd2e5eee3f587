#!/usr/bin/env python
"""bench.py -- scan-match Gauss-Newton iterations/sec on MI355X (BASELINE.json metric).

A "step" is one batched matchData over B = 4096 independent 1081-beam scans per GPU on a 2048^2 map
(BASELINE.json configs[2]: "batch=4096 concurrent 1081-beam scans, 2048^2 map, 1 GPU"), i.e. B x 6 Gauss-Newton
iterations (1 + 5, ScanMatcher.h:74,94-97) in ONE kernel launch, with scans, start poses and the map already
resident in HBM, in the library's DEFAULT parity mode (the reference's summation order: poses bit-identical to the
reference CPU matcher).  With --gpus N every rank holds a replica of the map and its own 4096 scans (weak scaling); the
one collective of the path -- an RCCL all-gather of the [B,3] poses -- is double buffered and asynchronous.

OUTPUT.  The LAST stdout line of rank 0 is ONE compact JSON object (< 4 KB, `compact_line`; tests/test_bench_line.py): the
driver's contract keys, `roofline` and `cpu_baseline`, and `details` = the path of the file that holds the full record
(gpurun_out/bench_details*.json, or $HSM_BENCH_DETAILS).  Round 4 printed the full record as the line -- 24.6 KB -- and the
driver could not parse it.

  roofline      the dominant kernel, one HIP event pair on its stream around the timed region.  `bound` names what binds
                it -- VALU instruction issue -- and achieved / peak / frac are wave64 VALU instructions per second
                against 1024 SIMDs x 2.4 GHz / 2 cycles, from counters collected IN THIS RUN: bench.py re-executes itself
                (`--leg pmc`) under `rocprofv3 --pmc`, one pass per counter group (N > 1 or --no-pmc: the committed
                profile profiles/r05/traffic.json, labelled).  `traffic` / `hbm_frac` = HBM bytes per launch from the same
                passes (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction).  `contract_8d` keeps SURVEY.md
                8(d)'s figure (algorithmic bytes / time against 8 TB/s), labelled: it exceeds 1 because endpoints and
                texels are served on chip, i.e. it is not a utilisation of anything.
  cpu_baseline  the reference CPU matcher (oracle/_ref, else the oracle port) on the SAME map and scans, single thread
                (the reference is single threaded), ~12 s of matching, plus the GPU-vs-CPU pose deviation on a 512-scan
                sample and the fraction of bit-identical poses.  Runs on a host thread WHILE the counter passes run in
                child processes, so the default run takes ~15 s on the GPU box.
  fast_mode     (details file; `fast_mode_value` in the line) the same launch with HSM_PARITY_FAST (tree summation, opt-in)

--all-configs adds to the DETAILS file: the same batch from SURVEY 8(d)'s start errors, HSM_PARITY_RELAXED, the all-cores CPU
leg, the 3-level pyramid leg, independent batches on 4 streams, and compact child runs of the other BASELINE configs
(configs[0], [1], [3] share, [4] replica), each with its own counter passes (~80 s).

--workload config2|config3pyr|config4|config5 measures one of the other BASELINE configs (latency of a single scan, 3-level
batch, 4096^2 pyramid, dense 16k-beam match+update loop) on its own, same output convention.  config3pyr / config4 / config5
also run with --gpus N: config4 is configs[3] itself at N = 8 (4096 scans per GPU on the 3-level 4096^2 pyramid, all-gather
of the poses); config5 is configs[4] (replicated pyramid: rank 0 matches, pose + scan are broadcast, every rank replays the
update, the maps are compared across ranks at the end).
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

# Extra single-GPU workloads (BASELINE.json configs other than the headline one), `--workload NAME`:
#   name: (beams, map size, resolution, room, sensor range, levels, batch per GPU)
WORKLOADS = {
    "config3": (1081, 2048, 0.05, (40.0, 30.0), 30.0, 1, 4096),        # headline (configs[2]), level-0 GN
    "config3pyr": (1081, 2048, 0.05, (40.0, 30.0), 30.0, 3, 4096),     # the same batch, full 3-level matchData
    "config2": (1081, 1024, 0.05, (40.0, 30.0), 30.0, 3, 1),           # configs[1]: one scan, latency
    "config4": (1081, 4096, 0.05, (160.0, 120.0), 120.0, 3, 4096),     # configs[3]: one GPU's share of 32768
    "config5": (16384, 8192, 0.05, (320.0, 240.0), 240.0, 3, 1),       # configs[4]: dense scan, match+update loop
}


def multi_rank_record(dt_local: float, kern_ms_local: float, dev, gathered=None):
    """N > 1: what every rank measured and whether all ranks hold the same gathered poses -- the self-check of the
    multi-rank path (a broken gather or a rank that did not run shows up in the line itself)"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([dt_local, kern_ms_local], dtype=torch.float64, device=dev)
    allv = torch.empty((world, 2), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(allv, mine.reshape(1, 2))
    rec = {"world_size": world, "backend": dist.get_backend(), "per_rank_timed_region_s": [float(x) for x in allv[:, 0].cpu()],
           "per_rank_kernel_ms": [float(x) for x in allv[:, 1].cpu()]}
    if gathered is not None:
        g = gathered.contiguous().view(torch.int32).to(torch.int64)
        dig = torch.stack([g.sum(), (g * torch.arange(1, g.numel() + 1, device=g.device).reshape(g.shape)).sum()]).reshape(1, 2)
        alld = torch.empty((world, 2), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(alld, dig.to(dev))
        rec["gathered_poses_identical_on_all_ranks"] = bool((alld == alld[0:1]).all().item())
        rec["gathered_rows"] = int(gathered.shape[0])
    return rec


def pose_stats(a, b):
    """how two sets of poses of the same scans compare: bit-identical fraction, fraction within 1e-4 m / 1e-4 rad, worst"""
    dd = np.abs(a.astype(np.float64) - b.astype(np.float64))
    dd[:, 2] = np.abs((dd[:, 2] + np.pi) % (2 * np.pi) - np.pi)
    return {"scans": int(a.shape[0]), "bit_identical": float((a.view(np.uint32) == b.view(np.uint32)).all(1).mean()),
            "within_1e-4": float(((dd[:, :2].max(1) <= 1e-4) & (dd[:, 2] <= 1e-4)).mean()), "max_abs_dxy_m": float(dd[:, :2].max())}


def algorithmic_bytes_per_iteration(n_beams: int) -> int:
    return 24 * n_beams + 60  # 8 B endpoint + 4 x 4 B samples per beam; 12 B pose in + 48 B H,dTr out


# ---- the ONE line the driver parses ------------------------------------------------------------------------------------------
# Round 4's line had grown to 24.6 KB (five configs, three parity modes, counter dumps) and the driver could not parse it.  The
# last stdout line is now a compact record (< 4 KB, checked by tests/test_bench_line.py); everything else goes to a details
# file next to it.
LINE_LIMIT = 4096
_ROOF_KEYS = ("kernel", "kernel_ms", "bound", "unit", "achieved", "peak", "frac", "traffic")
_CPU_KEYS = ("value", "unit", "cores", "kind", "all_cores", "sample", "bit_identical_pose_fraction", "max_abs_dxy_m", "max_abs_dtheta_rad",
             "parity_sample", "host_cpu", "ms_per_step", "max_abs_dxy_m_vs_gpu", "max_abs_dev_vs_gpu", "latency_us")
_CFG_KEYS = ("workload", "batch_per_gpu", "global_batch", "beams", "map", "levels", "gn_iterations_per_scan", "parallelism", "parity_mode", "gather")


def _short(v, n=160):
    return v if not isinstance(v, str) or len(v) <= n else v[: n - 3] + "..."


def compact_line(out: dict, details_path) -> str:
    """The driver's record: the contract keys + roofline + cpu_baseline, nothing nested deeper than one level, < LINE_LIMIT bytes."""
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = {k: _short(cfg[k], 220) for k in _CFG_KEYS if k in cfg}
    kern = cfg.get("kernel")
    if isinstance(kern, dict):
        line["config"]["parity_effective"] = kern.get("parity_effective")
    rf = out.get("roofline")
    if isinstance(rf, dict):
        r = {k: rf.get(k) for k in _ROOF_KEYS if k in rf}
        hbm = rf.get("hbm") or {}
        if hbm.get("frac") is not None:
            r["hbm_frac"] = hbm["frac"]
            r["traffic_over_algorithmic"] = hbm.get("traffic_over_algorithmic")
        con = rf.get("contract") or {}
        if con.get("frac") is not None:
            r["contract_8d"] = {"bound": "hbm", "achieved": con.get("achieved"), "peak": con.get("peak"), "unit": con.get("unit"), "frac": con["frac"]}
        if rf.get("counter_source"):
            r["counter_source"] = _short(rf["counter_source"], 120)
        line["roofline"] = r
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: _short(cb[k], 200) for k in _CPU_KEYS if k in cb}
    for k in ("matchdata_per_s", "match_ms", "update_ms"):
        if out.get(k) is not None:
            line[k] = out[k]
    ur = out.get("update_roofline")
    if isinstance(ur, dict):
        line["update_roofline"] = {k: ur.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "kernel_time_us") if k in ur}
    fm = out.get("fast_mode")
    if isinstance(fm, dict) and fm.get("value") is not None:
        line["fast_mode_value"] = fm["value"]
    su = out.get("sustained")
    if isinstance(su, dict):
        line["sustained"] = {k: su.get(k) for k in ("seconds", "launches", "ms_per_step", "value", "sclk_hz")}
    gl = out.get("gather_legs")
    if isinstance(gl, dict):
        line["gather_legs"] = {k: ({"value": v.get("value"), "ms_per_step": v.get("ms_per_step")} if "value" in v else v) for k, v in gl.items()}
    line["details"] = details_path
    s = json.dumps(line, separators=(",", ":"))
    if len(s) >= LINE_LIMIT:  # never exceed the limit: shed the optional blocks, longest first
        for k in ("update_roofline", "fast_mode_value", "matchdata_per_s", "gather_legs"):
            line.pop(k, None)
        line["config"] = {k: _short(v, 80) for k, v in line["config"].items()}
        if "cpu_baseline" in line:
            line["cpu_baseline"] = {k: _short(v, 80) for k, v in line["cpu_baseline"].items() if k in ("value", "unit", "cores", "kind", "sample")}
        s = json.dumps(line, separators=(",", ":"))
    assert len(s) < LINE_LIMIT, len(s)
    return s


def details_file(out: dict):
    """where the full record goes: gpurun_out/ on the GPU box (merged back by gpurun), overridable with HSM_BENCH_DETAILS"""
    path = os.environ.get("HSM_BENCH_DETAILS")
    if not path:
        n = out.get("n_gpus", 1)
        tag = (os.environ.get("HSM_BENCH_TAG") or "").strip()
        path = os.path.join(ROOT, "gpurun_out", f"bench_details{('_' + tag) if tag else ''}{('_n%d' % n) if n and n > 1 else ''}.json")
    return path


def flush_c_stdio():
    """RCCL prints a version banner ("RCCL version : ...", "Librccl path : ...") with printf; piped, that sits in libc's stdout
    buffer until the process exits -- i.e. it would land BEHIND the JSON line.  Flushing libc's streams first puts it in front."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


_DEFER_EMIT = False  # N > 1: the line is held back until the process group is gone and the other ranks have exited (main)
_PENDING = []


def emit(out: dict):
    """Top-level result: full record -> details file, compact record -> the LAST stdout line.  A child leg of another bench.py
    (run_child sets HSM_BENCH_CHILD=1) prints its full record for the parent to embed."""
    if _DEFER_EMIT:
        _PENDING.append(out)
        return
    flush_c_stdio()
    if os.environ.get("HSM_BENCH_CHILD") == "1":
        print(json.dumps(out))
        return
    path = details_file(out)
    rel = None
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        rel = os.path.relpath(path, ROOT)
    except OSError as e:
        rel = f"(not written: {e})"
    sys.stdout.flush()
    print(compact_line(out, rel), flush=True)


def make_inputs(rank: int, batch: int, n_build: int = 200):
    """Deterministic world, map-building scans and this rank's query batch (distinct per rank).  The child legs of one run
    (counter passes, pyramid, pipelined) re-use what the parent generated: HSM_BENCH_INPUT_CACHE names a directory the parent
    created for the purpose (ray casting 4296 scans is ~4 s of numpy per process otherwise)."""
    from hector_slam_amd import synth
    cache = os.environ.get("HSM_BENCH_INPUT_CACHE")
    cfile = os.path.join(cache, f"inputs_r{rank}_b{batch}_n{n_build}.npz") if cache else None
    if cfile and os.path.exists(cfile):
        z = np.load(cfile)
        bo = z["build_offs"]
        return (z["build_poses"], [z["build_pts"][bo[i]:bo[i + 1]] for i in range(len(bo) - 1)], z["truth"], z["init_l0"],
                z["init_pyr"], z["pts"], z["offs"], z["init_gentle"])
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
    # start estimates: SURVEY.md 8(d)'s +-0.15 m / +-0.05 rad, for the level-0 headline batch (round 6: the contract input;
    # rounds 1-5 started the level-0-only run within ~1 cell, 0.04 m / 0.01 rad, so that every hypothesis converged on the CPU
    # reference -- a reason that went away when the default mode became bit-identical to the reference whether it converges
    # or not) and for the 3-level pyramid run.  The gentle starts are kept as the `gentle_starts` leg of --all-configs.
    init_l0 = init_8d_level0(truth, rank)
    init_gentle = synth.perturb_poses(truth, np.random.default_rng(1238 + 7919 * rank), 0.04, 0.01)
    init_pyr = synth.perturb_poses(truth, np.random.default_rng(1239 + 7919 * rank), 0.15, 0.05)
    pts, offs = synth.pack_scans(scans)
    assert pts.shape[0] == batch * N_BEAMS
    if cfile and os.path.isdir(cache):
        bp, bo = synth.pack_scans(build_scans)
        tmp = cfile + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, build_poses=build_poses, build_pts=bp, build_offs=bo, truth=truth, init_l0=init_l0, init_pyr=init_pyr, pts=pts, offs=offs,
                 init_gentle=init_gentle)
        os.replace(tmp, cfile)
    return build_poses, build_scans, truth, init_l0, init_pyr, pts, offs, init_gentle


def init_8d_level0(truth, rank: int):
    """SURVEY 8(d)'s start errors (+-0.15 m / +-0.05 rad) for the level-0-only headline batch"""
    from hector_slam_amd import synth
    return synth.perturb_poses(truth, np.random.default_rng(1240 + 7919 * rank), 0.15, 0.05)


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


def extra_workload(name: str, args, local_rank: int, rank: int = 0, nranks: int = 1):
    """One of the non-headline BASELINE configs; rank 0 prints one JSON line in the same schema (metric = GN
    iterations/s of that workload; roofline on its matcher launch; reference CPU leg at N = 1).  With N > 1 ranks the
    batched workloads weak-scale (own scans per rank, replicated pyramid, one all-gather of the poses per launch) and
    config5 runs the replicated-map protocol of sharding.ReplicaSync."""
    import torch
    import torch.distributed as dist
    from hector_slam_amd import capi, sharding, synth
    beams, size, res, room, rmax, levels, batch = WORKLOADS[name]
    dev = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    sfac = float(np.float32(1.0) / np.float32(res))
    world = synth.World.make(room[0], room[1], seed=1234)
    rng_noise = np.random.default_rng(1235)
    its = 6 + 4 * (levels - 1)

    def cpu_oracle():
        from oracle import pyoracle
        pyoracle.build()
        kind = "hr" if pyoracle.available("hr") else "ho"
        o = pyoracle.Oracle(kind, res, size, size, levels)
        o.set_update_factor_free(0.4)
        o.set_update_factor_occupied(0.9)
        return o, ("reference" if kind == "hr" else "port")

    out = {"metric": "scan-match GN iterations/sec", "unit": "GN it/s", "n_gpus": nranks, "steps": args.steps,
           "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic"}

    if name == "config5":
        # trajectory: every step = matchData (14 GN it over ~16k beams) + updateByScan on all 3 levels, via
        # the host C ABI exactly as HectorSlamProcessor::update drives it (zero thresholds: always update)
        T = args.warmup + args.steps
        n_init = 8  # scans mapped at their true poses first, so that the matching starts well conditioned
        allp = synth.loop_trajectory(world, 40 * (T + n_init))[: T + n_init + 1].astype(np.float32)  # ~0.4 m apart
        alls = [synth.make_scan(world, p, beams, sfac, rng_noise, range_max=rmax) for p in allp]
        poses, scans = allp[n_init:], alls[n_init:]
        # update-heavy single-scan use: the plane layout (4 gathers per beam, no texel plane to maintain)
        lay = capi.LAYOUT_QUAD if os.environ.get("HSM_LAYOUT") == "quad" else capi.LAYOUT_PLANE
        # N > 1 (configs[4] on a node): one dense scan does not shard -- every rank holds a replica of the pyramid,
        # rank 0 matches, ONE broadcast carries pose + scan, every rank replays the (deterministic) update
        sync = sharding.ReplicaSync(beams, dev) if nranks > 1 else None
        lib = capi.load_library()

        def run_traj(parity=None):
            """the whole trajectory on a fresh context in the given parity mode (None = the library default); -> context, poses, s"""
            m = capi.MapRepMultiMap(res, size, size, levels, device=local_rank, layout=lay, **({} if parity is None else {"parity": parity}))
            m.setUpdateFactorFree(0.4)
            m.setUpdateFactorOccupied(0.9)
            for k in range(n_init + 1):
                m.matchData(allp[k], alls[k])      # retains the coarse-level containers (result unused)
                m.updateByScan(alls[k], allp[k])
                m.onMapUpdated()
            pose = poses[0]
            gpu_poses = []
            for t in range(1, T + 1):
                if t == args.warmup + 1:
                    m.synchronize()
                    if nranks > 1:
                        dist.barrier()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                if rank == 0:
                    hint = pose + (poses[t] - poses[t - 1])
                    pose, _ = m.matchData(hint, scans[t])
                    scan_t = scans[t]
                    if sync:
                        sync.broadcast(pose, scan_t)
                else:
                    pose, scan_t = sync.broadcast(None, None)
                    a = np.ascontiguousarray(scan_t, np.float32)  # what rank 0's matchData retained for the coarse levels
                    capi._check(lib.hsm_retain_scan(m._h, a.ctypes.data, a.shape[0], np.zeros(2, np.float32)), "hsm_retain_scan")
                m.updateByScan(scan_t, pose)     # returns when queued; the next matchData waits behind it
                m.onMapUpdated()
                gpu_poses.append(pose)
            m.synchronize()  # the last update is only queued when updateByScan returns
            if nranks > 1:
                dist.barrier()
            return m, gpu_poses, time.perf_counter() - t0

        def match_alone(m, gpu_poses):
            """matchData alone on the finished map (device idle before each call): median host-call seconds"""
            tm = []
            for t in range(max(1, T - 9), T + 1):
                m.synchronize()
                a = time.perf_counter()
                m.matchData(gpu_poses[t - 1], scans[t])
                tm.append(time.perf_counter() - a)
            return float(np.median(tm))

        m, gpu_poses, dt = run_traj(capi.PARITY_FAST if os.environ.get("HSM_BENCH_CONFIG5_PARITY") == "fast" else None)
        if args.leg == "pmc":  # counter pass of the parent: the launches above are all it wants
            m.close()
            torch.cuda.synchronize()
            if os.environ.get("HSM_BENCH_OS_EXIT") == "1":  # (diagnosis of the rc=-11 exits under rocprofv3, profiles/r04/README.md)
                sys.stdout.flush()
                os._exit(0)
            return
        if nranks > 1:
            out["ranks"] = multi_rank_record(dt, 0.0, dev)
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            dig = [sharding.map_digest(*m.download_level(lvl)) for lvl in range(levels)]
            out["replicas"] = {"protocol": "rank 0 matchData -> broadcast [pose, n, scan] (one RCCL broadcast per step) -> "
                                           "updateByScan replayed on every rank",
                               "maps_identical_across_ranks": bool(sync.digests_equal(dig)), "level_digests_rank0": dig}
            out["scaling"] = "strong"  # one SLAM instance: total work does not grow with N (replicas only)
        if rank != 0:
            return
        # attribution: matchData alone on the finished map (device idle before each call); the update's share
        # of a step is the rest
        t_match = match_alone(m, gpu_poses) * args.steps
        t_upd = dt - t_match
        nb = float(np.mean([s_.shape[0] for s_ in scans[1:]]))
        # what one updateByScan touches (16 B per distinct touched cell + 8 B per beam): one more update,
        # then count the cells that carry its two stamps (OccGridMapBase.h:167: currUpdateIndex + 1 / + 2), per level
        m.matchData(gpu_poses[-1], scans[T])
        m.updateByScan(scans[T], gpu_poses[-1])
        m.synchronize()
        touched, boxes = [], []
        for lvl in range(levels):
            _, ui = m.download_level(lvl)
            touched.append(int((ui >= int(ui.max()) - 1).sum()))
            bb = m.last_update_bbox(lvl)
            boxes.append(int(max(0, bb[2] - bb[0] + 1) * max(0, bb[3] - bb[1] + 1)))
            del ui
        upd_alg_bytes = 16 * sum(touched) + 8 * int(nb) * levels
        out.update({"value": args.steps * its / dt, "ms_per_step": dt / args.steps * 1e3,
                    "config": {"workload": f"configs[4] (one replica): dense {beams}-beam scans (mean {nb:.0f} valid), "
                                           f"{size}^2 map, {levels} levels, matchData + updateByScan interleaved",
                               "beams": beams, "map": size, "levels": levels, "gn_iterations_per_scan": its,
                               "kernel": m.last_launch_config()},
                    "match_ms": t_match / args.steps * 1e3, "update_ms": t_upd / args.steps * 1e3,
                    "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9, "traffic": None,
                                 "achieved": algorithmic_bytes_per_iteration(int(nb)) * its / (t_match / args.steps) / 1e9,
                                 "frac": algorithmic_bytes_per_iteration(int(nb)) * its / (t_match / args.steps) / HBM_PEAK,
                                 "note": "matchData: host-call latency of ONE scan (cooperative launch), not a throughput kernel"}})
        # the update is 3/4 of a step: its own roofline -- algorithmic bytes of one updateByScan (all levels) against the
        # summed duration and the summed HBM traffic of its kernels, from counter passes around `--workload config5 --leg pmc`
        upd = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9,
               "algorithmic_bytes_per_update": upd_alg_bytes, "touched_cells_per_level": touched, "dense_box_cells_per_level": boxes,
               "dense_box_over_touched": sum(boxes) / max(sum(touched), 1), "beams": int(nb),
               "achieved": upd_alg_bytes / (t_upd / args.steps) / 1e9, "frac": upd_alg_bytes / (t_upd / args.steps) / HBM_PEAK,
               "time_basis": "update_ms of the step (host timed: step - matchData)", "traffic": None, "kernels": None}
        if not args.no_pmc and nranks == 1 and not under_profiler():
            names = ["update_mark_occ_dense_kernel", "update_mark_occ_kernel", "update_mark_free_dense_kernel", "update_apply_dense_kernel", "update_mark_free_kernel",
                     "update_mark_kernel", "update_apply_kernel", "update_texels_kernel", "gn_match_coop_kernel", "gn_match_exact_dense_kernel", "gn_match_kernel"]
            pv, perr = pmc_collect(["--workload", "config5", "--leg", "pmc", "--no-cpu", "--no-pmc"], names, warmup=2)
            pmc_dump(args.pmc_dump, "config5", pv, perr, "configs[4] replica: 16 k-beam scans on the 8192^2 pyramid, match + update per step (plane layout)")
            if pv:
                ks, tot_ns, tot_hbm = {}, 0.0, 0.0
                for k, v in pv.items():
                    h = hbm_block(v, None, max(v.get("avg_ns", 0.0), 1.0) * 1e-9)
                    ks[k] = {"avg_us": v.get("avg_ns", 0.0) / 1e3, "launches": v.get("avg_ns_launches"),
                             "hbm_bytes_per_launch": h["bytes_per_launch"] if h else None,
                             "hbm_GBps": h["achieved_GBps"] if h else None,
                             "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_WAVES": v.get("SQ_WAVES")}
                    if k.startswith("update_") and h:
                        tot_ns += v.get("avg_ns", 0.0)
                        tot_hbm += h["bytes_per_launch"]
                upd["kernels"] = ks
                if tot_ns > 0:
                    upd.update({"traffic": tot_hbm, "traffic_over_algorithmic": tot_hbm / upd_alg_bytes,
                                "kernel_time_us": tot_ns / 1e3, "achieved": upd_alg_bytes / (tot_ns * 1e-9) / 1e9,
                                "frac": upd_alg_bytes / (tot_ns * 1e-9) / HBM_PEAK,
                                "hbm_frac_measured": tot_hbm / (tot_ns * 1e-9) / HBM_PEAK,
                                "time_basis": "summed average duration of the update kernels (rocprofv3 kernel trace of the counter passes)"})
            if perr:
                upd["pmc_errors"] = perr
        out["update_roofline"] = upd
        out["config"]["parity_mode"] = f"library default (HSM_PARITY_AUTO) -> {m.last_launch_config().get('parity_effective')} summation (single-scan entry point)"
        if nranks == 1 and not args.no_exact and m.last_launch_config().get("parity_effective") == "exact":
            # the opt-in tree summation beside it: the multi-workgroup dense matcher (HSM_PARITY_FAST), same trajectory, fresh context
            m.close()
            mf, poses_f, dtf = run_traj(capi.PARITY_FAST)
            tmf = match_alone(mf, poses_f)
            dd = np.abs(np.asarray(poses_f, np.float64) - np.asarray(gpu_poses, np.float64))
            out["fast_mode"] = {"mode": "HSM_PARITY_FAST (opt-in): tree summation, K <= 64 cooperating workgroups per dense scan",
                                "value": args.steps * its / dtf, "ms_per_step": dtf / args.steps * 1e3, "match_ms": tmf * 1e3,
                                "update_ms": (dtf / args.steps - tmf) * 1e3, "kernel": mf.last_launch_config(),
                                "max_abs_dxy_m_vs_default": float(dd[:, :2].max()), "max_abs_dtheta_vs_default": float(dd[:, 2].max())}
            m = mf
        if not args.no_cpu and nranks == 1:
            o, kind = cpu_oracle()
            o.proc_set_thresholds(0.0, 0.0)
            for k in range(n_init + 1):
                o.match(allp[k], alls[k])
                o.update_by_scan(allp[k], alls[k])
                o.on_map_updated()  # HectorSlamProcessor.h:93 -- the reference's probability cache must be dropped
            pose = poses[0]
            n_cpu = min(T, 3 if args.compact else 12)
            dmax = 0.0
            t0 = time.perf_counter()
            for t in range(1, n_cpu + 1):
                hint = pose + (poses[t] - poses[t - 1])
                pose, _ = o.match(hint, scans[t])
                o.update_by_scan(pose, scans[t])
                o.on_map_updated()
                dmax = max(dmax, float(np.abs(pose[:2].astype(np.float64) - gpu_poses[t - 1][:2]).max()))
            dtc = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": n_cpu * its / dtc, "unit": "GN it/s", "cores": 1, "kind": kind,
                                   "sample": f"{n_cpu} match+update steps of the same trajectory, {dtc:.1f} s",
                                   "ms_per_step": dtc / n_cpu * 1e3, "max_abs_dxy_m_vs_gpu": dmax}
        emit(out)
        return

    # map built from ground-truth posed scans by the product's own update kernels
    n_build = 100
    build_poses = synth.loop_trajectory(world, n_build).astype(np.float32)
    build_scans = [synth.make_scan(world, p, beams, sfac, rng_noise, range_max=rmax) for p in build_poses]
    m = capi.MapRepMultiMap(res, size, size, levels, device=local_rank)
    m.setUpdateFactorFree(0.4)
    m.setUpdateFactorOccupied(0.9)
    m.build_map(build_poses, build_scans)
    rng = np.random.default_rng(1236 + 7919 * rank)  # every rank matches its own scans
    nq = max(batch, 64)
    base = synth.loop_trajectory(world, nq, phase=rng.uniform(0, 2 * math.pi)).astype(np.float64)
    base[:, :2] += rng.uniform(-0.5, 0.5, size=(nq, 2)) * (room[0] / 40.0)
    base[:, 2] += rng.uniform(-0.3, 0.3, size=nq)
    truth = base.astype(np.float32)
    rng_q = np.random.default_rng(1237 + 7919 * rank)
    scans = [synth.make_scan(world, p, beams, sfac, rng_q, pad_to_full=True, range_max=rmax) for p in truth]
    init = synth.perturb_poses(truth, np.random.default_rng(1239 + 7919 * rank), 0.15 if levels > 1 else 0.04,
                               0.05 if levels > 1 else 0.01)
    pts, offs = synth.pack_scans(scans)

    if name == "config2" and args.leg == "pmc":  # counter pass of the parent: the match + update cycle, nothing else
        for k in range(60):
            q = k % len(build_scans)
            m.matchData(build_poses[q], build_scans[q])
            m.updateByScan(build_scans[q], build_poses[q])
            m.onMapUpdated()
        m.synchronize()
        return
    if name == "config2":
        # one scan at a time through the host entry (what the ROS node calls): latency
        lat = []
        for k in range(args.warmup + args.steps):
            q = k % nq
            a = time.perf_counter()
            pg, _ = m.matchData(init[q], scans[q])
            lat.append(time.perf_counter() - a)
        lat = np.array(lat[args.warmup:])
        default_cfg = m.last_launch_config()
        # the same call with HSM_PARITY=fast (tree summation; opt-in since round 5 -- the default above runs the reference's
        # summation order: nine sequential chains of n additions per GN step) and with HSM_PARITY=exact (the literal serial
        # chains; AUTO may pick any form that is bit-identical to them)
        for mode, key in ((capi.PARITY_FAST, "fast_single_scan_latency_us"), (capi.PARITY_EXACT, "exact_single_scan_latency_us")):
            m.set_parity(mode)
            lat_x = []
            for k in range(10 + min(args.steps, 100)):
                q = k % nq
                a = time.perf_counter()
                m.matchData(init[q], scans[q])
                lat_x.append(time.perf_counter() - a)
            out[key] = {"median": float(np.median(lat_x[10:])) * 1e6, "p90": float(np.percentile(lat_x[10:], 90)) * 1e6,
                        "kernel": m.last_launch_config()}
        m.set_parity(capi.PARITY_AUTO)
        # the other half of HectorSlamProcessor::update: updateByScan on all levels + onMapUpdated, host call
        m2 = capi.MapRepMultiMap(res, size, size, levels, device=local_rank)
        m2.setUpdateFactorFree(0.4)
        m2.setUpdateFactorOccupied(0.9)
        # updateByScan returns once its kernels are queued; the next call on the context waits behind them.
        # "call" = host time of updateByScan + onMapUpdated, "complete" = the same + hsm_synchronize,
        # "cycle" = one full HectorSlamProcessor::update (matchData + updateByScan + onMapUpdated) back to back
        ulat, ucomp, cyc = [], [], []
        nrep = min(args.steps, 400) + 10
        for k in range(nrep):
            q = k % len(build_scans)
            m2.matchData(build_poses[q], build_scans[q])
            a = time.perf_counter()
            m2.updateByScan(build_scans[q], build_poses[q])
            m2.onMapUpdated()
            b = time.perf_counter()
            m2.synchronize()
            ulat.append(b - a)
            ucomp.append(time.perf_counter() - a)
        m2.synchronize()
        for k in range(nrep):
            q = k % len(build_scans)
            a = time.perf_counter()
            m2.matchData(build_poses[q], build_scans[q])
            m2.updateByScan(build_scans[q], build_poses[q])
            m2.onMapUpdated()
            cyc.append(time.perf_counter() - a)
        m2.synchronize()
        stat = lambda v: {"median": float(np.median(v[10:])) * 1e6, "p90": float(np.percentile(v[10:], 90)) * 1e6}
        out["update_latency_us"] = stat(ulat)
        out["update_complete_us"] = stat(ucomp)
        out["slam_cycle_us"] = stat(cyc)
        # the same cycle where the ROS node sits: the reference's unchanged HectorSlamProcessor::update() in C++, once on
        # the reference's CPU map representation and once on the drop-in facade (no Python in the timed calls)
        try:
            import subprocess
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "node_cycle_bench.py"), "200" if args.compact else "400"],
                               capture_output=True, text=True, timeout=240)
            out["node_loop_cpp"] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:  # the two drivers are prebuilt where /root/reference exists
            out["node_loop_cpp"] = {"error": str(e)[:200]}
        out.update({"value": its / float(np.median(lat)), "ms_per_step": float(np.median(lat)) * 1e3,
                    "config": {"workload": f"configs[1]: ONE {beams}-beam scan, {levels}-level {size}/{size // 2}/{size // 4} "
                                           f"pyramid, hsm_match host call (H2D + 1 launch + D2H), median of {args.steps}",
                               "beams": beams, "map": size, "levels": levels, "gn_iterations_per_scan": its,
                               "parity_mode": f"library default (HSM_PARITY_AUTO) -> {default_cfg.get('parity_effective')} summation",
                               "kernel": default_cfg},
                    "latency_us": {"median": float(np.median(lat)) * 1e6, "p90": float(np.percentile(lat, 90)) * 1e6,
                                   "min": float(lat.min()) * 1e6},
                    "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK / 1e9, "traffic": None,
                                 "achieved": algorithmic_bytes_per_iteration(beams) * its / float(np.median(lat)) / 1e9,
                                 "frac": algorithmic_bytes_per_iteration(beams) * its / float(np.median(lat)) / HBM_PEAK,
                                 "note": "single-scan latency is launch/PCIe bound by construction"}})
        if not args.no_pmc and not under_profiler():
            # the node's cycle kernel by kernel: duration, instructions, HBM bytes (counter passes around `--leg pmc`)
            names = ["gn_match_kernel", "update_mark_kernel", "update_apply_kernel", "update_texels_kernel"]
            pv, perr = pmc_collect(["--workload", "config2", "--leg", "pmc", "--no-cpu", "--no-pmc"], names, warmup=5)
            pmc_dump(args.pmc_dump, "config2", pv, perr, "configs[1]: one 1081-beam scan on the 3-level 1024^2 pyramid, match + update cycle")
            if pv:
                out["roofline"]["kernels"] = {
                    k: {"avg_us": v.get("avg_ns", 0.0) / 1e3, "launches": v.get("avg_ns_launches"),
                        "SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "SQ_WAVES": v.get("SQ_WAVES"),
                        "hbm_bytes_per_launch": (hbm_block(v, None, 1.0) or {}).get("bytes_per_launch")} for k, v in pv.items()}
                mk = pv.get("gn_match_kernel")
                if mk and mk.get("avg_ns"):
                    alg = algorithmic_bytes_per_iteration(beams) * its
                    h = hbm_block(mk, alg, mk["avg_ns"] * 1e-9)
                    out["roofline"].update({"kernel": "gn_match_kernel (4 waves, one CU: 14 dependent GN steps)", "kernel_us": mk["avg_ns"] / 1e3,
                                            "achieved": alg / (mk["avg_ns"] * 1e-9) / 1e9, "frac": alg / (mk["avg_ns"] * 1e-9) / HBM_PEAK,
                                            "traffic": h["bytes_per_launch"] if h else None})
            if perr:
                out["roofline"]["pmc_errors"] = perr
        if not args.no_cpu:
            o, kind = cpu_oracle()
            o.build_map(build_poses, build_scans)
            for q in range(8):
                o.match(init[q], scans[q])
            t0 = time.perf_counter()
            n_cpu = 500 if args.compact else 2000
            for k in range(n_cpu):
                o.match(init[k % nq], scans[k % nq])
            dtc = time.perf_counter() - t0
            pairs = [(o.match(init[q], scans[q])[0], m.matchData(init[q], scans[q])[0]) for q in range(min(nq, 64))]
            d = max(float(np.abs(a.astype(np.float64) - b).max()) for a, b in pairs)
            same = float(np.mean([bool((a.view(np.uint32) == b.view(np.uint32)).all()) for a, b in pairs]))
            out["cpu_baseline"] = {"value": n_cpu * its / dtc, "unit": "GN it/s", "cores": 1, "kind": kind,
                                   "sample": f"{n_cpu} matchData calls, warm cache, {dtc:.1f} s",
                                   "latency_us": dtc / n_cpu * 1e6, "max_abs_dev_vs_gpu": d, "parity_sample": len(pairs),
                                   "bit_identical_pose_fraction": same}
        emit(out)
        return

    # batched workloads (config3pyr, config4); N > 1: weak scaling, one all-gather of the [B,3] poses per launch
    B = batch
    d_init = torch.from_numpy(init).to(dev)
    d_pts = torch.from_numpy(pts).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    d_pose = torch.zeros((B, 3), dtype=torch.float32, device=dev)
    d_cov = torch.zeros((B, 9), dtype=torch.float32, device=dev)

    # N > 1: ONE gather per batched match through the device-side exchange (--gather direct, the default), as the headline path
    direct = sharding.DirectRowGather(B * nranks, 3, dev, lag=1) if nranks > 1 and args.gather == "direct" else None

    def timed(steps, warmup):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gatherer = direct if direct is not None else (sharding.BucketedRowGather(B, 3, dev, bucket=args.gather_bucket)
                                                      if nranks > 1 and args.gather == "rccl" else None)

        def step():
            pose_buf = gatherer.next_local() if gatherer else d_pose
            m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), beams, pose_buf.data_ptr(),
                                 d_cov.data_ptr(), stream.cuda_stream)
            if gatherer:
                gatherer.launch()

        if args.prewarm_ms > 0 and args.leg != "pmc":  # engine clock settling (see run() of the headline path); no collective here
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < args.prewarm_ms * 1e-3:
                for _ in range(10):
                    m.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), beams, d_pose.data_ptr(),
                                         d_cov.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        if gatherer:
            gatherer.flush()
        if nranks > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record(stream)
        for k in range(steps):
            step()
        ev1.record(stream)
        if gatherer:
            gatherer.flush()
            gatherer.wait_all()
        if nranks > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if gatherer:
            allp = gatherer.last_result()
            d_pose.copy_(allp[rank * B:(rank + 1) * B])
            if direct is not None:
                direct.check()
        if nranks > 1:
            timed.ranks = multi_rank_record(dt, ev0.elapsed_time(ev1) / steps, dev, allp if gatherer else None)
            timed.ranks["gather"] = ("direct: hsm_exchange, one per batched match, no collective on the data path" if direct is not None else
                                     f"{args.gather}" + (f", {args.gather_bucket} matches per collective" if args.gather == "rccl" else ""))
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, ev0.elapsed_time(ev1) / steps  # back-to-back launches: average duration per launch

    if args.leg == "pmc":  # counter pass of the parent: fast-mode launches, then exact-mode launches
        m.set_parity(capi.PARITY_FAST)
        timed(args.steps, 3)
        m.set_parity(capi.PARITY_EXACT)
        timed(max(3, args.steps // 2), 2)
        return
    # `value` is the DEFAULT mode (HSM_PARITY_AUTO: exact summation for batches on maps above 2^23 cells, else fast); the
    # fast tree is timed first and reported beside it
    m.set_parity(capi.PARITY_FAST)
    dt, kern_ms = timed(args.steps, args.warmup)
    bytes_per_launch = algorithmic_bytes_per_iteration(beams) * its * B
    gpu_pose = d_pose.cpu().numpy()
    cfg = m.last_launch_config()
    total = B * nranks
    out.update({"value": total * its * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                "matchdata_per_s": total * args.steps / dt,
                "config": {"workload": f"{name}: batch={B}/GPU concurrent {beams}-beam scans, {levels}-level pyramid on a "
                                       f"{size}^2 map ({res} m cells, {room[0]:.0f} m x {room[1]:.0f} m room)"
                                       + (" = BASELINE configs[3] at 8 GPUs" if name == "config4" else ""),
                           "batch_per_gpu": B, "global_batch": total, "beams": beams, "map": size, "levels": levels,
                           "gn_iterations_per_scan": its, "parallelism": f"dp{nranks}", "kernel": cfg},
                })
    if nranks > 1:
        out["ranks"] = getattr(timed, "ranks", None)
    fast_kernel = "gn_match_cached_kernel" if cfg.get("texel_cache") else "gn_match_kernel"
    pv = perr = None
    if rank == 0 and nranks == 1 and not args.no_pmc and not under_profiler():
        pv, perr = pmc_collect(["--workload", name, "--leg", "pmc", "--no-cpu", "--no-pmc", "--steps", str(min(args.steps, 10))],
                               ["gn_match_exact_cached_kernel", "gn_match_exact_batch_kernel", "gn_match_cached_kernel", "gn_match_kernel"])
        pmc_dump(args.pmc_dump, name, pv, perr, f"{name}: batch of {B} x {beams}-beam scans, {levels}-level {size}^2 pyramid; fast-mode launches, then exact-mode launches")
    clock_hz = m.device_info()["clock_khz"] * 1e3
    out["roofline"] = roofline_block(fast_kernel, kern_ms, bytes_per_launch, beams, its, B, (pv or {}).get(fast_kernel), perr, clock_hz)
    default_is_exact = True  # round 4: HSM_PARITY_AUTO takes the reference's summation order for EVERY batch
    out["fast_mode"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "kernel_ms": kern_ms,
                        "note": "HSM_PARITY_FAST (tree summation)" + ("; NOT the default on this map size" if default_is_exact else " = the default on this map size")}
    if default_is_exact:
        m.set_parity(capi.PARITY_AUTO)
        dta, ka = timed(args.steps, 3)
        auto_pose = d_pose.cpu().numpy().copy()
        out.update({"value": total * its * args.steps / dta, "ms_per_step": dta / args.steps * 1e3, "matchdata_per_s": total * args.steps / dta})
        out["config"]["kernel"] = m.last_launch_config()
        out["fast_mode"]["roofline"] = out["roofline"]  # the line's `roofline` describes the kernel `value` was measured on
        aname = "gn_match_exact_cached_kernel" if out["config"]["kernel"].get("texel_cache") else "gn_match_exact_batch_kernel"
        out["roofline"] = roofline_block(aname, ka, bytes_per_launch, beams, its, B, (pv or {}).get(aname), perr, clock_hz)
        out["roofline"]["what_binds"] = ("VALU instruction issue plus the serial chain jobs of the reference's summation order "
                                         "(gn_match_exact.h): one workgroup barrier per 64-beam round, a 64-deep dependent fp32 chain behind it")
        out["config"]["parity_mode"] = "HSM_PARITY_AUTO -> exact summation (every batch, round 4)"
    if rank == 0 and not args.no_exact:
        m.set_parity(capi.PARITY_EXACT)
        steps_x = max(5, args.steps // 3)
        dtx, kx = timed(steps_x, 2) if nranks == 1 else (None, None)
        if nranks == 1:
            exact_pose = d_pose.cpu().numpy().copy()
            dd = np.abs(gpu_pose.astype(np.float64) - exact_pose)
            xk = m.last_launch_config()
            xname = "gn_match_exact_cached_kernel" if xk.get("texel_cache") else "gn_match_exact_batch_kernel"
            out["exact_parity"] = {"value": B * its * steps_x / dtx, "unit": "GN it/s", "kernel_ms": kx, "kernel": xname,
                                   "roofline": {k: v for k, v in roofline_block(xname, kx, bytes_per_launch, beams, its, B, (pv or {}).get(xname),
                                                                                  None, clock_hz).items()
                                                if k in ("kernel", "kernel_ms", "bound", "unit", "achieved", "peak", "frac", "traffic", "hbm", "valu", "counter_source")},
                                   "fast_vs_exact_all_scans": {
                                       "scans": B, "bit_identical": float((gpu_pose.view(np.uint32) == exact_pose.view(np.uint32)).all(1).mean()),
                                       "within_1e-4": float(((dd[:, :2].max(1) <= 1e-4) & (dd[:, 2] <= 1e-4)).mean()),
                                       "max_abs_dxy_m": float(dd[:, :2].max())}}
        m.set_parity(capi.PARITY_AUTO)
        if default_is_exact and nranks == 1:
            out["exact_parity"]["default_mode_bit_identical_to_exact"] = float((auto_pose.view(np.uint32) == exact_pose.view(np.uint32)).all(1).mean())
    if not args.no_cpu and nranks == 1:
        o, kind = cpu_oracle()
        o.build_map(build_poses, build_scans)
        n_cpu = min(B, 256 if args.compact else 1024)
        o.match_many(init[:64], pts, offs[:65])
        t0 = time.perf_counter()
        cpu_pose = o.match_many(init[:n_cpu], pts, offs[:n_cpu + 1])
        dtc = time.perf_counter() - t0
        d = np.abs(cpu_pose.astype(np.float64) - gpu_pose[:n_cpu])
        out["cpu_baseline"] = {"value": n_cpu * its / dtc, "unit": "GN it/s", "cores": 1, "kind": kind,
                               "sample": f"{n_cpu} matchData calls on the same map + scans, {dtc:.1f} s",
                               "fast_mode_frac_within_1e-4": float((d[:, :2].max(1) <= 1e-4).mean()),
                               "fast_mode_bit_identical": float((cpu_pose.view(np.uint32) == gpu_pose[:n_cpu].view(np.uint32)).all(1).mean())}
        if "exact_parity" in out:
            out["cpu_baseline"]["exact_mode_bit_identical"] = float(
                (cpu_pose.view(np.uint32) == exact_pose[:n_cpu].view(np.uint32)).all(1).mean())
        if default_is_exact:
            da = np.abs(cpu_pose.astype(np.float64) - auto_pose[:n_cpu])
            out["cpu_baseline"]["default_mode_frac_within_1e-4"] = float((da[:, :2].max(1) <= 1e-4).mean())
    if rank == 0:
        emit(out)


# ---- in-run counters: bench.py re-executes itself (`--leg pmc`) under rocprofv3, one pass per counter group -------------
PMC_GROUPS = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"),
              ("TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"))
# (SQ_ACTIVE_INST_VALU is gone from the line since round 4: it counts one quad-cycle per issued instruction whatever the
# instruction's real issue time -- a 1.9-cycle-per-instruction v_mul/v_add stream reads 4.0 "cycles" as well,
# profiles/r04/README.md -- so "cycles_per_wave64_instr" and "valu_active_frac_of_simd_time" measured nothing)


def pmc_dump(directory, tag, vals, errors=None, note=""):
    """raw counter values of one pmc_collect (mean per launch of each kernel) as a small text file: what the roofline
    fractions of the line are computed from, reproducible without parsing this script's JSON (profiles/rNN/pmc_<tag>.txt)"""
    if not directory or not vals:
        return
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, f"pmc_{tag}.txt"), "w") as f:
        f.write(f"# {note}\n# rocprofv3 --kernel-trace [--pmc <group>] around `bench.py --leg pmc ...`, one pass per group; mean per launch\n"
                f"# FETCH_SIZE / WRITE_SIZE in KB (gfx950: HBM bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024); avg_ns from the pass WITHOUT counters\n")
        for k, v in vals.items():
            f.write(f"kernel {k}\n")
            for c in sorted(v):
                f.write(f"  {c} = {v[c]:.6g}\n")
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v and v.get("avg_ns"):
                hbm = 2.0 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024
                f.write(f"  -> hbm_bytes_per_launch = {hbm:.6g}  ({hbm / (v['avg_ns'] * 1e-9) / 1e12:.4f} TB/s = {hbm / (v['avg_ns'] * 1e-9) / HBM_PEAK:.4f} of 8 TB/s)\n")
            if "SQ_INSTS_VALU" in v and v.get("avg_ns"):
                g = v["SQ_INSTS_VALU"] / (v["avg_ns"] * 1e-9) / 1e9
                f.write(f"  -> valu_issue = {g:.1f} G wave64 instr/s = {g / 1228.8:.4f} of 1228.8 G (1024 SIMDs x 2.4 GHz / 2 cycles)\n")
        if errors:
            f.write(f"# errors: {errors}\n")


def under_profiler() -> bool:
    e = os.environ
    return any(k in e for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "ROCPROF_OUTPUT_PATH")) or \
        "rocprof" in e.get("LD_PRELOAD", "")


def run_child(extra_args, timeout_s=300, env=None):
    """a leg of this script in a child process; returns the dict it printed as its last stdout line"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + extra_args
    env = dict(os.environ if env is None else env, HSM_BENCH_CHILD="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": f"child {' '.join(extra_args)} rc={r.returncode}: {r.stderr.strip()[-300:]}"}
    return json.loads(lines[-1])


def pmc_collect(child_args, kernels, warmup: int = 3, timeout_s: int = 300):
    """Counter passes around a child of this script: `rocprofv3 --kernel-trace --pmc <group> -- python bench.py <child_args>`,
    one pass per group of PMC_GROUPS (FETCH_SIZE and WRITE_SIZE do not fit one pass).  `kernels` = substrings of kernel
    names, most specific first; a dispatch is attributed to the first one it contains.  Returns ({key: {counter: mean per
    launch, counter_launches: n, "avg_ns": mean duration from the same passes' kernel trace}}, errors)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"

    def key_of(name):
        for k in kernels:
            if k in name:
                return k
        return None

    vals, errors = {k: {} for k in kernels}, []
    # durations come from a pass WITHOUT counters (group None): under --pmc the dense update kernels run up to 6x longer
    for group in (None,) + tuple(PMC_GROUPS):
        with tempfile.TemporaryDirectory(prefix="hsm_pmc_", dir="/tmp") as d:
            cmd = [rocprof, "--kernel-trace"] + (["--pmc", *group] if group else []) + ["--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__)] + list(child_args)
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
            except subprocess.TimeoutExpired:
                errors.append(f"{group[0] if group else 'kernel-trace'}: timeout")
                continue
            gname = group[0] if group else "kernel-trace"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv" if group else "*kernel_trace.csv"), recursive=True)
            if not files:
                errors.append(f"{gname}: rc={r.returncode} {r.stderr.strip()[-200:]}")
                continue
            if r.returncode != 0:  # (a child that dies in its exit handlers has delivered its output already)
                errors.append(f"{gname}: child rc={r.returncode}, output was written")
            acc = {k: {} for k in kernels}
            for f in (files if group else []):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = key_of(row.get("Kernel_Name", ""))
                        if k is not None:
                            acc[k].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
            dur = {k: [] for k in kernels}
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        k = key_of(row.get("Kernel_Name", ""))
                        if k is not None:
                            dur[k].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
            for k in kernels:
                for c, v in acc[k].items():
                    v = v[warmup:] if len(v) > warmup else v  # first launches touch cold L2 / page tables
                    vals[k][c] = sum(v) / len(v)
                    vals[k][c + "_launches"] = len(v)
                if dur[k] and group is None:
                    v = dur[k][warmup:] if len(dur[k]) > warmup else dur[k]
                    vals[k]["avg_ns"] = sum(v) / len(v)
                    vals[k]["avg_ns_launches"] = len(v)
    return ({k: v for k, v in vals.items() if v} or None), ("; ".join(errors) or None)


def pmc_leg(kernel_names, steps: int = 20, warmup: int = 3, extra=()):
    """mean counter values per launch of the headline child's kernels (`bench.py --leg pmc [extra]`: K launches in the default
    mode, then K in HSM_PARITY_FAST), collected by rocprofv3, one pass per counter group.  -> ({kernel: counters}, errors)"""
    return pmc_collect(["--leg", "pmc", "--steps", str(steps), "--warmup", str(warmup), *extra], list(kernel_names), warmup)


def hbm_block(pmc, algorithmic_bytes, seconds):
    """HBM traffic of one kernel from its FETCH_SIZE / WRITE_SIZE passes (KB; gfx950: reads tallied at half their size)"""
    if not pmc or "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        return None
    hbm = 2.0 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024
    return {"bytes_per_launch": hbm, "FETCH_SIZE_KB": pmc["FETCH_SIZE"], "WRITE_SIZE_KB": pmc["WRITE_SIZE"], "fetch_correction": 2.0,
            "achieved_GBps": hbm / seconds / 1e9, "peak_GBps": HBM_PEAK / 1e9, "frac": hbm / seconds / HBM_PEAK,
            "traffic_over_algorithmic": hbm / algorithmic_bytes if algorithmic_bytes else None}


def roofline_block(kernel_name, kern_ms, bytes_per_launch, beams, its, batch, pmc, pmc_err, clock_hz, sclk_hz=None,
                   committed_profile=None):
    """see the module docstring: VALU-issue utilisation + in-run HBM traffic + the labelled SURVEY 8(d) contract figure"""
    t = kern_ms * 1e-3
    # algorithmic fp32 operations: 51 per beam and GN iteration (25 mul + 26 add/sub, unfused by construction) +
    # ~100 per GN iteration for the 3x3 solve and the pose update; an FMA-capable lane retires 2 per cycle
    flops = (51 * beams + 100) * its * batch
    peak_flops = 256 * 128 * 2 * clock_hz  # 256 CUs x 128 fp32 lanes x 2 (FMA) x clock
    rf = {"kernel": kernel_name, "kernel_ms": kern_ms,
          "bound": "valu", "unit": "G wave64 VALU instr/s", "achieved": None, "peak": 1024 * clock_hz / 2 / 1e9,
          "frac": None, "traffic": None,
          "what_binds": "VALU instruction issue: 61 unfusable fp32/int instructions per beam and GN iteration (bit-exact "
                        "formulation, no FMA), texels and endpoints served from L2 / LDS / VGPRs; not HBM, not MFMA",
          "clock_hz": clock_hz,
          "flops": {"algorithmic_fp32_per_launch": flops, "achieved_tflops": flops / t / 1e12,
                    "peak_tflops_fp32_vector_fma": peak_flops / 1e12, "frac": flops / t / peak_flops},
          "contract": {"bound": "hbm", "algorithmic_bytes_per_launch": bytes_per_launch,
                       "achieved": bytes_per_launch / t / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                       "frac": bytes_per_launch / t / HBM_PEAK,
                       "note": "SURVEY.md 8(d) contract figure: (24 N + 60) B per GN iteration / kernel time.  NOT a "
                               "utilisation: the model counts endpoint re-reads (kept in LDS across the iterations) and "
                               "texel reads (served by L2) as HBM bytes, hence > 1"}}
    if pmc:
        src = "in-run: rocprofv3 --pmc around `bench.py --leg pmc`, one pass per group, mean per launch of this kernel"
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950: reads are tallied at half their size (guide, HBM section)
            hbm = 2.0 * pmc["FETCH_SIZE"] * 1024 + pmc["WRITE_SIZE"] * 1024
            rf["traffic"] = hbm
            rf["hbm"] = {"bytes_per_launch": hbm, "FETCH_SIZE_KB": pmc["FETCH_SIZE"], "WRITE_SIZE_KB": pmc["WRITE_SIZE"],
                         "fetch_correction": 2.0, "achieved_GBps": hbm / t / 1e9, "peak_GBps": HBM_PEAK / 1e9,
                         "frac": hbm / t / HBM_PEAK, "traffic_over_algorithmic": hbm / bytes_per_launch, "source": src}
        if "SQ_INSTS_VALU" in pmc:
            clk = clock_hz
            rf["achieved"] = pmc["SQ_INSTS_VALU"] / t / 1e9
            rf["frac"] = pmc["SQ_INSTS_VALU"] * 2 / (1024 * clk * t)
            rf["valu"] = {"SQ_INSTS_VALU_per_launch": pmc["SQ_INSTS_VALU"], "per_wave": pmc["SQ_INSTS_VALU"] / max(pmc.get("SQ_WAVES", batch), 1),
                          "SQ_INSTS_SALU_per_launch": pmc.get("SQ_INSTS_SALU"),
                          "gathers": {"SQ_INSTS_VMEM_RD_per_launch": pmc.get("SQ_INSTS_VMEM_RD"),
                                      "TCP_TCC_READ_REQ_per_launch": pmc.get("TCP_TCC_READ_REQ_sum"),
                                      "note": "wave-level vector-memory read instructions (a masked texel gather is one) and L1 -> L2 line requests"},
                          "mean_wave_lifetime_us": (pmc["SQ_WAVE_CYCLES"] * 4 / max(pmc.get("SQ_WAVES", batch), 1) / clk * 1e6
                                                    if pmc.get("SQ_WAVE_CYCLES") else None),
                          "full_rate_cycles_per_wave64_instr": 2, "source": src}
    if rf["frac"] is None and committed_profile is None:
        rf["counter_source"] = "none" + (": " + pmc_err if pmc_err else "")
    elif rf["frac"] is None:
        # no counters in this run (nested profiler, rocprofv3 missing, ...): the committed profile of this workload
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", committed_profile, "traffic.json")))[kernel_name]
            rf["traffic"] = tj["hbm_bytes_per_launch"]
            rf["achieved"] = tj["SQ_INSTS_VALU_per_launch"] / t / 1e9
            rf["frac"] = tj["SQ_INSTS_VALU_per_launch"] * 2 / (1024 * clock_hz * t)
            rf["counter_source"] = f"profiles/{committed_profile}/traffic.json (committed PMC profile of this workload; no counters in this run" + \
                (": " + pmc_err if pmc_err else "") + ")"
        except (OSError, KeyError, ValueError):
            rf["counter_source"] = "none" + (": " + pmc_err if pmc_err else "")
    elif pmc_err:
        rf["pmc_errors"] = pmc_err
    if sclk_hz and 0.5e9 < sclk_hz < 3.5e9 and rf.get("achieved"):
        # what the kernel actually got (DVFS): shader-clock ticks / 100 MHz wall-clock ticks over the lifetime of one wave of
        # the last timed launch, read inside the kernel.  `frac` above stays priced at the nominal peak clock.
        rf["clock_measured"] = {"sclk_hz": sclk_hz, "peak_at_measured_clock": 1024 * sclk_hz / 2 / 1e9,
                                "frac_at_measured_clock": rf["achieved"] / (1024 * sclk_hz / 2 / 1e9),
                                "source": "s_memtime vs the 100 MHz wall clock over the lifetime of the wave of scan 0 in the last "
                                          "timed launch (hsm_set_clock_probe)"}
    return rf


def group_leg(args):
    """`--group N`: the C++ single-process deployment shape (hsm_group_*): ONE process, one replica of the map per device,
    persistent worker threads, device-resident shards of 4096 scans per device, the poses of all shards gathered on replica
    0's device -- through the device-side exchange (hsm_exchange_*, the group's default), through RCCL (ncclCommInitAll + one
    grouped ncclAllGather per step, librccl dlopen'ed by the library) and through peer copies.  Prints one JSON line in the
    bench schema (value = the first of those that is available); `gathers` holds all three."""
    import torch
    from hector_slam_amd import capi, synth
    N = args.group
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py --group needs a HIP device")
    devices = [r % ndev for r in range(N)]
    B = args.batch
    build_poses, build_scans, truth, init, init_pyr, pts, offs, _ = make_inputs(0, B)
    grp = capi.MapRepGroup(RESOLUTION, MAP_SIZE, MAP_SIZE, 1, devices)
    grp.set_update_factors(0.4, 0.9)
    shards = []
    for r in range(N):
        grp.member(r).build_map(build_poses, build_scans)
        dev = torch.device("cuda", devices[r])
        init_r = init_8d_level0(truth, r)  # every replica its own hypotheses (SURVEY 8(d)'s start errors)
        shards.append({"init": torch.from_numpy(init_r).to(dev), "pts": torch.from_numpy(pts).to(dev), "offs": torch.from_numpy(offs).to(dev),
                       "init_host": init_r})
    torch.cuda.synchronize()
    rdev = torch.device("cuda", devices[0])
    d_all = torch.zeros((N * B, 3), dtype=torch.float32, device=rdev)
    its = grp.member(0).gn_iterations_per_match()
    counts = [B] * N
    ptrs = ([s_["init"].data_ptr() for s_ in shards], [s_["pts"].data_ptr() for s_ in shards], [s_["offs"].data_ptr() for s_ in shards])

    def step():
        grp.match_batch_device(counts, ptrs[0], ptrs[1], ptrs[2], N_BEAMS, 0, d_all.data_ptr(), 0)

    # what every shard's rows must be: the same shard matched by ONE context (replica 0) the ordinary way
    want = []
    d_tmp = torch.zeros((B, 3), dtype=torch.float32, device=rdev)
    for r in sorted({0, N - 1}):
        ini = torch.from_numpy(shards[r]["init_host"]).to(rdev)
        grp.member(0).match_batch_device(B, ini.data_ptr(), shards[0]["pts"].data_ptr(), shards[0]["offs"].data_ptr(), N_BEAMS, d_tmp.data_ptr(), 0, 0)
        grp.member(0).synchronize()
        want.append((r, d_tmp.cpu().numpy().copy()))
    gathers = {}
    modes = [("direct", capi.GATHER_DIRECT), ("rccl", capi.GATHER_RCCL), ("peer", capi.GATHER_PEER)]
    for name, mode in modes:
        try:
            grp.set_gather(mode)
        except capi.HsmError as e:
            gathers[name] = {"error": str(e)[:300]}
            continue
        for _ in range(args.warmup):
            step()
        grp.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        grp.synchronize()
        dt = time.perf_counter() - t0
        got = d_all.cpu().numpy()
        ok = all(bool((got[r * B:(r + 1) * B].view(np.uint32) == w.view(np.uint32)).all()) for r, w in want)
        gathers[name] = {"value": N * B * its * args.steps / dt, "ms_per_step": dt / args.steps * 1e3,
                         "gathered_rows_bit_identical_to_single_context": ok}
    best = next((gathers[k] for k in ("direct", "rccl", "peer") if "value" in gathers.get(k, {})), {})
    out = {"metric": "scan-match GN iterations/sec (1081-beam, 2048^2 map)", "value": best.get("value"), "unit": "GN it/s", "n_gpus": N,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": best.get("ms_per_step"), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"configs[2] per device: batch={B}/GPU concurrent {N_BEAMS}-beam scans, {MAP_SIZE}^2 map, level-0 matchData = {its} GN it/scan",
                      "parallelism": f"single process, hsm_group of {N} replicas on devices {devices} ({len(set(devices))} distinct)",
                      "batch_per_gpu": B, "global_batch": N * B, "parity_mode": grp.member(0).last_launch_config().get("parity_effective")},
           "gathers": gathers,
           "timing": "host wall clock around K asynchronous hsm_group_match_batch_device calls + hsm_group_synchronize (includes the hand-off to the "
                     "group's persistent worker threads)"}
    grp.close()
    emit(out)


def group_child_from_rank0(args, world, dist):
    """N > 1 under torch.distributed.run: after the timed region rank 0 runs `--group N` in a child process over the same N
    devices (the C++ deployment shape, both gathers) while the other ranks wait on the rendezvous store -- NOT on a GPU
    barrier, whose kernel would sit on the devices the child measures.  Never fatal: errors land in the record."""
    key = "hsm_group_leg_done"
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception as e:
        return {"error": f"no store: {e}"[:200]}
    rank = dist.get_rank()
    if rank != 0:
        try:
            import datetime
            store.wait([key], datetime.timedelta(seconds=420))
        except Exception:
            pass
        return None
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK",
                                                            "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID")}
    try:
        rec = run_child(["--group", str(world), "--steps", str(min(args.steps, 50)), "--warmup", "5", "--batch", str(args.batch)], timeout_s=360, env=env)
    except Exception as e:
        rec = {"error": str(e)[:300]}
    try:
        store.set(key, "1")
    except Exception:
        pass
    return rec


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves -- the command the
    driver uses (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) -- and pass their output through; rank 0's
    JSON line stays the last stdout line."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("HSM_BENCH_SHARE_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: {ndev} HIP device(s) visible.  (HSM_BENCH_SHARE_GPU=1 puts all ranks on device 0 over "
                         "gloo: a functional check of the multi-rank path, not a measurement.)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    raise SystemExit(subprocess.run(cmd).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 200 for the headline workload: ~12 ms, so that the barrier + "
                         "synchronise bracket of the timed region stays below 1 %% of it; 30 for the extras)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 10 / 5)")
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="scans per GPU")
    ap.add_argument("--levels", type=int, default=1, help="pyramid levels of the headline run")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-pyramid", action="store_true", help="skip the 3-level pyramid leg")
    ap.add_argument("--pyramid", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes")
    ap.add_argument("--no-exact", action="store_true", help="skip the HSM_PARITY_EXACT leg")
    ap.add_argument("--streams", type=int, default=4, help="caller-owned streams of the `pipelined` leg")
    ap.add_argument("--gather", default="direct", choices=["direct", "rccl", "none"],
                    help="N > 1: how every rank gets every rank's poses.  direct (default, the contract line): ONE gather per batched match "
                         "through the library's device-side exchange (hsm_exchange_*: each rank stores its [B,3] rows into every rank's "
                         "IPC-mapped mailbox, one small kernel per match on the matcher's stream, waits lag one match behind); rccl: "
                         "torch.distributed all-gathers, --gather-bucket matches per collective; none: no exchange")
    ap.add_argument("--gather-bucket", type=int, default=32,
                    help="--gather rccl (and the labelled `rccl_bucketed` comparison leg of the N > 1 line): batched matches whose poses "
                         "travel in ONE all-gather (1 = a collective per match).  Enqueueing a "
                         "torch.distributed collective costs the host ~45 us, and an RCCL kernel that runs beside a matcher launch takes "
                         "CUs from its one generation of workgroups (+45 us for that launch): measured with the real nccl backend, us per "
                         "step = 103 / 65.5 / 62.0 for buckets of 1 / 8 / >= the region's steps, 59.6 without any gather, 58.5 at N = 1")
    ap.add_argument("--sustain-s", type=float, default=6.0,
                    help="N = 1: seconds of back-to-back headline launches timed as ONE region after the K-step regions (the sustained "
                         "clock, and long enough for a 5-second device monitor to see the GPU busy); 0 = skip")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the multi-stream leg")
    ap.add_argument("--compact", action="store_true",
                    help="extra workloads: the short form the default run embeds (fewer steps, smaller CPU samples)")
    ap.add_argument("--no-configs", action="store_true", help="(--all-configs) skip the legs for the other BASELINE configs")
    ap.add_argument("--all-configs", "--full", dest="all_configs", action="store_true",
                    help="the long run: 8(d)-start leg, relaxed leg, all-cores CPU leg, pyramid, pipelined and the other BASELINE configs, all into "
                         "the details file (the default run keeps the headline, its counters, the fast-mode leg and the 1-thread CPU baseline: ~45 s)")
    ap.add_argument("--no-relaxed", action="store_true", help="skip the HSM_PARITY_RELAXED leg")
    ap.add_argument("--leg", default=None, choices=["pmc", "pyramid", "pipelined", "gentle"],
                    help="internal: a leg of the default run executed in a child process")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps launches each; the median one is reported (timed_regions keeps all)")
    ap.add_argument("--prewarm-ms", type=float, default=40.0,
                    help="untimed launches for this long before the warm-up steps of every timed run (engine clock settling)")
    ap.add_argument("--starts", default="headline", choices=["headline", "gentle"], help="internal (--leg pmc): start errors of the counter pass")
    ap.add_argument("--pmc-dump", default=None, help="directory for pmc_<leg>.txt files with the raw counter values of this run")
    ap.add_argument("--group", type=int, default=0,
                    help="N > 0: the single-process C++ deployment shape (hsm_group of N replicas, RCCL and peer gathers); see group_leg")
    ap.add_argument("--no-group", action="store_true", help="--gpus N > 1: skip the hsm_group child leg rank 0 runs after the timed region")
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS),
                    help="config3 = the headline (BASELINE configs[2]); others are the extra configs")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 200 if args.workload == "config3" and args.leg != "pyramid" else 30
        if args.workload != "config3" and (args.compact or args.leg == "pmc"):
            args.steps = {"config2": 150, "config5": 6}.get(args.workload, 10)
    if args.warmup is None:
        args.warmup = 10 if args.workload == "config3" else 5
        if args.workload == "config5" and (args.compact or args.leg == "pmc"):
            args.warmup = 2

    if args.group > 0:
        if args.steps is None or args.steps > 200:
            args.steps = 50
        return group_leg(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import torch
    import torch.distributed as dist
    from hector_slam_amd import capi, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    if os.environ.get("HSM_BENCH_SHARE_GPU") == "1":
        # debugging aid for 1-GPU boxes: all ranks on device 0 over gloo (RCCL refuses two ranks on one GPU).  Exercises
        # the multi-rank code paths only; the numbers mean nothing.
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # HSM_BENCH_FORCE_DIST=1: a process group even for ONE rank -- the double-buffered RCCL all-gather, the barriers and the
    # rank records of the N > 1 path run on a 1-GPU box through the real "nccl" backend (a gather of one shard)
    multi = world > 1 or os.environ.get("HSM_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if os.environ.get("HSM_BENCH_SHARE_GPU") == "1":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.workload != "config3":
        assert world == 1 or args.workload != "config2", "config2 is the single-scan latency measurement"
        global _DEFER_EMIT
        _DEFER_EMIT = world > 1
        extra_workload(args.workload, args, local_rank, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
            flush_c_stdio()
            _DEFER_EMIT = False
            if _PENDING:
                time.sleep(1.0)  # (see the end of main: the line comes last)
                emit(_PENDING.pop())
        return

    B = args.batch
    if args.leg is None and world == 1 and "HSM_BENCH_INPUT_CACHE" not in os.environ:
        # the child legs of this run load the inputs this process generates (make_inputs)
        import atexit
        import shutil
        import tempfile
        own_cache = tempfile.mkdtemp(prefix="hsm_bench_inputs_", dir="/tmp")
        os.environ["HSM_BENCH_INPUT_CACHE"] = own_cache
        atexit.register(shutil.rmtree, own_cache, ignore_errors=True)
    build_poses, build_scans, truth, init, init_pyr, pts, offs, init_gentle = make_inputs(rank, B)

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

    _direct = {}

    def direct_gatherer():
        """one exchange per process (its set-up is a collective over the process group): reused by every timed run"""
        if "g" not in _direct:
            _direct["g"] = sharding.DirectRowGather(total, 3, dev, lag=1)
        return _direct["g"]

    def run(matcher, d_init, steps, warmup, gather=True, repeats=1):
        """`repeats` timed regions of exactly `steps` launches each, every one bracketed by barrier + synchronize on both sides;
        returns the MEDIAN region (dt, kernel ms per launch) and keeps all of them in run.regions -- boxes settle at 2.0 or
        2.1 GHz, and one 20-step region is a 1 ms sample"""
        its = matcher.gn_iterations_per_match()
        # HIP events on the launch stream: ONE pair around the whole timed region (the launches queue back to
        # back, so elapsed / steps is the matcher's average duration per launch without a marker packet between
        # consecutive kernels; the overlapped all-gather of N > 1 runs on RCCL's own stream)
        # N > 1: the one collective of the path -- an all-gather of the [B,3] poses -- is double buffered and
        # asynchronous, so RCCL moves batch k's poses while the matcher already works on batch k+1
        # (bucketed: enqueueing one torch.distributed all-gather costs the host ~45 us, and its kernel beside a matcher launch
        # breaks that launch's single generation of workgroups -- measured with the real nccl backend, profiles/r05/README.md 7 --
        # so the poses of `--gather-bucket` consecutive batches travel in one collective)
        mode = gather if isinstance(gather, str) else (args.gather if gather else "none")
        if not multi or os.environ.get("HSM_BENCH_NO_GATHER") == "1":
            mode = "none"
        if mode == "direct":
            # ONE gather per batched match, no collective: the exchange kernel behind every matcher launch posts this rank's
            # [B,3] rows into every rank's mailbox and unpacks the batch before (lag 1); drained inside the timed region
            gatherer = direct_gatherer()
        elif mode == "rccl":
            gatherer = sharding.BucketedRowGather(B, 3, dev, bucket=args.gather_bucket)
        else:
            gatherer = None
        run.gather_mode = mode

        def step():
            pose_buf = gatherer.next_local() if gatherer else d_pose
            matcher.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N_BEAMS,
                                       pose_buf.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
            if gatherer:
                gatherer.launch()

        # clock probe (hsm_set_clock_probe): the wave of scan 0 stamps {shader-clock counter, 100 MHz wall clock} at its
        # first GN step and at its end; read after the timed loop = the clock the LAST timed launch ran at
        probe = torch.zeros(4, dtype=torch.int64, device=dev)
        matcher.set_clock_probe(probe.data_ptr())
        # the engine clock needs ~25 ms of load to settle (first 200-launch region of a cold run: 65 us per launch, second 61,
        # then 58.5 -- profiles/r04/README.md): untimed launches until it has, then the W warm-up steps of the contract
        # (kernel launches only -- no collective: the loop is time-based, so ranks run different numbers of iterations)
        if args.prewarm_ms > 0:
            t_pre = time.perf_counter()
            while time.perf_counter() - t_pre < args.prewarm_ms * 1e-3:
                for _ in range(20):
                    matcher.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N_BEAMS,
                                               d_pose.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize()
        for _ in range(warmup):
            step()
        if gatherer:
            gatherer.flush()
        regions = []
        for rep in range(max(1, repeats)):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record(stream)
            for k in range(steps):
                step()
            ev1.record(stream)
            if gatherer:
                gatherer.flush()  # the last, partially filled bucket travels inside the timed region
                gatherer.wait_all()
            if multi:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            dt_local = dt
            if multi:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            regions.append((dt, ev0.elapsed_time(ev1) / steps))
        # (after the LAST region only: the digest arithmetic and host copies of the rank record are milliseconds of other work, and
        # a 20-launch region that follows them runs on a decayed engine clock -- 66-69 instead of 58 us per launch, measured)
        if gatherer:  # every rank holds all poses; keep this rank's own rows for the checks below
            allp = gatherer.last_result()
            d_pose.copy_(allp[rank * B:(rank + 1) * B])
        if multi:
            run.ranks = multi_rank_record(dt_local, ev0.elapsed_time(ev1) / steps, dev, allp if gatherer else None)
            if mode == "direct":
                torch.cuda.synchronize()
                gatherer.check()  # a wait that timed out fails the run here
                run.ranks.update({"gather": "direct: hsm_exchange post + lagged wait, ONE per batched match, no collective on the data path",
                                  "gathers_total": gatherer.launched, "collectives_total": gatherer.collectives,
                                  "mailbox_memory": gatherer.x.memory_kind()})
            elif gatherer:
                run.ranks.update({"gather": f"rccl: torch.distributed all-gather of {gatherer.bucket} matches per collective",
                                  "gather_bucket": gatherer.bucket, "collectives_total": gatherer.collectives})
        order = sorted(range(len(regions)), key=lambda i: regions[i][0])
        dt, kern_ms = regions[order[len(order) // 2]]
        run.regions = {"repeats": len(regions), "steps_each": steps, "prewarm_ms": args.prewarm_ms, "ms_per_step": [r[0] / steps * 1e3 for r in regions],
                       "kernel_ms": [r[1] for r in regions], "reported": "median region",
                       "min_ms_per_step": min(r[0] for r in regions) / steps * 1e3, "max_ms_per_step": max(r[0] for r in regions) / steps * 1e3}
        st = probe.cpu().numpy().astype(np.uint64)
        matcher.set_clock_probe(0)
        run.sclk_hz = None
        if st[1] and st[3] > st[1]:  # (only the quad-layout texel-cache form carries the probe)
            run.sclk_hz = float(st[2] - st[0]) / float(st[3] - st[1]) * 100e6
        return dt, kern_ms, its

    def kernel_of(cfg):
        if cfg.get("kernel"):  # hsm_last_launch_kernel: the library says which kernel ran
            return cfg["kernel"].split(" ")[0]
        if cfg.get("parity_effective", cfg.get("parity")) == "exact":
            return "gn_match_exact_cached_kernel" if cfg.get("texel_cache") else "gn_match_exact_batch_kernel"
        return "gn_match_cached_kernel" if cfg.get("texel_cache") else "gn_match_kernel"

    # ---------------- child legs -------------------------------------------------------------------------------
    if args.leg == "pmc":  # the headline launches only, for the counter passes of the parent: default mode, then the fast tree
        matcher = build_matcher(1)
        d_in = torch.from_numpy(init_gentle).to(dev) if args.starts == "gentle" else d_init_l0
        run(matcher, d_in, args.steps, args.warmup)
        matcher.set_parity(capi.PARITY_FAST)
        run(matcher, d_in, args.steps, args.warmup)
        return
    if args.leg == "gentle":
        # The headline batch from the GENTLE start errors rounds 1-5 quoted (+-0.04 m / +-0.01 rad: the texel cache re-gathers only
        # lanes whose cell changed, so sub-cell starts are the easier input; the headline itself starts from SURVEY 8(d)'s
        # +-0.15 m / +-0.05 rad since round 6).  A child process, so that a kernel trace of the parent holds the headline's launches only.
        matcher = build_matcher(1)
        d_ig = torch.from_numpy(init_gentle).to(dev)
        leg = {"start_error": "+-0.04 m, +-0.01 rad (rounds 1-5's headline input), level 0 only, same 4096 scans"}
        poses_g = {}
        for mode, nm in ((capi.PARITY_AUTO, "default"), (capi.PARITY_FAST, "fast")):
            matcher.set_parity(mode)
            dtg, kg, itsg = run(matcher, d_ig, args.steps, 3, repeats=min(args.repeats, 3))
            poses_g[nm] = d_pose.cpu().numpy().copy()
            leg[nm] = {"value": B * itsg * args.steps / dtg, "kernel_ms": kg, "kernel": kernel_of(matcher.last_launch_config()),
                       "timed_regions": getattr(run, "regions", None)}
        leg["fast_vs_default_all_scans"] = pose_stats(poses_g["fast"], poses_g["default"])
        if not args.no_cpu:
            leg["default"]["parity_vs_cpu"] = cpu_baseline(build_poses, build_scans, init_gentle, pts, offs, poses_g["default"], 1, budget_s=0.0, n_par=512)
        print(json.dumps(leg))
        return
    if args.leg == "pipelined":
        # Independent batches issued round-robin on S caller-owned streams (hsm_match_batch_device is asynchronous on the
        # stream it is given).  One launch of 4096 scans is ONE generation of wavefronts -- one per scan, four per SIMD --
        # so ~16 % of its duration is tail (waves that finish early leave their slots empty) and the early, gather-heavy
        # GN steps of all waves coincide; with several launches in flight the next batch fills those slots and the
        # phases of different batches interleave.  Same kernels, same results (checked bit for bit against stream 0).
        S = max(1, args.streams)
        res = {"streams": S, "unit": "GN it/s", "note": "throughput of INDEPENDENT 4096-scan batches overlapped on several HIP "
               "streams; the headline `value` keeps one launch at a time (the latency of one batch)"}
        for levels, d_init, name in ((1, d_init_l0, "level0"), (3, d_init_pyr, "pyramid")):
            mm = build_matcher(levels)
            its = mm.gn_iterations_per_match()
            streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
            poses = [torch.zeros((B, 3), dtype=torch.float32, device=dev) for _ in range(S)]

            def pstep(k):
                mm.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N_BEAMS,
                                      poses[k % S].data_ptr(), 0, streams[k % S].cuda_stream)
            for k in range(3 * S):
                pstep(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(args.steps):
                pstep(k)
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t0
            same = all(bool(torch.equal(poses[0], p)) for p in poses[1:])
            res[name] = {"value": B * its * args.steps / dtp, "us_per_batch": dtp / args.steps * 1e6, "steps": args.steps,
                         "gn_iterations_per_scan": its, "all_streams_bit_identical": same}
            mm.close()
        res["value"] = res["level0"]["value"]
        print(json.dumps(res))
        return
    if args.leg == "pyramid":  # full 3-level matchData, SURVEY.md 8(d)'s start errors, both parity modes
        m3 = build_matcher(3)
        res = {"levels": 3, "start_error": "+-0.15 m, +-0.05 rad (SURVEY.md 8(d))", "unit": "GN it/s"}
        poses = {}
        for mode, name in ((capi.PARITY_FAST, "fast"), (capi.PARITY_EXACT, "exact")):
            m3.set_parity(mode)
            steps3 = args.steps if mode == capi.PARITY_FAST else max(5, args.steps // 3)
            dt3, k3, its3 = run(m3, d_init_pyr, steps3, 3, gather=False)
            poses[name] = d_pose.cpu().numpy().copy()
            res[name] = {"value": B * its3 * steps3 / dt3, "matchdata_per_s": B * steps3 / dt3, "kernel_ms": k3,
                         "kernel": kernel_of(m3.last_launch_config()), "steps": steps3}
        res["gn_iterations_per_scan"] = its3
        res["value"] = res["exact"]["value"]
        res["value_is"] = "the library default (HSM_PARITY_AUTO -> exact summation for batches); `fast` = HSM_PARITY_FAST beside it"
        if not args.no_cpu:
            for name in ("fast", "exact"):
                res[name]["parity_vs_cpu"] = cpu_baseline(build_poses, build_scans, init_pyr, pts, offs, poses[name], 3,
                                                          budget_s=0.0, n_par=512)
        d = np.abs(poses["fast"].astype(np.float64) - poses["exact"])
        res["fast_vs_exact_all_scans"] = {"scans": B, "within_1e-4": float(((d[:, :2].max(1) <= 1e-4) & (d[:, 2] <= 1e-4)).mean()),
                                          "bit_identical": float((poses["fast"].view(np.uint32) == poses["exact"].view(np.uint32)).all(1).mean()),
                                          "max_abs_dxy_m": float(d[:, :2].max())}
        print(json.dumps(res))
        return

    # ---------------- the headline ---------------------------------------------------------------------------------
    # `value` = the library's DEFAULT mode.  Since round 4 that is HSM_PARITY_AUTO -> the reference's summation order for every
    # batch (bit-identical poses): the scene sweep (profiles/r04/parity_scene_sweep.jsonl) found the fast tree beyond 1e-4 m on
    # some scans of every scene family once the reference's own iteration has not settled.  The fast tree is the `fast_mode` leg.
    d_in = d_init_l0 if args.levels == 1 else d_init_pyr
    h_in = init if args.levels == 1 else init_pyr
    matcher = build_matcher(args.levels)
    dt, kern_ms, its = run(matcher, d_in, args.steps, args.warmup, gather=args.gather, repeats=args.repeats)
    regions = getattr(run, "regions", None)
    headline_sclk = getattr(run, "sclk_hz", None)
    headline_ranks = getattr(run, "ranks", None)
    headline_gather = getattr(run, "gather_mode", "none")
    gpu_pose = d_pose.cpu().numpy().copy()
    cfg = matcher.last_launch_config()
    gather_legs = None
    if multi:
        # beside the contract line (one gather per batched match), labelled: the bucketed RCCL collective of round 5, a collective per
        # match, and no exchange at all -- same launches, same timing bracket (a leg that fails leaves its error, not the line)
        gather_legs = {}
        for name, gm, bucket in (("no_gather", "none", None), ("rccl_bucketed", "rccl", args.gather_bucket), ("rccl_per_match", "rccl", 1),
                                 ("direct_per_match", "direct", None)):
            if gm == headline_gather and (bucket is None or bucket == args.gather_bucket):
                continue
            if gm == "rccl" and os.environ.get("HSM_BENCH_SHARE_GPU") == "1" and name == "rccl_per_match":
                continue  # (gloo stands in for RCCL there: one figure of it is enough)
            keep = args.gather_bucket
            try:
                if bucket is not None:
                    args.gather_bucket = bucket
                dtl, kl, _ = run(matcher, d_in, args.steps, 3, gather=gm, repeats=min(args.repeats, 3))
                gather_legs[name] = {"value": total * its * args.steps / dtl, "ms_per_step": dtl / args.steps * 1e3, "kernel_ms": kl,
                                     **({"matches_per_collective": bucket} if bucket else {})}
            except Exception as e:
                gather_legs[name] = {"error": str(e)[:200]}
            finally:
                args.gather_bucket = keep
        d_pose.copy_(torch.from_numpy(gpu_pose))
    sustained = None
    if rank == 0 and world == 1 and not multi and args.sustain_s > 0 and args.leg is None:
        # >= args.sustain_s seconds of back-to-back launches as ONE region: the clock the device sustains (the K-step regions above
        # are ~1 ms samples behind a 40 ms pre-warm), and a stretch of load a 5-second device monitor cannot miss
        n_s = max(args.steps, int(args.sustain_s / max(kern_ms * 1e-3, 1e-6)))
        hold = args.prewarm_ms
        args.prewarm_ms = 0.0
        dts, ks, _ = run(matcher, d_in, n_s, 0, gather="none", repeats=1)
        args.prewarm_ms = hold
        sustained = {"seconds": dts, "launches": n_s, "ms_per_step": dts / n_s * 1e3, "kernel_ms": ks, "value": B * its * n_s / dts,
                     "sclk_hz": getattr(run, "sclk_hz", None)}
    value = total * its * args.steps / dt
    bytes_per_launch = algorithmic_bytes_per_iteration(N_BEAMS) * its * B
    kernel_name = kernel_of(cfg)
    clock_hz = matcher.device_info()["clock_khz"] * 1e3
    fast_name = "gn_match_cached_kernel"

    pmc_all = pmc_err = None
    want_pmc = rank == 0 and world == 1 and not args.no_pmc and B == BATCH_PER_GPU and args.levels == 1
    single = rank == 0 and world == 1
    fast_leg = None
    if single and not args.no_exact:
        # the fast tree (HSM_PARITY_FAST): the throughput form of rounds 1-3, opt-in since round 4 (timed BEFORE the CPU thread starts)
        matcher.set_parity(capi.PARITY_FAST)
        dtf, kf, _ = run(matcher, d_in, args.steps, 3, repeats=min(args.repeats, 3))
        fast_leg = (dtf, kf, d_pose.cpu().numpy().copy(), matcher.last_launch_config(), getattr(run, "regions", None), getattr(run, "sclk_hz", None))
        matcher.set_parity(capi.PARITY_AUTO)
    # the 1-thread CPU baseline runs on a host thread WHILE the counter passes run in child processes (the C loop releases the
    # GIL; the box has far more cores than the two need): the default run stays within ~45 s of wall clock
    cpu_box = {}
    cpu_thread = None
    if single and not args.no_cpu:
        import threading

        def _cpu():
            try:
                cpu_box["v"] = cpu_baseline(build_poses, build_scans, h_in, pts, offs, gpu_pose, args.levels)
            except Exception as e:  # never lose the line to the baseline leg
                cpu_box["v"] = {"error": str(e)[:300]}
        cpu_thread = threading.Thread(target=_cpu)
        cpu_thread.start()
    if want_pmc:
        if under_profiler():
            pmc_err = "this process already runs under a profiler"
        else:
            pmc_all, pmc_err = pmc_leg(["gn_match_exact_cached_kernel", "gn_match_exact_batch_kernel", fast_name, "gn_match_kernel"])
            pmc_dump(args.pmc_dump, "headline", pmc_all, pmc_err, "configs[2] headline batch (4096 x 1081 beams, 2048^2, level 0, 6 GN it), "
                     "start errors +-0.15 m / +-0.05 rad (SURVEY 8(d)): default mode (exact order) and HSM_PARITY_FAST launches of the same child")
    if cpu_thread is not None:
        cpu_thread.join()
    pmc = (pmc_all or {}).get(kernel_name)
    rf = roofline_block(kernel_name, kern_ms, bytes_per_launch, N_BEAMS, its, B, pmc, pmc_err, clock_hz,
                        sclk_hz=headline_sclk, committed_profile="r05")
    if cfg.get("parity_effective") == "exact":
        rf["what_binds"] = ("VALU instruction issue plus the serial chain jobs of the reference's summation order (gn_match_exact.h): one "
                            "workgroup barrier per 64-beam round, a 64-deep dependent fp32 chain behind it; texels and endpoints "
                            "served from L2 / LDS / VGPRs; not HBM, not MFMA")
    out = {
        "metric": "scan-match GN iterations/sec (1081-beam, 2048^2 map)",
        "value": value, "unit": "GN it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[2]: batch={B}/GPU concurrent {N_BEAMS}-beam scans (distinct pose+scan "
                               f"pairs), {MAP_SIZE}^2 map, {args.levels}-level matchData = {its} GN it/scan, starts +-0.15 m / +-0.05 rad (SURVEY 8(d))",
                   "batch_per_gpu": B, "global_batch": total, "beams": N_BEAMS, "map": MAP_SIZE,
                   "levels": args.levels, "gn_iterations_per_scan": its, "parallelism": f"dp{world}",
                   "parity_mode": f"library default (HSM_PARITY_AUTO) -> {cfg.get('parity_effective')} summation for this launch",
                   "start_error": "+-0.15 m, +-0.05 rad (SURVEY.md 8(d)); the `gentle_starts` leg of --all-configs runs the same batch from "
                                  "rounds 1-5's +-0.04 m / +-0.01 rad",
                   "kernel": cfg},
        "matchdata_per_s": total * args.steps / dt,
        "timed_regions": regions,
        "roofline": rf,
    }
    if sustained:
        out["sustained"] = sustained
    if multi:
        out["ranks"] = headline_ranks
        out["config"]["gather"] = {"direct": "ONE gather per batched match: device-side exchange (hsm_exchange_*), waits lag one match behind, drained inside the timed region",
                                   "rccl": f"torch.distributed all-gather, {args.gather_bucket} matches per collective",
                                   "none": "no exchange"}[headline_gather]
        out["gather_legs"] = gather_legs
        if args.all_configs and not args.no_group and os.environ.get("HSM_BENCH_SHARE_GPU") != "1":
            # the C++ single-process group over the same devices, RCCL gather and peer gather (child of rank 0)
            rec = group_child_from_rank0(args, world, dist)
            if rank == 0:
                out["group_cpp"] = rec
    conv = np.abs(gpu_pose.astype(np.float64) - truth.astype(np.float64))
    out["convergence"] = {"median_abs_err_xy_m": float(np.median(conv[:, :2])),
                          "median_abs_err_theta_rad": float(np.median(conv[:, 2]))}

    exact_pose = gpu_pose if cfg.get("parity_effective") == "exact" else None
    full = bool(args.all_configs)
    if fast_leg is not None:
        dtf, kf, fast_pose, fcfg, fregions, fsclk = fast_leg
        frf = roofline_block(kernel_of(fcfg), kf, bytes_per_launch, N_BEAMS, its, B, (pmc_all or {}).get(kernel_of(fcfg)), None, clock_hz, sclk_hz=fsclk)
        out["fast_mode"] = {"mode": "HSM_PARITY_FAST: lane-strided partial sums + folded wave tree (per-beam terms bit-exact, summation "
                                    "order differs); opt-in since round 4", "value": B * its * args.steps / dtf, "unit": "GN it/s",
                            "kernel_ms": kf, "ms_per_step": dtf / args.steps * 1e3, "kernel": kernel_of(fcfg), "timed_regions": fregions,
                            "roofline": {k: v for k, v in frf.items() if k in ("kernel", "kernel_ms", "bound", "unit", "achieved", "peak", "frac", "traffic",
                                                                               "hbm", "valu", "clock_measured", "counter_source")}}
        if exact_pose is not None:
            out["fast_mode"]["fast_vs_default_all_scans"] = pose_stats(fast_pose, exact_pose)
    if "v" in cpu_box:
        out["cpu_baseline"] = cpu_box["v"]
        out["cpu_baseline"]["concurrent_with"] = "the rocprofv3 counter passes of this run (child processes on other cores)" if want_pmc and not under_profiler() else None
    if full and single and not args.no_exact and args.levels == 1 and B == BATCH_PER_GPU:
        # the same batch from rounds 1-5's gentle start errors (child process: `--leg gentle`), with the counters of its launches
        leg = run_child(["--leg", "gentle", "--steps", str(max(10, args.steps // 2)), "--batch", str(B), "--repeats", str(args.repeats)] +
                        (["--no-cpu"] if args.no_cpu else []))
        if want_pmc and not under_profiler() and "error" not in leg:
            pg, eg = pmc_leg(["gn_match_exact_cached_kernel", fast_name], extra=("--starts", "gentle"))
            pmc_dump(args.pmc_dump, "gentle_starts", pg, eg, "the headline batch from rounds 1-5's start errors (+-0.04 m / +-0.01 rad)")
            for nm in ("default", "fast"):
                v = (pg or {}).get(leg[nm]["kernel"]) or {}
                h = (pmc_all or {}).get(leg[nm]["kernel"]) or {}
                leg[nm]["counters_per_launch"] = {k: v.get(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "TCP_TCC_READ_REQ_sum", "FETCH_SIZE", "WRITE_SIZE", "avg_ns")}
                leg[nm]["same_counters_headline_starts"] = {k: h.get(k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "TCP_TCC_READ_REQ_sum", "avg_ns")}
            if eg:
                leg["pmc_errors"] = eg
        out["gentle_starts"] = leg
    if full and single and not args.no_relaxed and args.levels == 1:
        # HSM_PARITY_RELAXED (opt-in): multiply-add pairs of the per-beam arithmetic contracted; bar = 1e-4 m / 1e-4 rad
        matcher.set_parity(capi.PARITY_RELAXED)
        steps_r = max(10, args.steps // 4)
        dtr, kr, _ = run(matcher, d_init_l0, steps_r, 3)
        relaxed_pose = d_pose.cpu().numpy().copy()
        matcher.set_parity(capi.PARITY_AUTO)
        out["relaxed"] = {"mode": "HSM_PARITY_RELAXED: v_fma_f32 for the rotation, blends, rotDeriv and the nine accumulations (32 "
                                  "instead of 51 fp32 operations per beam); opt-in, the headline `value` stays the default mode",
                          "value": B * its * steps_r / dtr, "unit": "GN it/s", "kernel_ms": kr, "steps": steps_r,
                          "speedup_vs_default": kern_ms / kr,
                          "speedup_vs_fast": (out["fast_mode"]["kernel_ms"] / kr) if "fast_mode" in out else None}
        if exact_pose is not None:
            out["relaxed"]["vs_default_all_scans"] = pose_stats(relaxed_pose, exact_pose)
        if not args.no_cpu:
            out["relaxed"]["parity_vs_cpu"] = cpu_baseline(build_poses, build_scans, init, pts, offs, relaxed_pose, 1, budget_s=0.0, n_par=512)
    if single and not args.no_cpu and args.leg is None:
        # single thread AND all cores in the same run (after the counter passes and the 1-thread leg: nothing else runs)
        try:
            ac = cpu_baseline_all_cores(build_poses, build_scans, h_in, pts, offs, args.levels)
        except Exception as e:
            ac = {"error": str(e)[:200]}
        out["cpu_baseline_all_cores"] = ac
        if "cpu_baseline" in out and "value" in ac:
            out["cpu_baseline"]["all_cores"] = {"value": ac["value"], "cores": ac["cores"], "unit": ac["unit"]}
    if full and single and not args.no_pyramid and args.levels == 1:
        out["pyramid"] = run_child(["--leg", "pyramid", "--steps", str(max(10, args.steps // 4)), "--batch", str(B)] +
                                   (["--no-cpu"] if args.no_cpu else []))
    if full and single and not args.no_pipelined and args.levels == 1:
        out["pipelined"] = run_child(["--leg", "pipelined", "--steps", str(max(40, args.steps)), "--batch", str(B),
                                      "--streams", str(args.streams)])
    if full and single and not args.no_configs and args.levels == 1 and B == BATCH_PER_GPU:
        # the other BASELINE configs in the details file: compact child runs, each with its own counter passes
        matcher.close()
        del matcher
        torch.cuda.empty_cache()
        extra = ["--compact"] + (["--no-cpu"] if args.no_cpu else []) + (["--no-pmc"] if args.no_pmc else []) + \
            (["--pmc-dump", args.pmc_dump] if args.pmc_dump else [])
        cf = {"configs[0]": config1_plumbing(capi) if not args.no_cpu else None}
        for key, wl in (("configs[1]", "config2"), ("configs[3] (one GPU's share)", "config4"), ("configs[4] (one replica)", "config5")):
            cf[key] = run_child(["--workload", wl] + extra, timeout_s=400)
        out["configs"] = cf
    if not full and single:
        out["not_run"] = "the gentle-start, relaxed, pyramid, pipelined and other-config legs: `bench.py --all-configs` (details file)"
    if multi:
        dist.barrier()
        dist.destroy_process_group()
        flush_c_stdio()
        if rank == 0:
            time.sleep(1.0)  # the other ranks exit now (and flush whatever their libraries still hold): the line comes last
    if rank == 0:
        emit(out)


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


if __name__ == "__main__":
    main()

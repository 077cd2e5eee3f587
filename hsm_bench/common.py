"""what every leg of bench.py shares: the workload constants, deterministic synthetic inputs, and the ONE compact JSON line the
driver parses (the full record goes to a details file)."""
from __future__ import annotations

import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")
if ROOT not in sys.path:
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
        cm = rf.get("clock_measured") or {}
        if cm.get("sclk_hz"):  # the shader clock the kernel actually ran at (read inside the kernel); `frac` stays priced at 2.4 GHz
            r["sclk_hz"] = cm["sclk_hz"]
            r["frac_at_measured_clock"] = cm.get("frac_at_measured_clock")
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


def defer_emit(on: bool):
    """N > 1: hold the line back until the process group is gone and the other ranks have exited (then emit_pending())"""
    global _DEFER_EMIT
    _DEFER_EMIT = on


def emit_pending():
    if _PENDING:
        emit(_PENDING.pop())


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

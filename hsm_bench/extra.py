"""the BASELINE configs other than the headline one (`bench.py --workload NAME`)"""
from __future__ import annotations

import json
import math
import os
import sys
import time

import numpy as np

from .common import HBM_PEAK, ROOT, WORKLOADS, algorithmic_bytes_per_iteration, emit, multi_rank_record
from .pmc import hbm_block, pmc_collect, pmc_dump, roofline_block, under_profiler


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
    direct, gather_note = None, ""
    if nranks > 1 and args.gather == "direct":  # (collective decision + self-test; a machine that cannot run it gets the collective)
        g0, kind, gather_note = sharding.make_row_gather(B * nranks, B, 3, dev, lag=1, fallback_bucket=args.gather_bucket)
        if kind == "direct":
            direct = g0
        else:
            args.gather = "rccl"

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
            if direct is not None:  # (a lost epoch is raised on every rank together: one rank leaving alone would hang the others)
                bad, why = 0, ""
                try:
                    direct.check()
                except Exception as exc:
                    bad, why = 1, str(exc)[:200]
                tb = torch.tensor([bad], dtype=torch.int32, device=dev)
                if nranks > 1:
                    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                if int(tb.item()):
                    raise RuntimeError("device-side gather lost an epoch: " + (why or "on a peer"))
        if nranks > 1:
            timed.ranks = multi_rank_record(dt, ev0.elapsed_time(ev1) / steps, dev, allp if gatherer else None)
            timed.ranks["gather"] = ("direct: hsm_exchange, one per batched match, no collective on the data path" if direct is not None else
                                     f"{args.gather}" + (f", {args.gather_bucket} matches per collective" if args.gather == "rccl" else "") +
                                     (f" -- FALLBACK: {gather_note[:200]}" if gather_note else ""))
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

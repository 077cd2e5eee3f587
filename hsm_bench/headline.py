"""the headline leg: BASELINE configs[2] (4096 x 1081-beam scans per GPU on a 2048^2 map) at N >= 1 GPUs, its child legs, and the
record the driver's line is cut from"""
from __future__ import annotations

import json
import math
import os
import sys
import time

import numpy as np

from . import common
from .common import (BATCH_PER_GPU, MAP_SIZE, N_BEAMS, RESOLUTION, algorithmic_bytes_per_iteration, emit, flush_c_stdio, make_inputs,
                     multi_rank_record, pose_stats)
from .cpu import config1_plumbing, cpu_baseline, cpu_baseline_all_cores
from .extra import extra_workload
from .group import group_child_from_rank0
from .pmc import pmc_dump, pmc_leg, roofline_block, run_child, under_profiler



class DirectGatherFailed(RuntimeError):
    """a wait of the device-side pose exchange timed out on some rank (raised on all of them together)"""


def headline(args):
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
        common.defer_emit(world > 1)
        extra_workload(args.workload, args, local_rank, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
            flush_c_stdio()
            common.defer_emit(False)
            time.sleep(1.0)  # (see the end of headline(): the line comes last)
            common.emit_pending()
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
            # (collective decision + self-test: a machine on which the exchange cannot be set up gets the RCCL gather, labelled)
            _direct["g"], _direct["kind"], _direct["note"] = sharding.make_row_gather(total, B, 3, dev, lag=1, fallback_bucket=args.gather_bucket)
        return _direct["g"]

    def run(matcher, d_init, steps, warmup, gather=True, repeats=1, probe_after=False):
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
            if _direct["kind"] != "direct":  # this machine cannot run the exchange: the collective, and the line says so
                mode = "rccl"
                run.gather_note = _direct["note"]
        elif mode == "rccl":
            gatherer = sharding.BucketedRowGather(B, 3, dev, bucket=args.gather_bucket)
        else:
            gatherer = None
        run.gather_mode = mode

        fused = mode == "direct" and os.environ.get("HSM_BENCH_UNFUSED_EXCHANGE") != "1"

        def step():
            if fused:  # ONE call: the matcher launch carries the exchange step (hsm_match_batch_device_gather)
                gatherer.match_and_launch(matcher, B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N_BEAMS, d_cov.data_ptr(), stream)
                return
            pose_buf = gatherer.next_local() if gatherer else d_pose
            matcher.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N_BEAMS,
                                       pose_buf.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
            if gatherer:
                gatherer.launch()

        # clock probe (hsm_set_clock_probe): workgroup 0 stamps {shader-clock counter, 100 MHz wall clock} at its start and at its
        # end.  On for the W warm-up launches only -- the stamps cost the reference-order kernel 0.4 us per launch (its probed
        # instantiation holds four more SGPRs) --, so what is read after the loop is the clock of the LAST WARM-UP launch, the one
        # right in front of the first timed region; the timed launches carry no probe code
        probe = torch.zeros(4, dtype=torch.int64, device=dev)
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
        matcher.set_clock_probe(probe.data_ptr())
        for _ in range(warmup):
            step()
        matcher.set_clock_probe(0)
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
                # a wait that timed out fails the run here -- on EVERY rank (a rank that raised alone would leave the others in
                # the next barrier): the verdicts are reduced first
                bad, why = 0, ""
                try:
                    gatherer.check()
                except Exception as exc:
                    bad, why = 1, str(exc)[:200]
                if os.environ.pop("HSM_BENCH_INJECT_DIRECT_FAILURE", None) == str(rank):  # (test hook: this rank reports a lost epoch, once)
                    bad, why = 1, "injected by HSM_BENCH_INJECT_DIRECT_FAILURE"
                t = torch.tensor([bad], dtype=torch.int32, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                if int(t.item()):
                    raise DirectGatherFailed(why or "a peer's wait of the device-side exchange timed out")
                run.ranks.update({"gather": "direct: hsm_exchange post + lagged wait, ONE per batched match, no collective on the data path; "
                                            + ("carried by the matcher launch itself (epilogue posts, tail workgroups unpack)" if fused else "one small kernel behind every matcher launch"),
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
        if probe_after:  # (the sustained leg: the clock at the END of its one long region -- three probed launches right behind it)
            matcher.set_clock_probe(probe.data_ptr())
            for _ in range(3):
                matcher.match_batch_device(B, d_init.data_ptr(), d_pts.data_ptr(), d_offs.data_ptr(), N_BEAMS,
                                           d_pose.data_ptr(), d_cov.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize()
            matcher.set_clock_probe(0)
        st = probe.cpu().numpy().astype(np.uint64)
        run.sclk_hz = None
        if st[1] and st[3] > st[1]:  # (the texel-cache forms carry the probe: workgroup 0's first wavefront)
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
    try:
        dt, kern_ms, its = run(matcher, d_in, args.steps, args.warmup, gather=args.gather, repeats=args.repeats)
    except DirectGatherFailed as exc:
        # the exchange passed its self-test and then lost an epoch under load: every rank is here (see run), the line is measured
        # with the collective instead and says so
        _direct["kind"], _direct["note"] = "rccl", f"device-side exchange failed during the timed run ({exc})"
        _direct["g"] = sharding.BucketedRowGather(B, 3, dev, bucket=args.gather_bucket)
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
        for name, gm, bucket in (("no_gather", "none", None), ("direct_separate_launch", "direct_unfused", None),
                                 ("rccl_bucketed", "rccl", args.gather_bucket), ("rccl_per_match", "rccl", 1), ("direct_per_match", "direct", None)):
            if gm == headline_gather and (bucket is None or bucket == args.gather_bucket):
                continue
            if gm.startswith("direct") and _direct.get("kind") == "rccl":
                continue  # (this machine could not set the exchange up: the headline already is the collective)
            if gm == "rccl" and os.environ.get("HSM_BENCH_SHARE_GPU") == "1" and name == "rccl_per_match":
                continue  # (gloo stands in for RCCL there: one figure of it is enough)
            keep = args.gather_bucket
            try:
                if bucket is not None:
                    args.gather_bucket = bucket
                if gm == "direct_unfused":  # the same exchange as its own small kernel behind every matcher launch
                    os.environ["HSM_BENCH_UNFUSED_EXCHANGE"] = "1"
                    gm = "direct"
                dtl, kl, _ = run(matcher, d_in, args.steps, 3, gather=gm, repeats=min(args.repeats, 3))
                gather_legs[name] = {"value": total * its * args.steps / dtl, "ms_per_step": dtl / args.steps * 1e3, "kernel_ms": kl,
                                     **({"matches_per_collective": bucket} if bucket else {})}
            except Exception as e:
                gather_legs[name] = {"error": str(e)[:200]}
            finally:
                args.gather_bucket = keep
                os.environ.pop("HSM_BENCH_UNFUSED_EXCHANGE", None)
        d_pose.copy_(torch.from_numpy(gpu_pose))
    sustained = None
    if rank == 0 and world == 1 and not multi and args.sustain_s > 0 and args.leg is None:
        # >= args.sustain_s seconds of back-to-back launches as ONE region: the clock the device sustains (the K-step regions above
        # are ~1 ms samples behind a 40 ms pre-warm), and a stretch of load a 5-second device monitor cannot miss
        n_s = max(args.steps, int(args.sustain_s / max(kern_ms * 1e-3, 1e-6)))
        hold = args.prewarm_ms
        args.prewarm_ms = 0.0
        dts, ks, _ = run(matcher, d_in, n_s, 0, gather="none", repeats=1, probe_after=True)
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
                        sclk_hz=headline_sclk, committed_profile="r06")
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
        if _direct.get("note"):
            out["config"]["gather"] += " -- FALLBACK: " + _direct["note"][:300]
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

"""`bench.py --group N`: the single-process C++ deployment shape (hsm_group_*)"""
from __future__ import annotations

import os
import time

import numpy as np

from .common import MAP_SIZE, N_BEAMS, RESOLUTION, emit, init_8d_level0, make_inputs
from .pmc import run_child


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

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
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# (re-exported: tests/test_bench_line.py and the tools under tools/ reach these through `import bench`)
from hsm_bench.common import (BATCH_PER_GPU, HBM_PEAK, LINE_LIMIT, MAP_SIZE, N_BEAMS, RESOLUTION, WORKLOADS,  # noqa: E402,F401
                              algorithmic_bytes_per_iteration, compact_line, details_file, emit, init_8d_level0, make_inputs, pose_stats)
from hsm_bench import pmc  # noqa: E402,F401
from hsm_bench.pmc import pmc_collect, pmc_dump, run_child  # noqa: E402,F401


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
        from hsm_bench.group import group_leg
        return group_leg(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    from hsm_bench.headline import headline
    return headline(args)


if __name__ == "__main__":
    main()

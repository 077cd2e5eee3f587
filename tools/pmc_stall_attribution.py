#!/usr/bin/env python
"""Where the headline kernel's wavefront cycles go (round-5 verdict item 4): rocprofv3 counter passes over the headline launches
(`bench.py --leg pmc`), one pass per group below, mean per launch of gn_match_exact_cached_kernel (default mode) and
gn_match_cached_kernel (fast tree), written as JSON + a small table.  SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked: s_waitcnt / barrier)
+ SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY (MI355X_MICROARCH.md, "rocprofv3 PMC slots"); all in quad-cycles.
usage: tools/pmc_stall_attribution.py OUT_DIR [--steps K]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

GROUPS = (
    ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"),
    ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT"),
    ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_SENDMSG"),
    ("SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM"),
    ("SQ_INST_CYCLES_VMEM", "SQ_INST_CYCLES_SALU", "SQ_INST_CYCLES_SMEM", "SQ_THREAD_CYCLES_VALU", "SQ_IFETCH", "SQ_VALU_MFMA_BUSY_CYCLES"),
)


def main():
    out_dir = sys.argv[1]
    steps = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "20"
    os.makedirs(out_dir, exist_ok=True)
    try:  # the counter names this rocprofv3 knows (a group with an unknown name fails as a whole)
        r = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120)
        open(os.path.join(out_dir, "rocprofv3_counter_list.txt"), "w").write(r.stdout[-200000:])
        known = r.stdout
    except Exception as e:
        known = ""
        print("rocprofv3 -L:", e, file=sys.stderr)
    groups = []
    for g in GROUPS:
        keep = tuple(c for c in g if not known or c in known)
        if keep:
            groups.append(keep)
        dropped = [c for c in g if c not in keep]
        if dropped:
            print("not offered by this rocprofv3:", dropped, file=sys.stderr)
    bench.pmc.PMC_GROUPS = tuple(groups)
    kernels = ["gn_match_exact_cached_kernel", "gn_match_cached_kernel"]
    vals, err = bench.pmc_collect(["--leg", "pmc", "--steps", steps, "--warmup", "3", "--no-cpu", "--no-pmc"], kernels, warmup=3)
    rec = {"kernels": vals, "errors": err, "groups": groups,
           "note": "mean per launch over the timed launches of `bench.py --leg pmc` (configs[2] headline batch, SURVEY 8(d) starts); SQ_* cycle "
                   "counters are quad-cycles summed over all wavefronts"}
    json.dump(rec, open(os.path.join(out_dir, "pmc_stall_attribution.json"), "w"), indent=1)
    for k, v in (vals or {}).items():
        wc = v.get("SQ_WAVE_CYCLES")
        print(k, "avg_us", round(v.get("avg_ns", 0) / 1e3, 2))
        for c in sorted(v):
            if c.endswith("_launches") or c == "avg_ns":
                continue
            frac = f"  {v[c] / wc:6.3f} of SQ_WAVE_CYCLES" if wc and ("CYCLES" in c or "WAIT" in c or "ACTIVE" in c or "LEVEL" in c) else ""
            print(f"  {c:28s} {v[c]:16.1f}{frac}")
    if err:
        print("errors:", err)


if __name__ == "__main__":
    main()

"""rocprofv3 counter passes around child runs of bench.py, and the roofline blocks built from them"""
from __future__ import annotations

import json
import os
import sys

from .common import BENCH_PY, HBM_PEAK, ROOT


# ---- in-run counters: bench.py re-executes itself (`--leg pmc`) under rocprofv3, one pass per counter group -------------
PMC_GROUPS = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"),
              ("TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"))


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
    cmd = [sys.executable, BENCH_PY] + extra_args
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
                   sys.executable, BENCH_PY] + list(child_args)
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
                                "source": "s_memtime vs the 100 MHz wall clock over the lifetime of the first wavefront of workgroup 0 in the last "
                                          "warm-up launch before the timed regions (hsm_set_clock_probe; timed launches carry no probe)"}
    return rf

#!/bin/bash
# round 5, call A: the default mode is exact on every entry point now.  Suite, smoke, the driver's bench command (does its last
# line parse, < 4 KB?), the same under rocprofv3 --kernel-trace --stats, configs[1] / configs[4] re-quoted in the default mode,
# and the single-scan leg of the scene sweep: every scan of every family / set-up through hsm_match in the default mode.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05a
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -15 "$OUT/pytest.log"
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== driver bench command"; S=$(date +%s)
HSM_BENCH_DETAILS=$OUT/bench_details.json timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-dump "$OUT/pmc" > "$OUT/bench_driver_like.out" 2> "$OUT/bench_driver_like.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/bench_driver_like.err"
python - "$OUT/bench_driver_like.out" <<'PY'
import json, sys
last = open(sys.argv[1]).read().strip().splitlines()[-1]
d = json.loads(last)
print("LINE bytes", len(last)); print(last)
PY
echo "== same command under rocprofv3 --kernel-trace --stats"; S=$(date +%s)
( cd /tmp && export TMPDIR=/tmp && HSM_BENCH_DETAILS=$OUT/bench_details_under_rocprof.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu > "$OUT/bench_under_rocprof.out" 2> "$OUT/stats.err" ); echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT" <<'PY'
import csv, glob, os, shutil, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f))}
    shutil.copy(f, os.path.join(out, "bench_driver_like_kernel_stats.csv"))
    print({n[:70]: v for n, v in rows.items() if "gn_match" in n})
print(open(os.path.join(out, "bench_under_rocprof.out")).read().strip().splitlines()[-1][:600])
PY
echo "== configs[1] (single scan) and configs[4] (dense loop) in the default mode"; S=$(date +%s)
HSM_BENCH_DETAILS=$OUT/bench_config2.json timeout 600 python bench.py --workload config2 --compact --no-pmc 2> "$OUT/config2.err" | tail -1 | cut -c1-1500
HSM_BENCH_DETAILS=$OUT/bench_config5.json timeout 900 python bench.py --workload config5 --compact --pmc-dump "$OUT/pmc" 2> "$OUT/config5.err" | tail -1 | cut -c1-1500
python - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
try:
    d = json.load(open(os.path.join(out, "bench_config2.json")))
    print("config2 default us", d["latency_us"], "fast", d.get("fast_single_scan_latency_us", {}).get("median"), "exact", d.get("exact_single_scan_latency_us", {}).get("median"),
          "cycle", d.get("slam_cycle_us"), "cpu", d.get("cpu_baseline", {}).get("latency_us"), "bitid", d.get("cpu_baseline", {}).get("bit_identical_pose_fraction"))
    print(" node loop", json.dumps(d.get("node_loop_cpp"))[:600])
except Exception as e:
    print("config2", e)
try:
    d = json.load(open(os.path.join(out, "bench_config5.json")))
    print("config5 default ms/step", d["ms_per_step"], "match", d["match_ms"], "update", d["update_ms"], "kernel", d["config"]["kernel"])
    print(" fast", json.dumps(d.get("fast_mode"))[:500])
    u = d["update_roofline"]
    print(" update kernels", {k: round(v["avg_us"], 1) for k, v in (u.get("kernels") or {}).items()}, "traffic/alg", u.get("traffic_over_algorithmic"), "frac", u.get("frac"))
    print(" cpu", d.get("cpu_baseline"))
except Exception as e:
    print("config5", e)
PY
echo "($(( $(date +%s) - S )) s)"
echo "== scene sweep, single-scan leg in the default mode (all scans of every family and set-up)"; S=$(date +%s)
timeout 1500 python tools/parity_scene_sweep.py --batch 4096 --sample 256 --single-all --no-single --out "$OUT/parity_scene_sweep_single_default.jsonl" 2>&1 | grep -v '^$' | tail -60
echo "($(( $(date +%s) - S )) s)"

#!/bin/bash
# round 4, call D: suite; dense line walk with per-beam records (+ the 2x unrolled variant): correctness + kernel times;
# default bench line (+ pmc dumps) and the same command under rocprofv3 --kernel-trace --stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04d
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=6 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -12 "$OUT/pytest.log"
echo "== update-kernel variants"; S=$(date +%s)
for v in u2; do
  lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  HSM_LIB=$lib timeout 600 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_full_size.py tests/test_property_hypothesis.py -k "dense or config5" -m gpu -q 2>&1 | tail -2 | sed "s/^/$v tests: /"
done
bash tools/exp_update_kernels.sh base u2 2>&1 | tee "$OUT/update_variants_kernel_us.txt"
echo "($(( $(date +%s) - S )) s)"
echo "== soak (dense, 1500 free-running exact steps) on the new walk"; S=$(date +%s)
timeout 600 python tests/tools/soak_dense.py 1500 --check 500 > "$OUT/soak_dense_recs_1500.json" 2> /dev/null; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-300 "$OUT/soak_dense_recs_1500.json"
echo "== default bench line"; S=$(date +%s)
timeout 900 python bench.py --pmc-dump "$OUT/pmc" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], d["timed_regions"]["ms_per_step"])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "kernel_ms", "frac", "achieved", "traffic", "counter_source", "pmc_errors")})
print("fast", round(d["fast_mode"]["value"] / 1e6, 1), d["fast_mode"]["kernel_ms"], d["fast_mode"].get("fast_vs_default_all_scans"))
print("8d", json.dumps(d.get("headline_8d_starts"))[:900])
for k, v in (d.get("configs") or {}).items():
    v = v or {}
    print(k, "value", v.get("value"), "ms", v.get("ms_per_step"), "match/update", v.get("match_ms"), v.get("update_ms"), "err", v.get("error"))
c = d["configs"]["configs[4] (one replica)"]["update_roofline"]
print("update", {k: round(v["avg_us"], 1) for k, v in c["kernels"].items()}, "traffic/alg", c.get("traffic_over_algorithmic"), "frac", c.get("frac"))
PY
echo "== same command under rocprofv3 --kernel-trace --stats"; S=$(date +%s)
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --no-configs --no-pmc > "$OUT/bench_default_under_rocprof.json" 2> "$OUT/stats.err" ); echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT" <<'PY'
import csv, glob, json, os, shutil, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f))}
    ex = max((c for n, (c, a) in rows.items() if "gn_match_exact_cached_kernel" in n), default=0)
    relaxed = any("gn_match_cached_kernel" in n and ", true>" in n for n in rows)
    name = "main" if relaxed else ("pipelined_child" if ex > 500 else ("pyramid_child" if any("gn_match" in n and a > 90000 for n, (c, a) in rows.items()) else "8d_child"))
    shutil.copy(f, os.path.join(out, f"bench_default_{name}_kernel_stats.csv"))
    print(name, {n[:58]: v for n, v in rows.items() if "gn_match" in n})
d = json.loads(open(os.path.join(out, "bench_default_under_rocprof.json")).read().strip().splitlines()[-1])
print("line under rocprof: ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"], d["timed_regions"]["kernel_ms"])
PY

#!/bin/bash
# round 4, call F: the record of the final build -- suite, default bench line with pmc dumps, the same command under
# rocprofv3 --kernel-trace --stats, the extra workloads on their own, ThreadSanitizer build of the drop-in driver
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04f
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -4 "$OUT/pytest.log"
echo "== default bench line"; S=$(date +%s)
timeout 900 python bench.py --pmc-dump "$OUT/pmc" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], d["timed_regions"]["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("pmc_errors"))
print("fast", round(d["fast_mode"]["value"] / 1e6, 1), d["fast_mode"]["kernel_ms"], "relaxed", round(d["relaxed"]["value"] / 1e6, 1))
print("8d", round(d["headline_8d_starts"]["default"]["value"] / 1e6, 1), round(d["headline_8d_starts"]["fast"]["value"] / 1e6, 1))
print("pyramid", round(d["pyramid"]["exact"]["value"] / 1e6, 1), d["pyramid"]["exact"]["kernel_ms"], round(d["pyramid"]["fast"]["value"] / 1e6, 1), d["pyramid"]["fast"]["kernel_ms"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cold_after_update"], d["cpu_baseline_all_cores"]["value"])
for k, v in (d.get("configs") or {}).items():
    v = v or {}
    print(k, "value", v.get("value"), "ms", v.get("ms_per_step"), "match/update", v.get("match_ms"), v.get("update_ms"), "err", v.get("error"))
PY
echo "== same command under rocprofv3 --kernel-trace --stats"; S=$(date +%s)
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --no-configs --no-pmc > "$OUT/bench_default_under_rocprof.json" 2> "$OUT/stats.err" ); echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT" <<'PY'
import csv, glob, json, os, shutil, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f))}
    ex = max((c for n, (c, a) in rows.items() if "gn_match_exact_cached_kernel" in n), default=0)
    exa = max((a for n, (c, a) in rows.items() if "gn_match_exact_cached_kernel" in n), default=0)
    relaxed = any("gn_match_cached_kernel" in n and ", true>" in n for n in rows)
    fast = any("gn_match_cached_kernel" in n and ", false>" in n for n in rows)
    name = "main" if relaxed else ("pipelined_child" if not fast else ("pyramid_child" if exa > 90000 else "8d_child"))
    shutil.copy(f, os.path.join(out, f"bench_default_{name}_kernel_stats.csv"))
    print(name, {n[:58]: v for n, v in rows.items() if "gn_match" in n})
d = json.loads(open(os.path.join(out, "bench_default_under_rocprof.json")).read().strip().splitlines()[-1])
print("line under rocprof: ms_per_step", d["ms_per_step"], "kernel_ms", d["roofline"]["kernel_ms"])
PY
echo "== extra workloads"; S=$(date +%s)
for w in config3pyr config4 config5 config2; do
  timeout 600 python bench.py --workload $w --pmc-dump "$OUT/pmc_full" > "$OUT/bench_extra_$w.json" 2> "$OUT/bench_extra_$w.err"
  python - "$OUT/bench_extra_$w.json" $w <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], round(d["value"] / 1e6, 3), "M it/s", d.get("ms_per_step"), (d.get("fast_mode") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
done
echo "($(( $(date +%s) - S )) s)"
echo "== tsan"; S=$(date +%s)
timeout 600 bash tools/tsan_facade.sh > "$OUT/sanitizer_tsan_facade.txt" 2>&1; echo "rc=$? ($(( $(date +%s) - S )) s)"; tail -5 "$OUT/sanitizer_tsan_facade.txt"

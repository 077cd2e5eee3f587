#!/bin/bash
# full gpu suite + the default bench line + rocprofv3 kernel stats of the same command + the extra workloads on their own
# (each with its own counter passes); the summaries to keep are copied to profiles/rNN afterwards
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/full
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -14 "$OUT/pytest.log"
S=$(date +%s); timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench default: $(( $(date +%s) - S )) s"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --no-configs > "$OUT/bench_default_under_rocprof.json" 2> "$OUT/stats.err"
for f in $(find "$OUT/stats" -name "*kernel_stats.csv"); do echo "== $f"; head -6 $f | cut -c1-160; done
# the three processes of that command (main, pyramid child, pipelined child) by what they launched -> fixed names next to the
# json files (the per-PID names differ from run to run, and gpurun MERGES into a local gpurun_out/ that may hold older runs)
python - "$OUT" <<'PY'
import csv, glob, os, shutil, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(f))}
    fast = max((c for n, c in rows.items() if "gn_match_cached_kernel" in n and "false" in n), default=0)
    relaxed = any("gn_match_cached_kernel" in n and "true" in n for n in rows)
    name = "main" if relaxed else ("pipelined_child" if fast > 300 else "pyramid_child")
    shutil.copy(f, os.path.join(out, f"bench_default_{name}_kernel_stats.csv"))
PY
cd "$ROOT"
for w in config3pyr config4 config2 config5; do
  timeout 600 python bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"
  cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$w" -- python $ROOT/bench.py --workload $w --leg pmc --no-cpu --no-pmc > /dev/null 2>&1; cd "$ROOT"
  cp "$(ls -t $(find "$OUT/stats_$w" -name "*kernel_stats.csv") | head -1)" "$OUT/bench_${w}_kernel_stats.csv" 2>/dev/null
done
python tools/node_cycle_bench.py 400 > "$OUT/node_cycle_cpp.json" 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(d["value"]/1e6,3), "M it/s", d.get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY

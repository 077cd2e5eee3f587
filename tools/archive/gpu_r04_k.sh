#!/bin/bash
# round 4, call K: the record of this session's final build -- the whole GPU suite, smoke(), the default bench line with the
# pmc dumps (python bench.py), the driver's command, single-scan timings per parity mode
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${CALL:-r04k}
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
grep -E "passed|failed|pytest rc" "$OUT/pytest.log" | tail -3
S=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$? ($(( $(date +%s) - S )) s)"
S=$(date +%s)
timeout 900 python bench.py --pmc-dump "$OUT/pmc" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_like.json" 2> "$OUT/bench_driver_like.err"; echo "driver-like bench rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT" <<'PY'
import json, sys, os
for f in ("bench_default.json", "bench_driver_like.json"):
    d = json.loads(open(os.path.join(sys.argv[1], f)).read().strip().splitlines()[-1])
    print(f, "value", round(d["value"] / 1e6, 1), "M it/s", round(d["ms_per_step"] * 1e3, 2), "us", [round(x * 1e3, 2) for x in d["timed_regions"]["ms_per_step"]], "frac", round(d["roofline"]["frac"], 3), d["roofline"].get("pmc_errors"))
    if "fast_mode" in d: print("  fast", round(d["fast_mode"]["value"] / 1e6, 1), "relaxed", round(d["relaxed"]["value"] / 1e6, 1), "8d", round(d["headline_8d_starts"]["default"]["value"] / 1e6, 1), "pyramid", round(d["pyramid"]["exact"]["value"] / 1e6, 1), round(d["pyramid"]["fast"]["value"] / 1e6, 1))
    for k, v in (d.get("configs") or {}).items():
        v = v or {}
        print("  ", k, "value", v.get("value"), "ms", v.get("ms_per_step"), "match/update", v.get("match_ms"), v.get("update_ms"), "err", v.get("error"))
PY
timeout 300 python tools/parity_scene_sweep.py --batch 1024 --sample 256 --families baseline_room,hall_100x80_dotted_walls --out "$OUT/sweep_single.jsonl" 2>&1 | grep "single scans"

#!/bin/bash
# round 4, call G: what the driver runs at round end, on the tree as committed -- the whole GPU suite, smoke(), and
# `python bench.py --gpus 1 --steps 20 --warmup 5`
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04g
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -12 "$OUT/pytest.log"
S=$(date +%s)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$? ($(( $(date +%s) - S )) s)"; tail -2 "$OUT/smoke.log"
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_like.json" 2> "$OUT/bench_driver_like.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/bench_driver_like.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], d["timed_regions"]["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"].get("pmc_errors"))
for k, v in (d.get("configs") or {}).items():
    v = v or {}
    print(k, "value", v.get("value"), "ms", v.get("ms_per_step"), "match/update", v.get("match_ms"), v.get("update_ms"), "err", v.get("error"))
PY

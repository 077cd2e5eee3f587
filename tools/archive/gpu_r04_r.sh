#!/bin/bash
# round 4, call R: 16 x 8 mark tiles adopted -- the whole GPU suite, the dense soak, configs[4] with its counter passes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04r
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
grep -E "passed|failed|pytest rc" "$OUT/pytest.log" | tail -3
S=$(date +%s)
timeout 900 python tests/tools/soak_dense.py 5000 --check 500 > "$OUT/soak_dense_8192beams_1024map_5000_tile16.json" 2> /dev/null; echo "dense soak rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-330 "$OUT/soak_dense_8192beams_1024map_5000_tile16.json"
timeout 900 python tests/tools/soak_dense.py 1500 --check 500 --beams 16384 --size 2048 > "$OUT/soak_dense_16384beams_2048map_1500_tile16.json" 2> /dev/null; echo "dense soak 16k rc=$?"
S=$(date +%s)
timeout 900 python bench.py --workload config5 --pmc-dump "$OUT/pmc" > "$OUT/bench_extra_config5.json" 2> "$OUT/bench_config5.err"; echo "config5 bench rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/bench_extra_config5.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
u = d.get("update_roofline") or d.get("roofline", {}).get("update") or {}
print("step ms", d.get("ms_per_step"), "match", d.get("match_ms"), "update", d.get("update_ms"))
for k in ("traffic", "traffic_over_algorithmic", "kernel_time_us", "frac", "algorithmic_bytes_per_update"):
    print(" ", k, u.get(k))
print({k: (v or {}).get("avg_us") for k, v in (u.get("kernels") or {}).items()})
PY

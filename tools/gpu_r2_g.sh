#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/g
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_facade_dropin.py tests/test_node_rows.py tests/test_gpu_exact_parity.py -m gpu -q -x 2>&1 | tail -4
for lay in quad plane; do
    HSM_LAYOUT=$lay timeout 300 python bench.py --workload config2 --steps 300 --no-cpu > "$OUT/config2_$lay.json" 2>/dev/null
    python - <<PY
import json
d=json.loads(open("$OUT/config2_$lay.json").read().strip().splitlines()[-1])
print("$lay", "match", round(d["latency_us"]["median"],1), "upd call", round(d["update_latency_us"]["median"],1), "upd complete", round(d["update_complete_us"]["median"],1), "cycle", round(d["slam_cycle_us"]["median"],1))
PY
done
timeout 300 python tools/node_cycle_bench.py 400 | tee "$OUT/node_cycle.json"
timeout 300 python bench.py --workload config5 --no-cpu | cut -c1-900

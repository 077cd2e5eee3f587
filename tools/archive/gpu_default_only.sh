#!/bin/bash
# the default bench line + the rocprofv3 kernel stats of the same command, nothing else (tools/gpu_full.sh does the rest)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/default_only
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --no-configs > "$OUT/bench_default_under_rocprof.json" 2> "$OUT/stats.err"
cd "$ROOT"
python - <<PY
import json
for f in ("bench_default.json", "bench_default_under_rocprof.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], d["roofline"].get("clock_measured", {}).get("sclk_hz"))
PY

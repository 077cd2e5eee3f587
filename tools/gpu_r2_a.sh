#!/bin/bash
# round-2 GPU call A: full gpu test suite, headline bench in both parity modes, kernel trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/a
mkdir -p "$OUT"
cd "$ROOT"
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=15 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -30 "$OUT/pytest.log"
timeout 300 python bench.py --no-cpu > "$OUT/bench_fast.json" 2> "$OUT/bench_fast.err"
HSM_PARITY=exact timeout 300 python bench.py --no-cpu --steps 50 > "$OUT/bench_exact.json" 2> "$OUT/bench_exact.err"
HSM_TEXEL_CACHE=0 timeout 300 python bench.py --no-cpu > "$OUT/bench_nocache.json" 2> "$OUT/bench_nocache.err"
cat "$OUT"/bench_*.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --no-cpu --steps 50 > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs head -8

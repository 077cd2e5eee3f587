#!/bin/bash
# round 5, call E: suite (60 hypothesis examples) + smoke on the current build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05e
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=60 HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1800 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -25 "$OUT/pytest.log"
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== PCIe-inclusive host-buffer batch"; timeout 300 python tools/bench_host_batch.py 2>&1 | tail -1

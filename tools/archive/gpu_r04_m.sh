#!/bin/bash
# round 4, call M: the free-running exact-mode soaks again on the session's final build (exact_round changed: the exact
# single-scan matcher drives both), and the exact single-scan parity test with the new beam counts
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04m
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_exact_parity.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -2
S=$(date +%s)
timeout 900 python tests/tools/soak.py 30000 --exact --free-run > "$OUT/soak_exact_free_run_30000_call_m.json" 2> /dev/null; echo "node soak rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-500 "$OUT/soak_exact_free_run_30000_call_m.json"
S=$(date +%s)
timeout 900 python tests/tools/soak_dense.py 5000 --check 500 > "$OUT/soak_dense_8192beams_1024map_5000_call_m.json" 2> /dev/null; echo "dense soak rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/soak_dense_8192beams_1024map_5000_call_m.json"

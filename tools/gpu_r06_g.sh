#!/bin/bash
# round 6, call G: full GPU suite + the driver's bench command (with counters) on the current build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${HSM_ROUND_TAG:-r06g}
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=${HSM_HYPOTHESIS_EXAMPLES:-30} HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 2400 python -m pytest tests -m gpu -q --durations=8 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -16 "$OUT/pytest.log"
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
echo "== driver bench command"; S=$(date +%s)
HSM_BENCH_DETAILS=$OUT/bench_driver_like_details.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-dump "$OUT/pmc" > "$OUT/bench_driver_like.out" 2> "$OUT/bench_driver_like.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"
tail -1 "$OUT/bench_driver_like.out"

#!/bin/bash
# round 6, final tree: the longest parity runs -- hypothesis 50 000 examples per strategy (batches 50 000), node loop 1 000 000 scans,
# free-running soak 1 000 000 steps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06soak3
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 1500 python tools/node_loop_parity.py 1000000 --parity auto > "$OUT/node_loop_parity_default_1000000.json" 2> "$OUT/node_loop.err"; echo "node loop rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/node_loop_parity_default_1000000.json"
S=$(date +%s)
timeout 1800 python tests/tools/soak.py 1000000 --default --free-run > "$OUT/soak_default_free_run_1000000.json" 2> /dev/null; echo "soak rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/soak_default_free_run_1000000.json"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=50000 timeout 1200 python -m pytest tests/test_property_hypothesis.py -m gpu -q -k "batches" > "$OUT/hypothesis_batches_50000_examples.log" 2>&1; echo "hypothesis batches rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/hypothesis_batches_50000_examples.log"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=50000 timeout 3000 python -m pytest tests/test_property_hypothesis.py -m gpu -q -k "not batches" > "$OUT/hypothesis_50000_examples.log" 2>&1; echo "hypothesis rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/hypothesis_50000_examples.log"

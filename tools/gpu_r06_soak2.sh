#!/bin/bash
# round 6, final tree: the long parity runs once more with larger counts -- hypothesis 10 000 examples per strategy (batches 20 000),
# node loop 200 000 scans, free-running soak 200 000 steps, scene sweep with 1024 reference samples per row
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06soak2
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=10000 timeout 3000 python -m pytest tests/test_property_hypothesis.py -m gpu -q -k "not batches" > "$OUT/hypothesis_10000_examples.log" 2>&1; echo "hypothesis rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/hypothesis_10000_examples.log"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=20000 timeout 3000 python -m pytest tests/test_property_hypothesis.py -m gpu -q -k "batches" > "$OUT/hypothesis_batches_20000_examples.log" 2>&1; echo "hypothesis batches rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/hypothesis_batches_20000_examples.log"
S=$(date +%s)
timeout 1800 python tools/node_loop_parity.py 200000 --parity auto > "$OUT/node_loop_parity_default_200000.json" 2> "$OUT/node_loop.err"; echo "node loop rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/node_loop_parity_default_200000.json"
S=$(date +%s)
timeout 1800 python tests/tools/soak.py 200000 --default --free-run > "$OUT/soak_default_free_run_200000.json" 2> /dev/null; echo "soak rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/soak_default_free_run_200000.json"
S=$(date +%s)
timeout 2400 python tools/parity_scene_sweep.py --batch 4096 --sample 1024 --out "$OUT/parity_scene_sweep_batch_default_1024_samples.jsonl" > "$OUT/parity_scene_sweep.log" 2>&1; echo "scene sweep rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/parity_scene_sweep_batch_default_1024_samples.jsonl" <<'PY'
import json, sys
ok = n = 0
for ln in open(sys.argv[1]):
    d = json.loads(ln); n += 1
    ok += d["exact_vs_reference_sample"]["bit_identical"] == 1.0 and d["default_vs_exact"]["bit_identical"] == 1.0
print("scene sweep rows bit-identical:", ok, "of", n)
PY

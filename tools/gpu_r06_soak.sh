#!/bin/bash
# round 6: long parity runs on the final build -- hypothesis (2000 examples per strategy, batches 5000), scene sweep through the batch kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06soak
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=2000 timeout 3000 python -m pytest tests/test_property_hypothesis.py -m gpu -q -k "not batches" > "$OUT/hypothesis_2000_examples.log" 2>&1; echo "hypothesis rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/hypothesis_2000_examples.log"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=5000 timeout 3000 python -m pytest tests/test_property_hypothesis.py -m gpu -q -k "batches" > "$OUT/hypothesis_batches_5000_examples.log" 2>&1; echo "hypothesis batches rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/hypothesis_batches_5000_examples.log"
S=$(date +%s)
timeout 2400 python tools/parity_scene_sweep.py --batch 4096 --sample 256 --out "$OUT/parity_scene_sweep_batch_default.jsonl" > "$OUT/parity_scene_sweep.log" 2>&1; echo "scene sweep rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/parity_scene_sweep_batch_default.jsonl" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    print({k: d[k] for k in d if k in ("family", "run", "exact_vs_reference_bit_identical", "exact_bit_identical_to_reference_frac", "reference_sample", "exact_kernel", "fast_within_1e-4")} or list(d)[:12])
PY

#!/bin/bash
# round 4, call J: the parity scene sweep with EVERY scan of every run compared with the reference (--sample 4096; call A
# compared a 256-scan sample), and the node loop in the default mode over 30 000 steps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04j
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 1500 python tools/parity_scene_sweep.py --batch 4096 --sample 4096 --out "$OUT/parity_scene_sweep_all_scans.jsonl" > "$OUT/sweep.log" 2>&1; echo "sweep rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/parity_scene_sweep_all_scans.jsonl" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("scans", "within_1e-4", "bit_identical", "worst_dxy_m")}) for k, v in d.items() if k in ("family", "run", "levels", "fast_vs_exact", "exact_vs_reference", "auto_vs_reference", "reference_unsettled")})
PY
S=$(date +%s)
timeout 900 python tools/node_loop_parity.py 30000 --parity auto > "$OUT/node_loop_parity_auto_30000.json" 2> "$OUT/node.err"; echo "node loop rc=$? ($(( $(date +%s) - S )) s)"
cut -c1-600 "$OUT/node_loop_parity_auto_30000.json"; tail -2 "$OUT/node.err"

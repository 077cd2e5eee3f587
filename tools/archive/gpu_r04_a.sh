#!/bin/bash
# round 4, call A: full gpu suite (incl. the new dense-edge / border-hypothesis tests), dense soak, parity scene sweep,
# node-loop parity.  Everything lands in gpurun_out/r04a/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04a
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -22 "$OUT/pytest.log"
S=$(date +%s)
timeout 900 python tests/tools/soak_dense.py 5000 --check 250 > "$OUT/soak_dense_8192beams_1024map_5000.json" 2> "$OUT/soak1.err"; echo "soak1 rc=$? ($(( $(date +%s) - S )) s)"
cut -c1-400 "$OUT/soak_dense_8192beams_1024map_5000.json"; tail -3 "$OUT/soak1.err"
S=$(date +%s)
timeout 900 python tests/tools/soak_dense.py 1500 --check 250 --beams 16384 --size 2048 > "$OUT/soak_dense_16384beams_2048map_1500.json" 2> "$OUT/soak2.err"; echo "soak2 rc=$? ($(( $(date +%s) - S )) s)"
cut -c1-400 "$OUT/soak_dense_16384beams_2048map_1500.json"; tail -3 "$OUT/soak2.err"
S=$(date +%s)
timeout 900 python tools/parity_scene_sweep.py --out "$OUT/parity_scene_sweep.jsonl" > "$OUT/sweep.log" 2>&1; echo "sweep rc=$? ($(( $(date +%s) - S )) s)"
cat "$OUT/sweep.log" | cut -c1-260
S=$(date +%s)
for mode in auto exact; do
  timeout 600 python tools/node_loop_parity.py 5000 --parity $mode > "$OUT/node_loop_parity_$mode.json" 2> "$OUT/node_$mode.err"; echo "node loop $mode rc=$?"
  cat "$OUT/node_loop_parity_$mode.json"; tail -2 "$OUT/node_$mode.err"
done
timeout 600 python tools/node_loop_parity.py 5000 --parity auto --min-dist 0.0 --min-ang 0.0 > "$OUT/node_loop_parity_auto_zero_thresholds.json" 2> "$OUT/node_z.err"; cat "$OUT/node_loop_parity_auto_zero_thresholds.json"
echo "node loops ($(( $(date +%s) - S )) s)"

#!/bin/bash
# round 4, call E (final record): suite; the 30 %-out-of-map family of the scene sweep; soaks on the final build; the bench as
# the driver invokes it (--steps 20 --warmup 5) and in its default form, + the group leg with two replicas on one device
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04e
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=5 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -9 "$OUT/pytest.log"
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
echo "== sweep: 30 % of the end points outside the map"; S=$(date +%s)
timeout 600 python tools/parity_scene_sweep.py --batch 4096 --sample 256 --families 30pct_beams_out_of_map --out "$OUT/parity_scene_sweep_30pct.jsonl" > "$OUT/sweep30.log" 2>&1; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-240 "$OUT/sweep30.log"
echo "== soaks"; S=$(date +%s)
timeout 900 python tests/tools/soak_dense.py 5000 --check 500 > "$OUT/soak_dense_8192beams_1024map_5000_final.json" 2> /dev/null; echo "dense rc=$?"; cut -c1-260 "$OUT/soak_dense_8192beams_1024map_5000_final.json"
timeout 900 python tests/tools/soak.py 30000 --exact --free-run > "$OUT/soak_exact_free_run_30000.json" 2> /dev/null; echo "node rc=$?"; cat "$OUT/soak_exact_free_run_30000.json"
echo "($(( $(date +%s) - S )) s)"
echo "== bench as the driver runs it"; S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driver_like.json" 2> "$OUT/bench_driver_like.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/bench_driver_like.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], d["timed_regions"]["ms_per_step"], "frac", d["roofline"]["frac"], "errors", d["roofline"].get("pmc_errors"))
print("fast", round(d["fast_mode"]["value"] / 1e6, 1), "cpu", d["cpu_baseline"]["value"], [k for k in d])
PY
echo "== group of two replicas on one device (peer gather; RCCL refuses a device listed twice)"
timeout 300 python bench.py --group 2 --steps 30 > "$OUT/bench_group2_one_device.json" 2> "$OUT/group2.err"; echo "rc=$?"; cut -c1-1200 "$OUT/bench_group2_one_device.json" | tail -c 700

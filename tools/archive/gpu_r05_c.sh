#!/bin/bash
# round 5, call C: suite with 60 hypothesis examples per GPU strategy; two ranks sharing the one device (does the N > 1 path end
# in the compact line?); the node loop and the free-running soaks in the library DEFAULT mode (must be bit-identical now); the
# long bench run (--all-configs) for the record.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05c
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=60 HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1800 python -m pytest tests -m gpu -q --durations=6 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -12 "$OUT/pytest.log"
echo "== two ranks on one device (gloo; exercises the N > 1 code path only)"; S=$(date +%s)
HSM_BENCH_SHARE_GPU=1 HSM_BENCH_DETAILS=$OUT/bench_details_2ranks_share_gpu.json timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > "$OUT/bench_2ranks.out" 2> "$OUT/bench_2ranks.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/bench_2ranks.out" <<'PY'
import json, sys
last = open(sys.argv[1]).read().strip().splitlines()[-1]
d = json.loads(last)
print("2-rank LINE bytes", len(last), "n_gpus", d["n_gpus"], "value", d["value"], "details", d["details"]); print(last[:700])
PY
python - "$OUT/bench_details_2ranks_share_gpu.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("ranks", json.dumps(d.get("ranks"))[:600])
PY
echo "== node loop, default mode, 30000 scans"; S=$(date +%s)
timeout 900 python tools/node_loop_parity.py 30000 --parity auto > "$OUT/node_loop_parity_default_30000.json" 2> "$OUT/node_loop.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-700 "$OUT/node_loop_parity_default_30000.json"
echo "== free-running soak, default mode, 30000 steps"; S=$(date +%s)
timeout 900 python tests/tools/soak.py 30000 --default --free-run > "$OUT/soak_default_free_run_30000.json" 2> /dev/null; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-500 "$OUT/soak_default_free_run_30000.json"
echo "== dense soak (8192 beams, 1024^2), 3000 steps"; S=$(date +%s)
timeout 900 python tests/tools/soak_dense.py 3000 --check 500 > "$OUT/soak_dense_8192beams_1024map_3000.json" 2> /dev/null; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/soak_dense_8192beams_1024map_3000.json"
echo "== the long bench run"; S=$(date +%s)
HSM_BENCH_DETAILS=$OUT/bench_all_configs_details.json timeout 1500 python bench.py --all-configs --pmc-dump "$OUT/pmc" > "$OUT/bench_all_configs.out" 2> "$OUT/bench_all_configs.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"; tail -1 "$OUT/bench_all_configs.out" | cut -c1-300
python - "$OUT/bench_all_configs_details.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"])
print("fast", round(d["fast_mode"]["value"] / 1e6, 1), d["fast_mode"]["kernel_ms"])
for k in ("headline_8d_starts", "relaxed", "pyramid", "pipelined", "cpu_baseline_all_cores"):
    print(k, json.dumps(d.get(k))[:500])
for k, v in (d.get("configs") or {}).items():
    v = v or {}
    print(k, "value", v.get("value"), "ms", v.get("ms_per_step"), "match/update", v.get("match_ms"), v.get("update_ms"), "err", v.get("error"))
PY

#!/bin/bash
# Round-end validation of the current build on one MI355X box (gpurun): suite (60 hypothesis examples), smoke, the driver's bench command + the same under
# rocprofv3 --kernel-trace --stats, the long bench run, node loop + soaks in the default mode, ThreadSanitizer over the facade
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${HSM_ROUND_TAG:-round_end}
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_HYPOTHESIS_EXAMPLES=60 HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1800 python -m pytest tests -m gpu -q --durations=6 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -14 "$OUT/pytest.log"
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
echo "== driver bench command"; S=$(date +%s)
HSM_BENCH_DETAILS=$OUT/bench_driver_like_details.json timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-dump "$OUT/pmc_driver_like" > "$OUT/bench_driver_like.out" 2> "$OUT/bench_driver_like.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/bench_driver_like.out" <<'PY'
import json, sys
last = open(sys.argv[1]).read().strip().splitlines()[-1]
d = json.loads(last)
print("LINE bytes", len(last)); print(last)
PY
echo "== same command under rocprofv3 --kernel-trace --stats"; S=$(date +%s)
( cd /tmp && export TMPDIR=/tmp && HSM_BENCH_DETAILS=$OUT/bench_details_under_rocprof.json timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu > "$OUT/bench_under_rocprof.out" 2> "$OUT/stats.err" ); echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT" <<'PY'
import csv, glob, os, shutil, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f))}
    shutil.copy(f, os.path.join(out, "bench_driver_like_kernel_stats.csv"))
    print({n[:70]: v for n, v in rows.items() if "gn_match" in n})
PY
echo "== the long bench run"; S=$(date +%s)
HSM_BENCH_DETAILS=$OUT/bench_all_configs_details.json timeout 1500 python bench.py --all-configs --pmc-dump "$OUT/pmc" > "$OUT/bench_all_configs.out" 2> "$OUT/bench_all_configs.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT/bench_all_configs_details.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], "kernel", d["roofline"]["kernel_ms"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"])
print("fast", round(d["fast_mode"]["value"] / 1e6, 1), d["fast_mode"]["kernel_ms"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("bit_identical_pose_fraction"))
for k in ("gentle_starts", "relaxed", "pyramid", "pipelined", "cpu_baseline_all_cores", "sustained"):
    print(k, json.dumps(d.get(k))[:300])
for k, v in (d.get("configs") or {}).items():
    v = v or {}
    print(k, "value", v.get("value"), "ms", v.get("ms_per_step"), "match/update", v.get("match_ms"), v.get("update_ms"), "fast", (v.get("fast_mode") or {}).get("ms_per_step"), "err", v.get("error"))
PY
echo "== multi-rank path on this box: one rank with a process group (nccl), two ranks sharing the device (self-launched), the C++ group"; S=$(date +%s)
HSM_BENCH_FORCE_DIST=1 HSM_BENCH_DETAILS=$OUT/bench_one_rank_nccl_group_details.json timeout 600 python bench.py --gpus 1 --steps 200 --warmup 10 --no-pmc --no-cpu > "$OUT/bench_one_rank_nccl_group.out" 2> "$OUT/bench_one_rank_nccl_group.err"; echo "rc=$?"; tail -1 "$OUT/bench_one_rank_nccl_group.out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('direct gather per match', d['ms_per_step'], d.get('gather_legs'))"
HSM_BENCH_SHARE_GPU=1 HSM_BENCH_DETAILS=$OUT/bench_two_ranks_one_device_details.json timeout 600 python bench.py --gpus 2 --steps 100 --warmup 10 --no-pmc --no-cpu > "$OUT/bench_two_ranks_one_device.out" 2> "$OUT/bench_two_ranks_one_device.err"; echo "rc=$?"; tail -1 "$OUT/bench_two_ranks_one_device.out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 ranks', d['ms_per_step'], d.get('gather_legs'))"
timeout 600 python bench.py --group 2 --steps 50 > "$OUT/bench_group2.out" 2> "$OUT/bench_group2.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"
echo "== node loop, default mode, 30000 scans"; S=$(date +%s)
timeout 900 python tools/node_loop_parity.py 30000 --parity auto > "$OUT/node_loop_parity_default_30000.json" 2> "$OUT/node_loop.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/node_loop_parity_default_30000.json"
echo "== free-running soak, default mode, 30000 steps"; S=$(date +%s)
timeout 900 python tests/tools/soak.py 30000 --default --free-run > "$OUT/soak_default_free_run_30000.json" 2> /dev/null; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-400 "$OUT/soak_default_free_run_30000.json"
echo "== dense soak (16384 beams, 2048^2), 1500 steps"; S=$(date +%s)
timeout 900 python tests/tools/soak_dense.py 1500 --beams 16384 --size 2048 --check 500 > "$OUT/soak_dense_16384beams_2048map_1500.json" 2> /dev/null; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-300 "$OUT/soak_dense_16384beams_2048map_1500.json"
echo "== batch order: the headline batch and the 4096^2 pyramid in trajectory / random order, as given and through HSM_ORDER_MORTON"; S=$(date +%s)
timeout 600 python tools/study/batch_order_locality.py 2> /dev/null | grep -v amdgpu.ids > "$OUT/batch_order_locality.txt"; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-150 "$OUT/batch_order_locality.txt"
echo "== exchange cost by world size (one device)"; S=$(date +%s)
timeout 600 python tools/study/exchange_world_cost.py 2> /dev/null | grep -v amdgpu.ids > "$OUT/exchange_cost_by_world_size_one_device.txt"; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-200 "$OUT/exchange_cost_by_world_size_one_device.txt"
echo "== ThreadSanitizer over the facade"; bash tools/tsan_facade.sh 2>&1 | tail -4; cp gpurun_out/tsan/tsan_stdout.txt "$OUT/sanitizer_tsan_facade.txt" 2>/dev/null
echo "== default-mode batches by size (chain-wavefront form up to 3072 scans)"; S=$(date +%s)
for L in 1 3; do timeout 300 python tools/batch_size_sweep.py --levels $L --variants "default;HSM_EXACT_CHAIN_WAVE=0" > "$OUT/batch_size_sweep_l$L.jsonl" 2> /dev/null; done; echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT" <<'PY'
import json, sys
for L in (1, 3):
    for ln in open(f"{sys.argv[1]}/batch_size_sweep_l{L}.jsonl"):
        d = json.loads(ln)
        print(L, d["batch"], " | ".join("%s %.1f us%s" % (k[:24], v["us"], "" if v["bit_identical_to_first"] else " DIFF") for k, v in d.items() if isinstance(v, dict)))
PY

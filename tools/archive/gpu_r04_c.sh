#!/bin/bash
# round 4, call C: full suite; single-scan leg of the scene sweep; default bench line with pmc dumps + the same command under
# rocprofv3 --kernel-trace --stats; what-if for verdict item 7 (hardware sin/cos in the single-scan kernel); the config5 child
# under rocprofv3 (exit code); 2-rank share-GPU smoke of the multi-rank bench path.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04c
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
HSM_PARITY_STATS=$OUT/parity_stats.jsonl timeout 1500 python -m pytest tests -m gpu -q --durations=6 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -12 "$OUT/pytest.log"
echo "== single-scan sweep"; S=$(date +%s)
timeout 900 python tools/parity_scene_sweep.py --batch 1024 --sample 256 --out "$OUT/parity_scene_sweep_single_scans.jsonl" > "$OUT/sweep_single.log" 2>&1; echo "rc=$? ($(( $(date +%s) - S )) s)"
grep "single scans" -B1 "$OUT/sweep_single.log" | cut -c1-230
echo "== default bench line"; S=$(date +%s)
timeout 900 python bench.py --pmc-dump "$OUT/pmc" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], d["timed_regions"]["ms_per_step"])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "kernel_ms", "frac", "achieved", "traffic", "counter_source", "pmc_errors")})
print("fast", round(d["fast_mode"]["value"] / 1e6, 1), d["fast_mode"]["kernel_ms"], d["fast_mode"].get("fast_vs_default_all_scans"))
print("8d", json.dumps(d.get("headline_8d_starts"))[:1200])
for k, v in (d.get("configs") or {}).items():
    v = v or {}
    print(k, "value", v.get("value"), "ms", v.get("ms_per_step"), "err", v.get("error"), json.dumps(v.get("roofline", {}).get("pmc_errors")), json.dumps((v.get("update_roofline") or {}).get("pmc_errors")))
PY
ls "$OUT/pmc"
echo "== same command under rocprofv3 --kernel-trace --stats"; S=$(date +%s)
( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python $ROOT/bench.py --no-configs --no-pmc > "$OUT/bench_default_under_rocprof.json" 2> "$OUT/stats.err" ); echo "rc=$? ($(( $(date +%s) - S )) s)"
python - "$OUT" <<'PY'
import csv, glob, os, shutil, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"])) for r in csv.DictReader(open(f))}
    ex = max((c for n, (c, a) in rows.items() if "gn_match_exact_cached_kernel" in n), default=0)
    fast = max((c for n, (c, a) in rows.items() if "gn_match_cached_kernel" in n and "true" not in n.split("gn_match_cached_kernel")[1][:40]), default=0)
    relaxed = any("gn_match_cached_kernel" in n and "Lb1" in n or ("gn_match_cached_kernel" in n and ", true>" in n) for n in rows)
    name = "main" if ex >= 1000 else ("pipelined_child" if fast > 300 or ex > 300 else "pyramid_child")
    shutil.copy(f, os.path.join(out, f"bench_default_{name}_kernel_stats.csv"))
    print(name, {n[:60]: v for n, v in rows.items() if "gn_match" in n})
PY
echo "== item 7 what-if: hardware sin/cos in the single-scan kernel"; S=$(date +%s)
cd /tmp; export TMPDIR=/tmp
for v in base fsc; do
  lib=""; [ "$v" != base ] && lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  rm -rf /tmp/k2_$v
  HSM_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2_$v -- python $ROOT/bench.py --workload config2 --leg pmc --no-cpu --no-pmc > /dev/null 2>&1
  f=$(find /tmp/k2_$v -name "*kernel_stats.csv" | tail -1)
  python - "$v" "$f" <<'PY' | tee -a "$OUT/item7_sincos_whatif.txt"
import csv, sys
v, f = sys.argv[1], sys.argv[2]
rows = {r["Name"].split("(")[0].replace("void ", "").replace("hsm::", ""): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f))}
print(v, {k: x for k, x in rows.items() if "gn_match" in k or "update_" in k})
PY
done
echo "== config5 child under rocprofv3: exit code"
rm -rf /tmp/kt5; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $ROOT/bench.py --workload config5 --leg pmc --no-cpu --no-pmc > /dev/null 2> "$OUT/config5_child.err"; echo "rc=$? trace files: $(find /tmp/kt5 -name '*kernel_trace.csv' | wc -l)" | tee "$OUT/config5_child_exit.txt"
cd "$ROOT"
echo "($(( $(date +%s) - S )) s)"
echo "== 2 ranks sharing the GPU (gloo): multi-rank bench path"; S=$(date +%s)
HSM_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu > "$OUT/two_rank_share_gpu_config3.json" 2> "$OUT/two_rank.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"
tail -1 "$OUT/two_rank_share_gpu_config3.json" | cut -c1-600

#!/bin/bash
# round 4, call B: suite on the pruned library; RCCL group leg; single-scan leg of the scene sweep; SQ_ACTIVE_INST_VALU
# calibration on a pure VALU stream; why rocprofv3 children of the config5 leg die at exit; update-kernel variants
# (beams per wavefront, duplicate suppression): correctness + kernel times; the new default bench line.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04b
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> "$OUT/pytest.log"
tail -14 "$OUT/pytest.log"
echo "== bench --group 1"; S=$(date +%s)
timeout 300 python bench.py --group 1 --steps 40 > "$OUT/bench_group1.json" 2> "$OUT/group1.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"; cut -c1-900 "$OUT/bench_group1.json"; tail -3 "$OUT/group1.err"
echo "== single-scan sweep"; S=$(date +%s)
timeout 900 python tools/parity_scene_sweep.py --batch 1024 --sample 512 --out "$OUT/parity_scene_sweep_single_scans.jsonl" > "$OUT/sweep_single.log" 2>&1; echo "rc=$? ($(( $(date +%s) - S )) s)"
grep "single scans" -B1 "$OUT/sweep_single.log" | cut -c1-230
echo "== ubench_valu under the counters"; S=$(date +%s)
hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_valu tools/ubench_valu.hip 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$OUT/ubench_pmc" -- /tmp/ubench_valu > "$OUT/ubench_valu_under_pmc.txt" 2>&1 )
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/ubench_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:40], r.get("Workgroup_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/ubench_valu_counter_calibration.txt", "w") as fh:
    for k, v in sorted(acc.items()):
        m = {c: sum(x) / len(x) for c, x in v.items()}
        line = f"{k}: " + " ".join(f"{c}={m[c]:.4g}" for c in sorted(m))
        if m.get("SQ_INSTS_VALU"):
            line += f"  -> SQ_ACTIVE_INST_VALU*4/SQ_INSTS_VALU = {m.get('SQ_ACTIVE_INST_VALU', 0) * 4 / m['SQ_INSTS_VALU']:.3f} 'cycles' per instruction"
        print(line); fh.write(line + "\n")
PY
grep -i "cycles\|W=" "$OUT/ubench_valu_under_pmc.txt" | head -12
echo "($(( $(date +%s) - S )) s)"
echo "== rocprofv3 + config5 child exit"; S=$(date +%s)
cd /tmp; export TMPDIR=/tmp
for v in base nocoop osexit; do
  rm -rf /tmp/kt5_$v
  case $v in
    base) envs="";;
    nocoop) envs="HSM_COOP_MIN=100000000";;
    osexit) envs="HSM_BENCH_OS_EXIT=1";;
  esac
  env $envs timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5_$v -- python -X faulthandler $ROOT/bench.py --workload config5 --leg pmc --no-cpu --no-pmc > "$OUT/segv_$v.out" 2> "$OUT/segv_$v.err"
  echo "variant $v: rc=$? trace files: $(find /tmp/kt5_$v -name '*kernel_trace.csv' | wc -l)"; grep -n "Fatal\|Segmentation\|File \"" "$OUT/segv_$v.err" | head -12
done
cd "$ROOT"
echo "($(( $(date +%s) - S )) s)"
echo "== update-kernel variants"; S=$(date +%s)
for v in base g2 g4 d0 g2d0 g4d0; do
  lib=""; [ "$v" != base ] && lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  [ "$v" != base ] && [ ! -f "$lib" ] && { echo "$v: not built"; continue; }
  HSM_LIB=$lib timeout 600 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_full_size.py -k "dense or config5" -m gpu -q -x 2>&1 | tail -2 | sed "s/^/$v tests: /"
done
bash tools/exp_update_kernels.sh base g2 g4 d0 g2d0 g4d0 2>&1 | tee "$OUT/update_variants_kernel_us.txt"
echo "($(( $(date +%s) - S )) s)"
echo "== default bench line"; S=$(date +%s)
timeout 900 python bench.py --pmc-dump "$OUT/pmc" > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"; tail -3 "$OUT/bench_default.err"
python - "$OUT/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"] / 1e6, 1), "M it/s", d["ms_per_step"], d["config"]["parity_mode"], d["timed_regions"])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "kernel_ms", "frac", "achieved", "traffic", "counter_source", "pmc_errors")})
for k in ("fast_mode", "headline_8d_starts", "relaxed"):
    print(k, json.dumps(d.get(k))[:700])
print("cpu", json.dumps(d.get("cpu_baseline"))[:400])
print("pyramid", json.dumps(d.get("pyramid"))[:600])
for k, v in (d.get("configs") or {}).items():
    print(k, json.dumps(v)[:500])
PY
ls "$OUT/pmc" 2>/dev/null

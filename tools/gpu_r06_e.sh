#!/bin/bash
# round 6, call E: the early-production schedule of the exact batch kernel (HSM_XEARLY): parity, then A/B timing against -DHSM_XEARLY=0
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06e
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_exact_parity.py tests/test_gpu_full_size.py -m gpu -x -q --durations=5 -k "batch or config3 or config4 or generation or chain_wavefront or randomised or golden" > "$OUT/pytest_batch.log" 2>&1; echo "batch parity rc=$? ($(( $(date +%s) - S )) s)"; tail -8 "$OUT/pytest_batch.log"
for v in new early0 early1 new early0; do
  if [ $v = new ]; then unset HSM_LIB; else export HSM_LIB=$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so; fi
  HSM_BENCH_DETAILS=$OUT/bench_$v.json timeout 600 python bench.py --gpus 1 --steps 200 --warmup 10 --no-pmc --no-cpu --sustain-s 0 > "$OUT/bench_$v.out" 2> "$OUT/bench_$v.err"
  echo "$v rc=$?"; tail -1 "$OUT/bench_$v.out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1),'M it/s', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'fast', round(d.get('fast_mode_value',0)/1e6,1))"
done
unset HSM_LIB

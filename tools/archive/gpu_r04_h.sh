#!/bin/bash
# round 4, call H: decoupled rounds of the exact-order batch kernel (-DHSM_XDECOUPLE=1) against the barrier form
#   VARIANTS="a b" whole exact-parity suite per variant; SWEEP="base a b" headline timing; DEBUG="x y" wait-cycle dump
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p "$OUT"
cd "$ROOT"
for v in ${VARIANTS:-}; do
  lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  S=$(date +%s)
  HSM_LIB=$lib timeout 600 python -m pytest tests/test_gpu_exact_parity.py -m gpu -q -x > "$OUT/pytest_exact_$v.log" 2>&1
  echo "$v exact parity suite: $(tail -1 $OUT/pytest_exact_$v.log) ($(( $(date +%s) - S )) s)"
done
[ -n "${SWEEP:-}" ] && bash tools/exp_exact_params.sh $SWEEP 2>&1 | tee -a "$OUT/sweep.txt"
for v in ${DEBUG:-}; do
  echo "$v $(HSM_LIB=$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so timeout 300 python tools/exp_decouple_debug.py $OUT/timeline_$v.json 2>/dev/null | tail -1)" | tee -a "$OUT/debug_waits.txt"
done

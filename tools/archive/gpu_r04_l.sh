#!/bin/bash
# round 4, call L: the dirty-group flags of the dense update (-DHSM_DIRTY_SB=1) against the shipped form: parity tests of the
# dense path under each library, configs[4] step / match / update, kernel durations (rocprofv3 kernel trace)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04l
mkdir -p "$OUT"
cd "$ROOT"
for v in ${VARIANTS:-base2 dsb}; do
  lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  S=$(date +%s)
  HSM_LIB=$lib timeout 900 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_full_size.py tests/test_property_hypothesis.py -m gpu -q -x -k "dense or config5 or border" > "$OUT/pytest_$v.log" 2>&1
  echo "$v dense tests: $(grep -E 'passed|failed|error' $OUT/pytest_$v.log | tail -1) ($(( $(date +%s) - S )) s)" | tee -a "$OUT/summary.txt"
done
bash tools/exp_update_ab.sh ${VARIANTS:-base2 dsb} 2>&1 | tee -a "$OUT/summary.txt"
bash tools/exp_update_kernels.sh ${VARIANTS:-base2 dsb} 2>&1 | tee -a "$OUT/summary.txt"

#!/bin/bash
# round 4, call I: how many texel rows the exact-order batch kernel should cache (-DHSM_XBPC=n): headline batch at a settled
# clock, then the 3-level batch and the 4096^2 pyramid (200 back-to-back launches each)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04i
mkdir -p "$OUT"
cd "$ROOT"
bash tools/exp_exact_params.sh $SWEEP 2>&1 | tee -a "$OUT/sweep.txt"
for v in $SWEEP; do
  lib=""; [ "$v" != base ] && lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  HSM_LIB=$lib timeout 600 python tools/exp_exact_cached.py ${WORKLOADS:-config3pyr config4} --variants "HSM_X=1" --steps 200 2>/dev/null | python -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln)
    if d['variant']!='fast': print('$v', d['workload'], round(d['kernel_us'],1), 'us', round(d['M_it_per_s'],1), 'M it/s')" | tee -a "$OUT/sweep.txt"
done

#!/bin/bash
# headline batch in the default (exact-order) mode for library variants: us per launch (median of 5 regions of 200, pre-warmed)
#   tools/exp_exact_params.sh base xb14 ...      (variants from tools/build_variants.py)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
export HSM_BENCH_INPUT_CACHE=/tmp/hsm_inputs_exact_params; mkdir -p $HSM_BENCH_INPUT_CACHE
for v in "$@"; do
  lib=""; [ "$v" != base ] && lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  [ "$v" != base ] && [ ! -f "$lib" ] && { echo "$v: not built"; continue; }
  t=$(HSM_LIB=$lib timeout 300 python -m pytest tests/test_gpu_exact_parity.py -k "batch_ragged and (cached or auto)" -m gpu -q 2>&1 | tail -1)
  HSM_LIB=$lib timeout 300 python bench.py --no-cpu --no-pmc --no-configs --no-pyramid --no-pipelined --no-relaxed --no-exact 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['timed_regions']
print('$v', 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'regions', [round(x*1e3,2) for x in r['kernel_ms']], 'M it/s', round(d['value']/1e6,1), '| tests: $t')"
done

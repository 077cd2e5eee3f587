#!/bin/bash
# config4 (4096^2 pyramid), fast + exact legs, library variants: tools/exp_config4_exact.sh "base sw1"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for rep in 1 2; do for v in $1; do
 if [ $v != base ]; then export HSM_LIB=$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so; else unset HSM_LIB; fi
 python bench.py --workload config4 --no-cpu --no-pmc --no-pyramid --no-pipelined --steps 50 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['exact_parity']
print('config4 $v fast kernel_us', round(d['roofline']['kernel_ms']*1e3,1), 'exact kernel_us', round(e['kernel_ms']*1e3,1))"
done; done

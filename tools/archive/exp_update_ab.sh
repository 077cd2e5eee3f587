#!/bin/bash
# A/B of compile-time variants of the dense update kernels on configs[4] (tools/build_variants.py name:"-D..." first):
#   tools/exp_update_ab.sh base xc0 ...   -> one line per variant: ms per step, match, update
for v in "$@"; do
  lib=""
  [ "$v" != base ] && lib="hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  HSM_LIB=$lib python bench.py --workload config5 --no-cpu --no-pmc 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'], 4), round(d.get('match_ms', 0), 4), round(d.get('update_ms', 0), 4), d.get('self_check', d.get('parity', '')) if False else '')"
done

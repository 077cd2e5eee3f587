#!/bin/bash
# A/B of library variants on the headline batch, fast mode (tools/build_variants.py first): kernel us per launch
#   tools/exp_fast_ab.sh base epnt ...
for v in "$@"; do
  lib=""
  [ "$v" != base ] && lib="hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  HSM_LIB=$lib python tools/exp_exact_cached.py ${WORKLOADS:-config3 config3pyr} --variants "HSM_EXACT_CACHED=1" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('$v', d['workload'], 'fast' if d['variant'] == 'fast' else 'exact', round(d['kernel_us'], 1))"
done

#!/bin/bash
# A/B of compile-time variants of the exact batch kernel (tools/build_variants.py name:"-D..." first):
#   tools/exp_exact_ab.sh base pp1 ph8 ...     -> gpurun_out/xab_<name>.jsonl, one summary line per variant and workload
# "base" is the in-tree library.  Workloads: $WORKLOADS (default "config3 config4").
mkdir -p gpurun_out
for v in "$@"; do
  lib=""
  [ "$v" != base ] && lib="hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  HSM_LIB=$lib python tools/exp_exact_cached.py ${WORKLOADS:-config3 config4} --variants "HSM_EXACT_CACHED=0;HSM_EXACT_CACHED=1" \
    > gpurun_out/xab_$v.jsonl 2> gpurun_out/xab_$v.err
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
for l in open(f"gpurun_out/xab_{v}.jsonl"):
    d = json.loads(l)
    if d["variant"] == {"HSM_EXACT_CACHED": "1"}:
        print(v, d["workload"], round(d["kernel_us"], 1), d["pose_identical_to_first_variant"], d["cov_identical_to_first_variant"])
PY
done

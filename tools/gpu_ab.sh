#!/bin/bash
# A/B of library variants (hector_slam_amd/lib/variants/libhector_mi355_<name>.so; "base" = the in-tree build) on the
# batched workloads: kernel time per launch and a pose checksum (bit-identity across variants)
# usage: tools/gpu_ab.sh "base v1 v2" "config3 config3pyr config4"
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab
mkdir -p "$OUT"
cd "$ROOT"
for wl in ${2:-config3}; do
 for rep in 1 2; do
  for v in $1; do
    if [ $v != base ]; then export HSM_LIB=$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so; else unset HSM_LIB; fi
    python bench.py --workload $wl --no-cpu --no-exact --no-pmc --no-pyramid --steps ${STEPS:-100} > $OUT/${wl}_$v.json 2>$OUT/${wl}_$v.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/${wl}_$v.json").read().strip().splitlines()[-1])
    print("$wl", "$v", "rep$rep", "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "Mit/s", round(d["value"]/1e6,1), "conv", d.get("convergence",{}).get("median_abs_err_xy_m"))
except Exception as e:
    print("$wl $v failed", e, open("$OUT/${wl}_$v.err").read()[-300:])
PY
  done
 done
done

#!/bin/bash
# per-kernel durations of the configs[4] step for library variants (rocprofv3 kernel trace; counter-free):
#   tools/exp_update_kernels.sh base mw1 ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=""
  [ "$v" != base ] && lib="$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so"
  rm -rf /tmp/kt_$v
  HSM_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python $ROOT/bench.py --workload config5 --leg pmc --no-cpu --no-pmc > /dev/null 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | tail -1)
  python - "$v" "$f" <<'PY'
import csv, sys
v, f = sys.argv[1], sys.argv[2]
rows = {r["Name"].split("(")[0].replace("void ", "").replace("hsm::", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
print(v, {k: round(x, 1) for k, x in rows.items() if "update_" in k or "coop" in k})
PY
done

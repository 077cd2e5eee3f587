#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/h
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for zc in 4096 0; do
HSM_UPDATE_ZEROCOPY_MAX=$zc HSM_LAYOUT=plane timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$zc" -- python $ROOT/bench.py --workload config2 --steps 100 --no-cpu > "$OUT/c2_$zc.json" 2>"$OUT/c2.err"
echo "zero-copy max $zc"; find "$OUT/trace$zc" -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-150
done

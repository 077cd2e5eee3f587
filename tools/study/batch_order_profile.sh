#!/bin/bash
# kernel durations of a permuted batch launch (sort kernel + matcher) under rocprofv3 --kernel-trace --stats
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/order_prof
rm -rf "$OUT"; mkdir -p "$OUT"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python "$ROOT/tools/study/batch_order_profile.py" > "$OUT/run.log" 2>&1 )
tail -3 "$OUT/run.log"
python - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY

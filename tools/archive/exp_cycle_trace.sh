#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp_trace
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
HSM_LAYOUT=plane timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python $ROOT/bench.py --workload config2 --steps 100 --no-cpu > "$OUT/c2.json" 2>"$OUT/c2.err"
f=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 40 dispatches: name, duration, gap to previous end
prev=None
for r in rows[-48:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(f'{r["Kernel_Name"][:50]:50s} dur {(e-s)/1e3:7.2f} us  gap {((s-prev)/1e3 if prev else 0):8.2f} us  grid {r.get("Grid_Size_X","?")}x{r.get("Grid_Size_Y","?")}')
    prev=e
PY

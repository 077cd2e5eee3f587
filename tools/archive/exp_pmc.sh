#!/bin/bash
# PMC passes around an arbitrary command on the GPU box; per-kernel means.   usage: tools/exp_pmc.sh <tag> "<groups ;-separated>" -- cmd...
set -u
TAG=$1; GROUPS_=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
IFS=';' read -ra GS <<< "$GROUPS_"
for g in "${GS[@]}"; do
  name=$(echo "$g" | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $g --output-format csv -d "$OUT/pmc_$name" -- "$@" > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $name failed" >> "$OUT/errors.log"
done
cd "$ROOT"
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0][-70:]
            if "gn_match" in k or "update_" in k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(os.path.join(out, "pmc_summary.txt"), "w") as fo:
    for k, cs in sorted(acc.items()):
        line = k + " " + str({c: (round(sum(v[2:]) / max(len(v[2:]), 1), 1), len(v)) for c, v in sorted(cs.items())})
        print(line)
        fo.write(line + "\n")
PY

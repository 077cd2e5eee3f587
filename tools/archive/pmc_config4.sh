#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  name=$(echo "$pmc" | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $ROOT/gpurun_out/c4_$name -- python $ROOT/bench.py --workload config4 --no-cpu --steps 10 --warmup 2 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$ROOT/gpurun_out/c4_$name/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gn_match" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k: sum(v)/len(v) for k, v in acc.items()})
PY
done

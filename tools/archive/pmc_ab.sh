#!/bin/bash
ROOT=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for v in base ${VARIANTS:-tile}; do
  if [ $v != base ]; then export HSM_LIB=$ROOT/hector_slam_amd/lib/variants/libhector_mi355_$v.so; else unset HSM_LIB; fi
  rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $ROOT/gpurun_out/ab_$v -- python $ROOT/bench.py --no-cpu --no-pyramid --steps 10 --warmup 2 > $ROOT/gpurun_out/ab_$v.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("$ROOT/gpurun_out/ab_$v/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "gn_match" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$v", {k: sum(v)/len(v) for k, v in acc.items()})
PY
  grep -o '"kernel_ms": [0-9.]*' $ROOT/gpurun_out/ab_$v.log
done

#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp_pc
mkdir -p "$OUT"
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "texel_cache or hint or batch" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
run() { # name, workload, env...
  name=$1; wl=$2; shift; shift
  env "$@" python $ROOT/bench.py --workload $wl --no-cpu --no-exact --no-pmc --no-pyramid --steps 30 > $OUT/t_$name.json 2>/dev/null
  env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$name -- python $ROOT/bench.py --workload $wl --no-cpu --no-exact --no-pmc --no-pyramid --steps 10 --warmup 2 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections, json
d=json.loads(open("$OUT/t_$name.json").read().strip().splitlines()[-1])
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/f_$name/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gn_match" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k: sum(v)/len(v) for k, v in acc.items()}
print("$name", "kernel_ms", round(d["roofline"]["kernel_ms"],4), "Mit/s", round(d["value"]/1e6,1), "FETCH_MB", round(m.get("FETCH_SIZE",0)/1024,1), d["config"]["kernel"]["layout"], d["config"]["kernel"]["texel_cache"])
PY
}
run c4_quad config4 A=1
run c4_plane config4 HSM_LAYOUT=plane
run c3_quad config3 A=1
run c3_plane config3 HSM_LAYOUT=plane
run c3pyr_quad config3pyr A=1
run c3pyr_plane config3pyr HSM_LAYOUT=plane

#!/bin/bash
# config4 (4096^2 pyramid): time + HBM-side fetch for layout / tiling variants
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp_c4
mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
run() { # name, env...
  name=$1; shift
  env "$@" python $ROOT/bench.py --workload config4 --no-cpu --no-exact --steps 30 > $OUT/t_$name.json 2>/dev/null
  env "$@" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f_$name -- python $ROOT/bench.py --workload config4 --no-cpu --no-exact --steps 10 --warmup 2 > /dev/null 2>&1
  env "$@" rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/h_$name -- python $ROOT/bench.py --workload config4 --no-cpu --no-exact --steps 10 --warmup 2 > /dev/null 2>&1
  python - <<PY
import csv, glob, collections, json
d=json.loads(open("$OUT/t_$name.json").read().strip().splitlines()[-1])
acc = collections.defaultdict(list)
for pat in ("$OUT/f_$name/*/*counter_collection.csv","$OUT/h_$name/*/*counter_collection.csv"):
    for f in glob.glob(pat):
        for r in csv.DictReader(open(f)):
            if "gn_match" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k: sum(v)/len(v) for k, v in acc.items()}
print("$name", "kernel_ms", round(d["roofline"]["kernel_ms"],4), "Mit/s", round(d["value"]/1e6,1), "FETCH_MB", round(m.get("FETCH_SIZE",0)/1024,1), "L2 hit", round(m.get("TCC_HIT_sum",0)/max(1,m.get("TCC_HIT_sum",0)+m.get("TCC_MISS_sum",0)),4), d["config"]["kernel"])
PY
}
run quad_cached A=1
run quad_nocache HSM_TEXEL_CACHE=0
run plane HSM_LAYOUT=plane
run tile_cached HSM_LIB=$ROOT/hector_slam_amd/lib/variants/libhector_mi355_tile.so
run noswizzle_cached HSM_XCD=0

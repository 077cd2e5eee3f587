#!/bin/bash
# (round 2 experiment; since round 4 the producer / chain-wavefront form and its HSM_EXACT_SHAPE / HSM_EXACT_BATCH switches exist in
#  -DHSM_EXPERIMENTS builds only: tools/build_variants.py x:"-DHSM_EXPERIMENTS" and HSM_LIB=<that library>)
# exact batch form: 7 producers + 1 consumer vs 8 + 2 per workgroup (HSM_EXACT_SHAPE), kernel time per launch
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/exact_shape
for wl in config3 config3pyr; do
 for shape in 7 8 7 8; do
  HSM_EXACT_SHAPE=$shape python bench.py --workload $wl --no-cpu --no-pmc --no-pyramid --no-pipelined --steps 50 > gpurun_out/exact_shape/${wl}_$shape.json 2> gpurun_out/exact_shape/${wl}_$shape.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/exact_shape/${wl}_$shape.json").read().strip().splitlines()[-1])
e=d["exact_parity"]
print("$wl shape $shape exact kernel_us", round(e["kernel_ms"]*1e3,1), "Mit/s", round(e["value"]/1e6,1), "fast_vs_exact", e.get("fast_vs_exact_all_scans",{}).get("bit_identical"))
PY
 done
done

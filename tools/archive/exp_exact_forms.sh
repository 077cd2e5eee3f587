#!/bin/bash
# (round 2 experiment; since round 4 the producer / chain-wavefront form and its HSM_EXACT_SHAPE / HSM_EXACT_BATCH switches exist in
#  -DHSM_EXPERIMENTS builds only: tools/build_variants.py x:"-DHSM_EXPERIMENTS" and HSM_LIB=<that library>)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/exp_exact
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_exact_parity.py tests/test_gpu_full_size.py -m gpu -q -x 2>&1 | tail -6
for eb in 1 0; do
for wl in config3 config3pyr config4; do
  HSM_EXACT_BATCH=$eb HSM_PARITY=exact timeout 300 python bench.py --workload $wl --no-cpu --no-exact --no-pmc --no-pyramid --steps 40 > $OUT/x_$wl.json 2>$OUT/x_$wl.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/x_$wl.json").read().strip().splitlines()[-1])
    print("exact_batch=$eb", "$wl", "kernel_us", round(d["roofline"]["kernel_ms"]*1e3,2), "Mit/s", round(d["value"]/1e6,1), d["config"]["kernel"]["block"], d["config"]["kernel"]["parity"])
except Exception as e:
    print("$wl failed", e, open("$OUT/x_$wl.err").read()[-400:])
PY
done
done

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/tmp
run() { # name, env..., args
  name=$1; shift
  env "$@" HSM_BENCH_DETAILS=$ROOT/gpurun_out/tmp/x_$name.json timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-cpu --no-exact ${EXTRA:-} > /dev/null 2>&1
  python - $name <<'PY'
import json,sys
d=json.load(open('gpurun_out/tmp/x_%s.json' % sys.argv[1])); t=d['timed_regions']
print(sys.argv[1], 'ms/step', [round(x,4) for x in t['ms_per_step']], 'kernel_ms', [round(x,4) for x in t['kernel_ms']])
PY
}
run plain A=1
run dist_nogather HSM_BENCH_FORCE_DIST=1 HSM_BENCH_NO_GATHER=1
EXTRA="--gather-bucket 20" run dist_bucket20 HSM_BENCH_FORCE_DIST=1
EXTRA="--gather-bucket 8" run dist_bucket8 HSM_BENCH_FORCE_DIST=1
EXTRA="--gather-bucket 8 --steps 200" run dist_bucket8_200 HSM_BENCH_FORCE_DIST=1
EXTRA="--steps 200" run dist_nogather_200 HSM_BENCH_FORCE_DIST=1 HSM_BENCH_NO_GATHER=1
EXTRA="--steps 200" run plain_200 A=1

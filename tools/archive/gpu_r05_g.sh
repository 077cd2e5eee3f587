#!/bin/bash
# round 5, call G: the dense line walk with the beams in the lanes (update_mark_free_lanes_kernel): every dense test, then the
# per-kernel durations of the configs[4] step with it and without, then configs[4] itself
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05g
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_full_size.py tests/test_property_hypothesis.py tests/test_gpu_parity.py tests/test_gpu_exact_parity.py -m gpu -q -x -k "dense or config5 or 65535 or serial or borders or beam" 2>&1 | tail -6
for lanes in 1 0; do
  echo "== HSM_MARK_LANES=$lanes"
  HSM_MARK_LANES=$lanes bash tools/exp_update_kernels.sh base 2>&1 | tee -a "$OUT/update_walk_lanes_kernel_us.txt"
  HSM_MARK_LANES=$lanes HSM_PARITY=fast HSM_BENCH_DETAILS=$OUT/bench_config5_lanes$lanes.json timeout 600 python bench.py --workload config5 --compact --no-pmc --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  config5 (HSM_PARITY=fast) ms/step', round(d['ms_per_step'],4), 'match', round(d['match_ms'],4), 'update', round(d['update_ms'],4))"
done
echo "== dense soak with the new walk"; timeout 600 python tests/tools/soak_dense.py 1500 --check 500 2>/dev/null | cut -c1-300

#!/bin/bash
# round 6, call C: the speculative-carry exact matcher (gn_match_spec_kernel) on hardware: parity suites, then timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06c
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_exact_parity.py -m gpu -x -q --durations=5 -k "dense or speculative" > "$OUT/pytest_dense.log" 2>&1; echo "dense rc=$? ($(( $(date +%s) - S )) s)"; tail -25 "$OUT/pytest_dense.log"
S=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_full_size.py -m gpu -x -q --durations=5 -k "dense or config5" > "$OUT/pytest_dense2.log" 2>&1; echo "dense2 rc=$? ($(( $(date +%s) - S )) s)"; tail -12 "$OUT/pytest_dense2.log"
echo "== config5 (16k beams, 8192^2): spec (default) vs literal dense"
for v in 1 0; do
  HSM_EXACT_SPEC=$v HSM_BENCH_DETAILS=$OUT/bench_config5_spec$v.json timeout 600 python bench.py --workload config5 --compact --no-pmc --no-cpu > "$OUT/bench_config5_spec$v.out" 2> "$OUT/bench_config5_spec$v.err"
  echo "spec=$v rc=$?"; tail -1 "$OUT/bench_config5_spec$v.out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','match_ms','update_ms')}, d['config'].get('parity_effective'))"
done
echo "== config2 (one 1081-beam scan): default team form vs spec form forced (HSM_EXACT_DENSE_MIN=512)"
for v in 4096 512; do
  HSM_EXACT_DENSE_MIN=$v HSM_BENCH_DETAILS=$OUT/bench_config2_min$v.json timeout 600 python bench.py --workload config2 --compact --no-pmc --no-cpu > "$OUT/bench_config2_min$v.out" 2> "$OUT/bench_config2_min$v.err"
  echo "dense_min=$v rc=$?"; tail -1 "$OUT/bench_config2_min$v.out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d.get(k) for k in ('value','ms_per_step','latency_us')})"
done

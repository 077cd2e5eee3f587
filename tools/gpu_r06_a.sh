#!/bin/bash
# round 6, call A: the device-side pose exchange on hardware -- its GPU tests, the group tests, then bench.py at N = 1 (the
# driver's command), one rank with a process group (HSM_BENCH_FORCE_DIST=1) and two ranks sharing the device (self-launched)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r06a
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
S=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_exchange.py -m gpu -x -q --durations=5 > "$OUT/pytest_exchange.log" 2>&1; echo "exchange rc=$? ($(( $(date +%s) - S )) s)"; tail -12 "$OUT/pytest_exchange.log"
S=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "group" --durations=5 > "$OUT/pytest_group.log" 2>&1; echo "group rc=$? ($(( $(date +%s) - S )) s)"; tail -8 "$OUT/pytest_group.log"
echo "== driver bench command (N=1)"; S=$(date +%s)
HSM_BENCH_DETAILS=$OUT/bench_n1_details.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --pmc-dump "$OUT/pmc" > "$OUT/bench_n1.out" 2> "$OUT/bench_n1.err"; echo "bench rc=$? ($(( $(date +%s) - S )) s)"; tail -1 "$OUT/bench_n1.out"; tail -3 "$OUT/bench_n1.err"
echo "== one rank, nccl process group, direct gather per match"; S=$(date +%s)
HSM_BENCH_FORCE_DIST=1 HSM_BENCH_DETAILS=$OUT/bench_force_dist_details.json timeout 600 python bench.py --gpus 1 --steps 200 --warmup 10 --no-pmc --no-cpu > "$OUT/bench_force_dist.out" 2> "$OUT/bench_force_dist.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"; tail -1 "$OUT/bench_force_dist.out"; tail -3 "$OUT/bench_force_dist.err"
echo "== two ranks sharing the device (self-launch)"; S=$(date +%s)
HSM_BENCH_SHARE_GPU=1 HSM_BENCH_DETAILS=$OUT/bench_2ranks_details.json timeout 600 python bench.py --gpus 2 --steps 100 --warmup 10 --no-pmc --no-cpu > "$OUT/bench_2ranks.out" 2> "$OUT/bench_2ranks.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"; tail -1 "$OUT/bench_2ranks.out"; tail -5 "$OUT/bench_2ranks.err"
echo "== group bench (2 replicas on device 0): direct / rccl / peer"; S=$(date +%s)
timeout 600 python bench.py --group 2 --steps 50 > "$OUT/bench_group2.out" 2> "$OUT/bench_group2.err"; echo "rc=$? ($(( $(date +%s) - S )) s)"; tail -1 "$OUT/bench_group2.out" | cut -c1-1500; tail -3 "$OUT/bench_group2.err"

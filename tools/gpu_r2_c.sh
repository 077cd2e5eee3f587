#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/c
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
tail -3 "$OUT/bench_default.time"; tail -5 "$OUT/bench_default.err"
export HSM_BENCH_SHARE_GPU=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 > "$OUT/mr2_config3.json" 2> "$OUT/mr2_config3.err"; tail -3 "$OUT/mr2_config3.err"
cut -c1-300 $OUT/mr2_config3.json

#!/bin/bash
# round-2 GPU call B: the new bench line (default), the extra workloads, multi-rank code paths on one GPU (gloo)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/b
mkdir -p "$OUT"
cd "$ROOT"
( time timeout 600 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2> "$OUT/bench_default.time"
tail -3 "$OUT/bench_default.time"; tail -5 "$OUT/bench_default.err"
for w in config3pyr config4 config2 config5; do
  timeout 600 python bench.py --workload $w > "$OUT/bench_$w.json" 2> "$OUT/bench_$w.err"; tail -2 "$OUT/bench_$w.err"
done
export HSM_BENCH_SHARE_GPU=1
for w in config3 config4 config5; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --workload $w --steps 10 --warmup 2 > "$OUT/mr2_$w.json" 2> "$OUT/mr2_$w.err"; tail -3 "$OUT/mr2_$w.err"
done
cd $OUT; for f in *.json; do echo "== $f"; cut -c1-400 $f | tail -1; done

#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05g; mkdir -p "$OUT"; cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_dense_edges.py tests/test_gpu_full_size.py -m gpu -q -x -k "dense or config5" 2>&1 | tail -3
bash tools/exp_update_kernels.sh base lanes_nomem 2>&1 | tee -a "$OUT/update_walk_lanes_variants_kernel_us.txt"
HSM_MARK_LANES=0 bash tools/exp_update_kernels.sh base 2>&1 | tee -a "$OUT/update_walk_lanes_variants_kernel_us.txt"

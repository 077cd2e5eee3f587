#!/bin/bash
# round 5, call F: what a binned LDS tile rasteriser could gain on the dense line walk -- timing-only builds of the walk with no
# memory operation / with its marks in LDS bytes (per-kernel durations of the configs[4] step, rocprofv3 kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05f
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
bash tools/exp_update_kernels.sh base walk_nomem walk_lds base 2>&1 | tee "$OUT/update_walk_whatif_kernel_us.txt"

#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench workload.
# usage: tools/gpu_profile.sh <tag> [extra bench args]
# Output: gpurun_out/<tag>/{stats,pmc_*}/... (CSV); copy the summaries to profiles/ afterwards.
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu --no-pyramid --no-pmc --no-exact --steps 20 --warmup 3 $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH > "$OUT/stats.log" 2>&1
if [ "${LIGHT:-0}" = "1" ]; then PMCS=("FETCH_SIZE" "WRITE_SIZE"); else PMCS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"); fi
for pmc in "${PMCS[@]}"; do
  name=$(echo "$pmc" | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$OUT/pmc_$name" -- $BENCH > "$OUT/pmc_$name.log" 2>&1 || echo "pmc $name failed" >> "$OUT/errors.log"
done
for pmc in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d "$OUT/calib_$pmc" -- python $ROOT/tools/calib_traffic.py > "$OUT/calib_$pmc.log" 2>&1
done
cd "$ROOT"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"

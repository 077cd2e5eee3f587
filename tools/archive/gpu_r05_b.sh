#!/bin/bash
# round 5, call B: the block-scan form of the reference's fp32 chains (exact_scan.h) against the literal chain on real per-beam
# products (one wavefront each; bits must agree), and the scene-sweep families call A did not reach.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05b
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
for nb in 16384 1081 4096; do python tools/study/binade_stats.py $nb --dump > /dev/null 2>&1; done
ls -la /tmp/products_*.bin
for f in /tmp/products_16384_0_start.bin /tmp/products_16384_0_converged.bin /tmp/products_16384_1_start.bin /tmp/products_16384_1_converged.bin \
         /tmp/products_4096_0_start.bin /tmp/products_4096_1_converged.bin \
         /tmp/products_1081_0_start.bin /tmp/products_1081_0_converged.bin /tmp/products_1081_1_start.bin /tmp/products_1081_1_converged.bin; do
  n=$(( $(stat -c %s $f) / 36 ))
  echo "== $f n=$n"
  timeout 120 tools/_bin/ubench_exact_scan $f $n 9 | tee -a "$OUT/ubench_exact_scan.jsonl"; echo "rc=$?"
done
echo "== scene sweep: remaining families"; S=$(date +%s)
timeout 1500 python tools/parity_scene_sweep.py --batch 4096 --sample 256 --single-all --no-single --families 30pct_beams_out_of_map,coarse_cells_res02_noisy --out "$OUT/parity_scene_sweep_single_default.jsonl" 2>&1 | grep -v '^$' | tail -30
echo "($(( $(date +%s) - S )) s)"

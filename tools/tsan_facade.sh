#!/bin/bash
# ThreadSanitizer over the drop-in facade's two-thread contract (HectorMappingRos.cpp:435-476,577-595): the scan thread
# matches and updates while a publisher thread fetches getGridMap(0) and reads every cell under the map mutex.
# Runs oracle/_ref/tsan/slam_driver_mi355 (make -C oracle tsan) on the GPU box; output -> gpurun_out/tsan/
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/tsan
mkdir -p "$OUT"
cd "$ROOT"
python - <<PY
import sys, os
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
from hector_slam_amd import synth
from test_facade_dropin import write_scenario
sc = synth.make_scene(n_beams=1081, map_size=512, levels=3, resolution=0.05, n_build=80, n_query=1, room=(20.0, 15.0), seed=99)
write_scenario("$OUT/s.bin", sc, 60, hooks=2)
PY
TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4" setarch $(uname -m) -R \
  timeout 300 oracle/_ref/tsan/slam_driver_mi355 "$OUT/s.bin" "$OUT/out.bin" > "$OUT/tsan_stdout.txt" 2> "$OUT/tsan_stderr.txt"
echo "rc=$?" >> "$OUT/tsan_stdout.txt"
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $OUT/tsan_stderr.txt)" | tee -a "$OUT/tsan_stdout.txt"
grep -A12 "WARNING: ThreadSanitizer" "$OUT/tsan_stderr.txt" | head -80
tail -3 "$OUT/tsan_stdout.txt"

#!/bin/bash
# round 5, call D: the producers-ahead-of-the-chain form of the exact dense matcher (gn_match_exact_dense_kernel): its test, the
# suite with 60 hypothesis examples (the reference's own crashes are now discarded inputs), configs[4] with it and without.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05d
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$ROOT"
timeout 600 python -m pytest tests/test_gpu_exact_parity.py -m gpu -q -x -k "dense" 2>&1 | tail -4
S=$(date +%s)
echo skip-suite
true
for dense in 1 0; do
  echo "== configs[4], HSM_EXACT_DENSE=$dense"
  HSM_EXACT_DENSE=$dense HSM_BENCH_DETAILS=$OUT/bench_config5_dense$dense.json timeout 900 python bench.py --workload config5 --compact --no-pmc --no-cpu 2> "$OUT/config5_$dense.err" | tail -1 | cut -c1-200
  python - "$OUT/bench_config5_dense$dense.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("  ms/step", round(d["ms_per_step"], 4), "match", round(d["match_ms"], 4), "update", round(d["update_ms"], 4), "kernel", d["config"]["kernel"].get("kernel"), "| fast:", round(d["fast_mode"]["ms_per_step"], 4))
PY
done
echo "== dense exact match latency by beams (host call, 3-level 2048^2 pyramid), dense form vs team form"
python - <<'PY'
import os, time, numpy as np
from hector_slam_amd import capi, synth
sc = synth.make_scene(n_beams=1081, map_size=2048, levels=3, resolution=0.05, n_build=60, n_query=4, room=(40.0, 30.0), seed=9)
s = float(np.float32(1.0) / np.float32(sc.resolution))
rng = np.random.default_rng(1)
for n in (2048, 4096, 8192, 16384, 32768):
    pts = synth.make_scan(sc.world, sc.query_truth[0], n, s, rng, pad_to_full=True)
    row = {}
    for dense in ("1", "0"):
        os.environ["HSM_EXACT_DENSE"] = dense
        os.environ["HSM_EXACT_DENSE_MIN"] = "1920"
        g = capi.MapRepMultiMap(sc.resolution, sc.map_size, sc.map_size, sc.levels)
        g.setUpdateFactorFree(0.4); g.setUpdateFactorOccupied(0.9)
        g.build_map(sc.build_poses, sc.build_scans)
        lat = []
        for k in range(30):
            a = time.perf_counter(); p, c = g.matchData(sc.query_init[0], pts); lat.append(time.perf_counter() - a)
        row[dense] = (float(np.median(lat[5:])) * 1e6, g.last_launch_config()["kernel"], p.copy())
        g.close()
    print(n, "beams: dense form", round(row["1"][0], 1), "us", row["1"][1], "| team form", round(row["0"][0], 1), "us", row["0"][1], "| same bits", bool((row["1"][2].view(np.uint32) == row["0"][2].view(np.uint32)).all()),
          "| chain floor", round(n * 14 * 8.5 / 2400, 1), "us at 2.4 GHz")
PY

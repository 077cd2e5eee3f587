#!/bin/bash
# quad layout: texels written by the apply pass (HSM_SCATTER_TEXELS_MAX) vs the separate texel pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/scatter
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
keys = ("update_latency_us", "update_complete_us", "slam_cycle_us", "latency_us", "match_ms", "update_ms", "ms_per_step")
print(sys.argv[2], {k: (round(d[k]["median"], 1) if isinstance(d[k], dict) else round(d[k], 4)) for k in keys if k in d},
      "node_loop", d.get("node_loop_cpp", {}).get("mi355x_dropin", {}).get("median_us"))
PY
}
for rep in 1 2; do
 for mx in 0 4096; do
  HSM_SCATTER_TEXELS_MAX=$mx python bench.py --workload config2 --no-cpu > gpurun_out/scatter/c2_$mx.json 2> gpurun_out/scatter/c2_$mx.err; show gpurun_out/scatter/c2_$mx.json "config2 quad scatter_max=$mx"
 done
done
for mx in 0 1000000; do
  HSM_LAYOUT=quad HSM_SCATTER_TEXELS_MAX=$mx python bench.py --workload config5 --no-cpu > gpurun_out/scatter/c5_$mx.json 2> gpurun_out/scatter/c5_$mx.err; show gpurun_out/scatter/c5_$mx.json "config5 quad scatter_max=$mx"
done
python bench.py --workload config5 --no-cpu > gpurun_out/scatter/c5_plane.json 2> gpurun_out/scatter/c5_plane.err; show gpurun_out/scatter/c5_plane.json "config5 plane"

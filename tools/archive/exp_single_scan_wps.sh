#!/bin/bash
# single-scan matchData latency vs waves per scan (HSM_WPS) and layout
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for lay in plane quad; do for w in 2 4 8 16; do
HSM_LAYOUT=$lay HSM_WPS=$w python bench.py --workload config2 --steps 300 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lay wps $w match_us', round(d['latency_us']['median'],1), 'min', round(d['latency_us']['min'],1), 'cycle', round(d['slam_cycle_us']['median'],1), d['config']['kernel']['beams_per_lane'])"
done; done

#!/bin/bash
# experimental two-waves-per-scan texel-cache form vs the default (one wave per scan)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for wl in config3 config3pyr; do for v in base w2 base w2; do
 if [ $v = w2 ]; then export HSM_CACHED_WPS2=1 HSM_WPS=2; else unset HSM_CACHED_WPS2 HSM_WPS; fi
 python bench.py --workload $wl --no-cpu --no-pmc --no-pyramid --no-pipelined --no-exact --steps 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $v kernel_us', round(d['roofline']['kernel_ms']*1e3,1), 'Mit/s', round(d['value']/1e6,1), d['config']['kernel'].get('waves_per_scan'), d['config']['kernel'].get('texel_cache'))"
done; done

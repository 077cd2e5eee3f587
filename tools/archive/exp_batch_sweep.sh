#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for b in 1024 2048 4096 8192 16384 32768; do
  python bench.py --batch $b --no-cpu --no-exact --no-pmc --no-pyramid --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch',$b,'kernel_us',round(d['roofline']['kernel_ms']*1e3,2),'us per 4096 scans',round(d['roofline']['kernel_ms']*1e3*4096/$b,2),'Mit/s',round(d['value']/1e6,1))"
done

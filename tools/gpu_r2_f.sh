#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
HSM_HOST_PROFILE=1 HSM_LAYOUT=plane timeout 300 python bench.py --workload config2 --steps 300 --no-cpu 2>&1 >/dev/null | grep HOSTPROF | tail -3

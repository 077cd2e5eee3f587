#!/bin/bash
# run a pytest selection on the GPU box: tools/gpu_quick.sh "<pytest args>"
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
timeout 1200 python -m pytest $1 -m gpu -q -x 2>&1 | tail -${2:-12}

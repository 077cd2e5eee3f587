#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "group or facade or hint" 2>&1 | tail -15

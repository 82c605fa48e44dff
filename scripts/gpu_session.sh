#!/bin/bash
set -u
TAG=${1:-s10}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wider or c1_size" > $OUT/${TAG}_pytest.log 2>&1
grep -n "^E   \|^E       Assert\|passed\|failed\|^tests/.*Error" $OUT/${TAG}_pytest.log | cut -c1-900 | head -30

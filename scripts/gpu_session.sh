#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
df -h /dev/shm | tail -1
timeout 900 python -m pytest tests/test_rccl_stub.py -q -m gpu -x 2>&1 | tail -30

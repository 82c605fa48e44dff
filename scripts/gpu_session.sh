#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rccl_stub.py tests/test_gpu_parity.py -q -m gpu -x -k "stub or expensive_bounds" --durations=5 2>&1 | tail -25

#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 200 python -m pytest tests -m gpu -x -q -k "partitioned_q1_annulus" > gpurun_out/r04q4_pytest.log 2>&1
tail -15 gpurun_out/r04q4_pytest.log

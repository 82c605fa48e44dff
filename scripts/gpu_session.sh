#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -q -m gpu -x --durations=12 2>&1 | tail -30 > gpurun_out/r05f_pytest_gpu.txt
cat gpurun_out/r05f_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

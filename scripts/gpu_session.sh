#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -24 > gpurun_out/r05n_pytest_gpu.txt
cat gpurun_out/r05n_pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

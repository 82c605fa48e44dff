#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 120 python bench.py > gpurun_out/r05w_bench_as_the_driver_runs_it.json 2> gpurun_out/r05w_bench.err
tail -c 300 gpurun_out/r05w_bench_as_the_driver_runs_it.json
timeout 900 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r05w_pytest_gpu.txt 2>&1
tail -3 gpurun_out/r05w_pytest_gpu.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 100 python -m pytest tests/test_gpu_parity_fullsize.py tests/test_gpu_parity.py -q -m gpu -x -k "test_c2_bench_state or (large_meshes and euler_2d)" 2>&1 | tail -3 | tee gpurun_out/r05z_pytest_focus.txt
grep -q "passed" gpurun_out/r05z_pytest_focus.txt && ! grep -q "failed" gpurun_out/r05z_pytest_focus.txt || exit 1
bash scripts/profile_round.sh r05z step2d 2>&1 | tail -3

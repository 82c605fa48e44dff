#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
rm -f gpurun_out/r05_parity_stats.jsonl
RYUJIN_PARITY_STATS=$R/gpurun_out/r05_parity_stats.jsonl timeout 1200 python -m pytest tests/test_gpu_parity_fullsize.py -q -m gpu 2>&1 | tail -4
wc -l gpurun_out/r05_parity_stats.jsonl

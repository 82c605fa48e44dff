#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-r04o}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
timeout 60 python scripts/overhead_loopback.py --dim 2 > $OUT/${TAG}_overhead_loopback.log 2>&1
timeout 90 python scripts/overhead_loopback.py --dim 3 >> $OUT/${TAG}_overhead_loopback.log 2>&1
tail -8 $OUT/${TAG}_overhead_loopback.log
OMP_NUM_THREADS=16 timeout 100 python scripts/long_run_compare.py 40 200 > $OUT/${TAG}_long_run_compare.log 2>&1
tail -8 $OUT/${TAG}_long_run_compare.log

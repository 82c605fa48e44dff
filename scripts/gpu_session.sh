#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-r04p}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof -- python $R/bench.py --cells-per-unit 130 --no-cpu-baseline --steps 60 --warmup 6 > /tmp/prof.log 2>&1
grep -h "^{" /tmp/prof.log | head -1 > $OUT/${TAG}_bench_profiled_c1.json
python $R/scripts/rocpd_summary.py /tmp/prof/*/*.db "$TAG kernel trace (C1, 43 109 gridpoints): rocprofv3 --kernel-trace --stats -- python bench.py --cells-per-unit 130 --no-cpu-baseline --steps 60 --warmup 6" > $OUT/${TAG}_kernel_trace_c1.md
head -24 $OUT/${TAG}_kernel_trace_c1.md | cut -c1-150
python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench_profiled_c1.json').read()); print(d['ms_per_step'], d['sweep_ms'])"

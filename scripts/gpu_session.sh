#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
rm -f gpurun_out/parity_stats.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x --durations=15 2>&1 | tail -45 > gpurun_out/r05b_pytest_gpu.txt
tail -45 gpurun_out/r05b_pytest_gpu.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r05b_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], d['cpu_baseline']['value'], d['sweep_ms'])"
tail -3 gpurun_out/r05b_bench.err

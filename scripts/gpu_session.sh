#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04r_bench_as_the_driver_runs_it.json 2> gpurun_out/r04r_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r04r_bench_as_the_driver_runs_it.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['metric'], d['value'], d['unit'], d['n_gpus'], d['steps'], d['warmup'], d['ms_per_step'], r['frac'], r['traffic_frac'], r['valu']['issue_frac'], d['cpu_baseline']['value'], d['config']['simulated_time_at_start'])"
tail -3 gpurun_out/r04r_bench.err

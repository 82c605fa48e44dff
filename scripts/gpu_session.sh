#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_binding_run.py tests/test_shim_cpp.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r05a_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], d['cpu_baseline']['value']); print(json.dumps(d['binding'], indent=1))"
tail -3 gpurun_out/r05a_bench.err

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dg_q1.py tests/test_gpu_parity_fullsize.py -q -m gpu -x -k "sw or shallow or multistage or wider or dg or erk or c5" 2>&1 | tail -4
for w in sw2d; do
  timeout 600 python bench.py --workload $w --steps 18 --warmup 6 --no-cpu-baseline --binding device > gpurun_out/r05h_bench_$w.json 2> gpurun_out/r05h_bench_$w.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r05h_bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$w', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['frac'],3), d['sweep_ms'])"
done

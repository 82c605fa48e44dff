#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
echo "== correctness subset"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_fullsize.py -q -m gpu -x -k "3d or cylinder or radial or c4 or c3 or c2_bench or tetrahedra or rows_wider or partition" 2>&1 | tail -5
for w in step2d cylinder3d sedov3d; do
  timeout 600 python bench.py --workload $w --steps 18 --warmup 6 --no-cpu-baseline --binding device > gpurun_out/r05d_bench_$w.json 2> gpurun_out/r05d_bench_$w.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r05d_bench_$w.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$w', round(d['value'],1), round(d['ms_per_step'],4), r['kernel'], round(r['frac'],3), d['limiter']['limited_slice_fraction'], d['sweep_ms'])"
done

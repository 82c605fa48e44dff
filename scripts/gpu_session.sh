#!/bin/bash
# One gpurun call of round 4 (scratch driver script; results under gpurun_out/<tag>_*)
set -u
TAG=${1:-s2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
L=ryujin_amd/lib
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q > $OUT/${TAG}_pytest.log 2>&1
tail -3 $OUT/${TAG}_pytest.log
timeout 600 python -m pytest tests/test_gpu_parity_fullsize.py -x -q -k "c2 or c4" > $OUT/${TAG}_pytest_c2.log 2>&1
tail -3 $OUT/${TAG}_pytest_c2.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench.json').read().splitlines()[0]); print(d['ms_per_step'], d['limiter'], d['sweep_ms'])"
timeout 600 python bench.py --develop-time 0 --develop 900 --save-state /tmp/c2_startup.npz --no-cpu-baseline --steps 6 --reps 1 > /dev/null 2>&1
V="auto=$L/libryujin_hip.so plain=$L/libryujin_hip.so:debug_pij_storage=-1 nopred=$L/libryujin_hip.so:debug_pij_storage=1 pslice=$L/variants/pslice.so"
timeout 900 python scripts/ab_variants.py --load-state /tmp/c2_startup.npz --perturbation 1e-3 \
  --perturbed-fractions 0,0.1,0.25,0.4,0.5,0.6,0.75,1.0 --steps 12 --rounds 2 $V > $OUT/${TAG}_ab_limited_fraction_2d.log 2>&1
grep -v "limiter statistics: {'limited_slice_fraction': [0-9.]*, 'pij_stored': 'everywhere'" $OUT/${TAG}_ab_limited_fraction_2d.log | tail -70
timeout 900 python scripts/ab_variants.py --workload sedov3d --develop 150 --perturbation 1e-3 \
  --perturbed-fractions 0,0.25,0.5,0.75,1.0 --steps 6 --rounds 2 $V > $OUT/${TAG}_ab_limited_fraction_3d.log 2>&1
tail -45 $OUT/${TAG}_ab_limited_fraction_3d.log

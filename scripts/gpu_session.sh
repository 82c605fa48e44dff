#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-s19}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
L=ryujin_amd/lib
timeout 600 python bench.py --workload cylinder3d --no-cpu-baseline --save-state /tmp/c4.npz --steps 6 --reps 1 > /dev/null 2>&1
timeout 900 python scripts/ab_variants.py --dim 3 --cells-per-unit 96 --load-state /tmp/c4.npz --steps 15 --rounds 3 base=$L/libryujin_hip.so pipe2=$L/variants/pipe2.so pipe3=$L/variants/pipe3.so > $OUT/${TAG}_ab_3d.log 2>&1
cut -c1-170 $OUT/${TAG}_ab_3d.log
RYUJIN_HIP_LIB=$L/variants/pipe2.so timeout 600 python bench.py --workload sedov3d --no-cpu-baseline --steps 30 --reps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[0]); print('sedov3d pipe2', round(d['ms_per_step'],4), d['sweep_ms'])"
timeout 600 python bench.py --workload sedov3d --no-cpu-baseline --steps 30 --reps 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().splitlines()[0]); print('sedov3d base', round(d['ms_per_step'],4), d['sweep_ms'])"
RYUJIN_HIP_LIB=$L/variants/pipe2.so timeout 600 python -m pytest tests -m gpu -x -q -k "3d or wide" > $OUT/${TAG}_pytest.log 2>&1
tail -2 $OUT/${TAG}_pytest.log

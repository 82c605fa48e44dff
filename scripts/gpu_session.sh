#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-s14}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
L=ryujin_amd/lib
timeout 900 python -m pytest tests -m gpu -x -q -k "parity or device or golden or conserv" > $OUT/${TAG}_pytest.log 2>&1
tail -5 $OUT/${TAG}_pytest.log
timeout 600 python bench.py --workload cylinder3d --no-cpu-baseline --save-state /tmp/c4.npz --steps 6 --reps 1 > /dev/null 2>&1
timeout 600 python bench.py --no-cpu-baseline --save-state /tmp/c2.npz --steps 6 --reps 1 > /dev/null 2>&1
for v in base rec rec_occ3 rec_nopf; do
  lib=$L/variants/$v.so; [ $v = rec ] && lib=$L/libryujin_hip.so; [ $v = base ] && lib=$L/variants/cp2.so
  RYUJIN_HIP_LIB=$lib timeout 600 python bench.py --workload cylinder3d --no-cpu-baseline --load-state /tmp/c4.npz --steps 30 --reps 3 > $OUT/${TAG}_c4_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$OUT/${TAG}_c4_$v.json').read().splitlines()[0]); print('c4 $v', round(d['ms_per_step'],4), d['sweep_ms'])"
  RYUJIN_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --load-state /tmp/c2.npz --steps 60 --reps 3 > $OUT/${TAG}_c2_$v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$OUT/${TAG}_c2_$v.json').read().splitlines()[0]); print('c2 $v', round(d['ms_per_step'],4), d['sweep_ms'])"
done

#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-s18}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
L=ryujin_amd/lib
timeout 600 python bench.py --no-cpu-baseline --save-state /tmp/c2.npz --steps 6 --reps 1 > /dev/null 2>&1
timeout 600 python bench.py --workload cylinder3d --no-cpu-baseline --save-state /tmp/c4.npz --steps 6 --reps 1 > /dev/null 2>&1
timeout 900 python scripts/ab_variants.py --load-state /tmp/c2.npz --steps 30 --rounds 4 base=$L/variants/base.so xcd=$L/variants/xcd.so sel=$L/libryujin_hip.so > $OUT/${TAG}_ab_2d.log 2>&1
cut -c1-170 $OUT/${TAG}_ab_2d.log
timeout 900 python scripts/ab_variants.py --dim 3 --cells-per-unit 96 --load-state /tmp/c4.npz --steps 15 --rounds 4 base=$L/variants/base.so xcd=$L/variants/xcd.so sel=$L/libryujin_hip.so > $OUT/${TAG}_ab_3d.log 2>&1
cut -c1-170 $OUT/${TAG}_ab_3d.log
timeout 900 python -m pytest tests -m gpu -x -q -k "parity_2d or parity_3d or partitioned or c1_size or large_meshes or unstructured" > $OUT/${TAG}_pytest.log 2>&1
tail -3 $OUT/${TAG}_pytest.log

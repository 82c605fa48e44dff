#!/bin/bash
# One gpurun call of round 4 (scratch driver script; results under gpurun_out/<tag>_*)
set -u
TAG=${1:-s5}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
L=ryujin_amd/lib
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -k "c1_size or unstructured_p1 or large_meshes or partitioned_hip or sw_ or step_parity or multistage or aeos" > $OUT/${TAG}_pytest.log 2>&1
tail -5 $OUT/${TAG}_pytest.log
timeout 600 python bench.py --no-cpu-baseline --save-state /tmp/c2_t2.npz > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench.json').read().splitlines()[0]); print(d['ms_per_step'], d['limiter'], d['sweep_ms'])"
V="compact=$L/libryujin_hip.so oldtail=$L/variants/oldtail.so"
timeout 900 python scripts/ab_variants.py --load-state /tmp/c2_t2.npz --steps 12 --rounds 3 $V > $OUT/${TAG}_ab_compact_tail_2d_t2.log 2>&1
tail -6 $OUT/${TAG}_ab_compact_tail_2d_t2.log
timeout 600 python bench.py --workload cylinder3d --no-cpu-baseline --save-state /tmp/c4.npz > $OUT/${TAG}_bench_cylinder3d.json 2> $OUT/${TAG}_bench_cylinder3d.err
python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench_cylinder3d.json').read().splitlines()[0]); print(d['ms_per_step'], d['limiter'], d['sweep_ms'])"
timeout 900 python scripts/ab_variants.py --dim 3 --cells-per-unit 96 --length 1.25 --load-state /tmp/c4.npz --steps 6 --rounds 2 $V > $OUT/${TAG}_ab_compact_tail_3d_c4.log 2>&1
tail -6 $OUT/${TAG}_ab_compact_tail_3d_c4.log

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
LIB=ryujin_amd/lib/libryujin_hip.so
V=ryujin_amd/lib/variants
timeout 300 python bench.py --steps 6 --warmup 3 --reps 1 --no-cpu-baseline --binding device --save-state /tmp/c2.npz > /dev/null 2> gpurun_out/r05s_save.err; tail -1 gpurun_out/r05s_save.err
echo "== 2-D t=2: per tile / per tile, every tile predicted (= stored everywhere through the tile path) / without step 6's counters / everywhere"
timeout 900 python scripts/ab_variants.py --load-state /tmp/c2.npz --steps 15 --rounds 4 tile=$LIB all_predicted_frozen=$LIB:debug_pij_storage=3 all_predicted_rewritten=$LIB:debug_pij_storage=4 everywhere=$LIB:debug_pij_storage=-1 2>&1 | tee gpurun_out/r05t_ab_tile_step6_overhead3.log | grep -v statistics | tail -6

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
LIB=ryujin_amd/lib/libryujin_hip.so
timeout 300 python bench.py --steps 6 --warmup 3 --reps 1 --no-cpu-baseline --binding device --save-state /tmp/c2.npz > /dev/null 2> gpurun_out/r05p_save.err; tail -1 gpurun_out/r05p_save.err
echo "== 2-D: stacked blocks"
timeout 900 python scripts/ab_variants.py --load-state /tmp/c2.npz --steps 15 --rounds 3 base=$LIB:debug_band_stride=-1 b47=$LIB:debug_band_stride=47 b46=$LIB:debug_band_stride=46 b93=$LIB:debug_band_stride=93 b12=$LIB:debug_band_stride=12 2>&1 | tee gpurun_out/r05p_ab_band_2d.log | grep -v "limiter statistics" | tail -8
echo "== 3-D cylinder share"
timeout 900 python scripts/ab_variants.py --dim 3 --cells-per-unit 96 --length 1.25 --develop 300 --steps 9 --rounds 3 base=$LIB:debug_band_stride=-1 b365=$LIB:debug_band_stride=365 b2=$LIB:debug_band_stride=2 b91=$LIB:debug_band_stride=91 2>&1 | tee gpurun_out/r05p_ab_band_3d.log | grep -v "limiter statistics" | tail -7

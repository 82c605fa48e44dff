#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
V=ryujin_amd/lib/variants
timeout 300 python bench.py --steps 6 --warmup 3 --reps 1 --no-cpu-baseline --binding device --save-state /tmp/c2.npz > /dev/null 2> gpurun_out/r05e_save.err; tail -1 gpurun_out/r05e_save.err
echo "== 2-D step 4: 2 waves / 3 waves / 3 waves with f_i in LDS"
timeout 600 python scripts/ab_variants.py --load-state /tmp/c2.npz --steps 15 --rounds 4 low2=$V/libryujin_hip_low2.so low3=$V/libryujin_hip_low3.so low3p=$V/libryujin_hip_low3p.so 2>&1 | tee gpurun_out/r05e_ab_low_order_2d.log | tail -8
echo "== 2-D EulerAEOS step 5: 2 waves / 3 waves"
timeout 600 python scripts/ab_variants.py --workload step2d_aeos --develop 450 --steps 15 --rounds 3 aeos2=$V/libryujin_hip_low2.so aeos3=$V/libryujin_hip_aeos3.so 2>&1 | tee gpurun_out/r05e_ab_aeos.log | tail -6

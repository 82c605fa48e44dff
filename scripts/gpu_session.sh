#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-r04i}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
timeout 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 1500 $OUT/${TAG}_bench.json
bash scripts/profile_round.sh r04h step2d_aeos 2>&1 | tail -14
timeout 600 python bench.py --no-cpu-baseline --save-state /tmp/c2.npz --steps 6 --reps 1 > /dev/null 2>&1
timeout 900 python scripts/ab_variants.py --load-state /tmp/c2.npz --steps 30 --rounds 4 base=ryujin_amd/lib/libryujin_hip.so occ2=ryujin_amd/lib/variants/lij0occ2.so > $OUT/${TAG}_ab_lij0_occ.log 2>&1
cat $OUT/${TAG}_ab_lij0_occ.log | cut -c1-170

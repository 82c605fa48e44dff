#!/bin/bash
# scratch driver of one gpurun call (rewritten per session): the round's final measurement set
set -u
TAG=${1:-r04k}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest_gpu_full.log 2>&1
tail -4 $OUT/${TAG}_pytest_gpu_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/profile_round.sh $TAG step2d cylinder3d sedov3d sw2d step2d_aeos > $OUT/${TAG}_profile_round.log 2>&1
tail -3 $OUT/${TAG}_profile_round.log | cut -c1-300
cp $OUT/${TAG}_pmc*.md $OUT/${TAG}_kernel_trace*.md profiles/
bash scripts/bench_lines.sh $TAG
timeout 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 900 $OUT/${TAG}_bench.json

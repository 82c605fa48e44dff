#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-r04h}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/${TAG}_pytest.log 2>&1
tail -5 $OUT/${TAG}_pytest.log
bash scripts/profile_round.sh $TAG step2d cylinder3d sedov3d sw2d 2>&1 | tail -80

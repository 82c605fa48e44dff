#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
for t in "" "201,8,8" "201,16,4" "64,8,8"; do
  echo "== tile=$t"
  RYUJIN_SYNTH_TILE=$t timeout 600 python bench.py --workload sedov3d --size 160 --steps 12 --warmup 6 --reps 3 --no-cpu-baseline --binding device 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), d['limiter']['limited_slice_fraction'], d['sweep_ms'])"
done 2>&1 | tee gpurun_out/r05m_tile_numbering_3d.log

#!/bin/bash
# scratch driver of one gpurun call (rewritten per session)
set -u
TAG=${1:-r04n}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
OUT=gpurun_out
mkdir -p $OUT
timeout 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 700 $OUT/${TAG}_bench.json
for w in cylinder3d sw2d step2d_aeos; do
  timeout 500 python bench.py --workload $w --no-cpu-baseline > $OUT/${TAG}_bench_$w.json 2> /dev/null
  python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench_$w.json').read().splitlines()[0]); r=d['roofline']; print('$w', round(d['ms_per_step'],4), r['kernel'], round(r['frac'],3), r['traffic_frac'], r['valu'] and round(r['valu']['issue_frac'],3))"
done

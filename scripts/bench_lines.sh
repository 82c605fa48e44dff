#!/bin/bash
# The round's bench lines, run on the GPU box:  gpurun -- 'bash scripts/bench_lines.sh r03q'
# One line per BASELINE.json configuration and per stress case, each the driver's own command line plus the
# workload switch; written to gpurun_out/<tag>_bench_<name>.json (copy to profiles/). The C2 line with the CPU
# baseline comes from profile_round.sh (<tag>_bench.json).
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
run() { # name, arguments
  local name=$1; shift
  timeout 600 python bench.py "$@" > "$OUT/${TAG}_bench_$name.json" 2> "$OUT/${TAG}_bench_$name.err"
  python - "$OUT/${TAG}_bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[0])
    print(sys.argv[2], "ms/update", round(d["ms_per_step"], 4), "roofline_update", round(d["roofline_update"]["frac"], 3),
          "dominant", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3))
except Exception as e:  # noqa: BLE001
    print(sys.argv[2], "FAILED", e)
PY
}
run c1 --cells-per-unit 130
run cylinder3d --workload cylinder3d --no-cpu-baseline
run sedov3d --workload sedov3d --no-cpu-baseline
run sw2d --workload sw2d --no-cpu-baseline
run step2d_aeos --workload step2d_aeos --no-cpu-baseline
run perturbed --perturbation 1e-3 --no-cpu-baseline
run 4x --cells-per-unit 1990 --no-cpu-baseline

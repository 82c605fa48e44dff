#!/bin/bash
# The round's bench lines, run on the GPU box:  gpurun -- 'bash scripts/bench_lines.sh r04a [name ...]'
# One line per BASELINE.json configuration and per named variant of the C2 state, each the driver's own command line
# plus the switches below; written to gpurun_out/<tag>_bench_<name>.json (copy to profiles/). The C2 line with the
# CPU baseline comes from profile_round.sh (<tag>_bench.json).
set -u
TAG=${1:-rXX}
shift || true
ONLY=" ${*:-} "
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd "$R"
run() { # name, arguments
  local name=$1; shift
  if [ "$ONLY" != "  " ] && [[ "$ONLY" != *" $name "* ]]; then return; fi
  timeout 600 python bench.py "$@" > "$OUT/${TAG}_bench_$name.json" 2> "$OUT/${TAG}_bench_$name.err"
  python - "$OUT/${TAG}_bench_$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[0])
    print(sys.argv[2], "ms/update", round(d["ms_per_step"], 4), "roofline_update", round(d["roofline_update"]["frac"], 3),
          "dominant", d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), "limiter", d["limiter"],
          "t", round(d["config"]["simulated_time_at_start"], 4))
except Exception as e:  # noqa: BLE001
    print(sys.argv[2], "FAILED", e)
PY
}
# the C2 state: the default line is the flow at t = 2.0 of the reference's run to t = 4.0; earlier and later times,
# the start-up phase (round 3's line: 900 updates from the uniform state) and the pessimistic perturbed state
run t1 --develop-time 1.0 --no-cpu-baseline
run t4 --develop-time 4.0 --no-cpu-baseline
run startup --develop-time 0 --develop 900 --no-cpu-baseline
run perturbed --perturbation 1e-3 --no-cpu-baseline
run c1 --cells-per-unit 130
run cylinder3d --workload cylinder3d --no-cpu-baseline
run sedov3d --workload sedov3d --no-cpu-baseline
run sw2d --workload sw2d --no-cpu-baseline
run step2d_aeos --workload step2d_aeos --no-cpu-baseline
run 4x --cells-per-unit 1990 --no-cpu-baseline

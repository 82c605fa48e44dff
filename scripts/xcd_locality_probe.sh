#!/bin/bash
# XCD-local block ranges (ryujin_hip_params::debug_xcd_chunk) and the mesh numbering against the L2-miss traffic of the
# sweeps: for every variant one timing run of bench.py (per-sweep hipEvent times) and one rocprofv3 --pmc FETCH_SIZE
# pass, same developed state.  usage (GPU box): bash scripts/xcd_locality_probe.sh <tag> <workload> "<chunk> ..." [tile]
#   -> gpurun_out/<tag>_xcd_probe_<workload>[_tile<tile>].md
set -u
TAG=$1; W=$2; CHUNKS=$3; TILE=${4:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
SUF=""; [ -n "$TILE" ] && { export RYUJIN_SYNTH_TILE=$TILE; SUF="_tile$(echo $TILE | tr , x)"; }
BENCH="python $R/bench.py --workload $W"
STATE=/tmp/state_${W}${SUF}.npz
[ -f $STATE ] || timeout 900 $BENCH --save-state $STATE --steps 3 --warmup 0 --reps 1 --no-cpu-baseline --binding device > /tmp/state.log 2>&1
MD=$OUT/${TAG}_xcd_probe_${W}${SUF}.md
{
echo "# $TAG: XCD-local block ranges on $W${TILE:+, nodes numbered in tiles of $TILE}"
echo
echo "bench.py --workload $W --load-state <developed state> --steps 30 --warmup 6 --reps 3 (per-sweep hipEvent means, ms);"
echo "FETCH = rocprofv3 --pmc FETCH_SIZE of the same command with --steps 6, mean per dispatch, 2 x FETCH_SIZE x 1024 bytes, in MB"
echo
echo "| debug_xcd_chunk | ms/update | step 2 | 3 | 4 | 5 | 6 | 7 | FETCH MB: step 2 | 3 | 4 | 5 | 6 | 7 |"
echo "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"
} > $MD
for C in $CHUNKS; do
  export RYUJIN_XCD_CHUNK=$C
  timeout 600 $BENCH --load-state $STATE --steps 30 --warmup 6 --reps 3 --no-cpu-baseline --binding device > /tmp/t.json 2> /tmp/t.err
  rm -rf /tmp/pmc_probe
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_probe -- $BENCH --load-state $STATE --steps 6 --warmup 3 --reps 1 --no-cpu-baseline --binding device > /tmp/p.log 2>&1
  python - "$C" /tmp/t.json /tmp/pmc_probe/*/*.db >> $MD <<'PY'
import json, sqlite3, sys
c, tj, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
try:
    d = json.loads([l for l in open(tj) if l.startswith("{")][0])
    sw = d["sweep_ms"]
    t = [d["ms_per_step"]] + [sw[k] for k in sorted(sw) if k[0] in "234567"]
except Exception as e:
    t = [float("nan")] * 7
fetch = {}
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    for k, v, n in cur.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name"):
        fetch[k.split("(")[0].replace("void ", "").replace("ryujin_hip::", "")] = (v * 2 * 1024 / 1e6, n)
def best(prefixes, last=None):
    cand = [(v[1] * v[0], v[0]) for k, v in fetch.items() if k.startswith(prefixes) and (last is None or ("true" in k) == last or "cached" in k)]
    return max(cand)[1] if cand else float("nan")
f = [best(("k_dij_alpha",)), best(("k_dij_diag",)), best(("k_low_order",)), best(("k_lij_stage0", "k_pij_lij")),
     best(("k_high_order_next_cached",)), best(("k_high_order_last_cached",))]
print("| " + c + " | " + " | ".join("%.4f" % x for x in t) + " | " + " | ".join("%.0f" % x for x in f) + " |")
PY
done
cat $MD

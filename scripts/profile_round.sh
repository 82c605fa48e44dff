#!/bin/bash
# Round profile set, run on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r01f'
# Writes gpurun_out/<tag>_{bench.json,bench_profiled.json,kernel_trace.md,pmc.md}; copy them to profiles/.
# PMC counters are collected in their own passes, without any trace domain (see MI355X_MICROARCH.md).
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
  n=$(echo $c | cut -d" " -f1)
  rm -rf /tmp/pmc_$n
  timeout 250 rocprofv3 --pmc $c -d /tmp/pmc_$n -- $BENCH --steps 6 --warmup 3 --develop 300 --no-cpu-baseline > /tmp/pmc_$n.log 2>&1
done
python $R/scripts/pmc_summary.py "$TAG PMC: rocprofv3 --pmc <counters> -- python bench.py --steps 6 --warmup 3 --develop 300 --no-cpu-baseline (2.50M gridpoints, developed Mach-3 step flow; FETCH_SIZE, WRITE_SIZE and SQ counters in three separate passes)" /tmp/pmc_*/*/*.db > $OUT/${TAG}_pmc.md
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -- $BENCH --steps 60 --warmup 12 --develop 300 --no-cpu-baseline > /tmp/prof.log 2>&1
grep -h "^{" /tmp/prof.log | head -1 > $OUT/${TAG}_bench_profiled.json
python $R/scripts/rocpd_summary.py /tmp/prof/*/*.db "$TAG kernel trace: rocprofv3 --kernel-trace --stats -- python bench.py --steps 60 --warmup 12 --develop 300 --no-cpu-baseline" > $OUT/${TAG}_kernel_trace.md
cd $R
timeout 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json

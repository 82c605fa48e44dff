#!/bin/bash
# Round profile set, run on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r02d [workload ...]'
# For every workload (default: step2d; others: sedov3d cylinder3d sw2d step2d_aeos) writes
#   gpurun_out/<tag>_{kernel_trace,pmc}[_<workload>].md  and  gpurun_out/<tag>_bench_profiled[_<workload>].json
# plus, for step2d, the plain bench line gpurun_out/<tag>_bench.json. Copy them to profiles/.
# PMC counters are collected in their own passes, without any trace domain (see MI355X_MICROARCH.md).
set -u
TAG=${1:-rXX}
shift || true
WORKLOADS=${*:-step2d}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for W in $WORKLOADS; do
  SUF=""; [ "$W" != step2d ] && SUF="_$W"
  BENCH="python $R/bench.py --workload $W"
  PMCARGS="--steps 6 --warmup 3 --develop 900 --no-cpu-baseline"
  rm -rf /tmp/pmc_*
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
    n=$(echo $c | cut -d" " -f1)
    timeout 400 rocprofv3 --pmc $c -d /tmp/pmc_$n -- $BENCH $PMCARGS > /tmp/pmc_$n.log 2>&1
  done
  python $R/scripts/pmc_summary.py "$TAG PMC ($W): rocprofv3 --pmc <counters> -- python bench.py --workload $W $PMCARGS (developed flow; FETCH_SIZE, WRITE_SIZE and SQ counters in three separate passes)" /tmp/pmc_*/*/*.db > $OUT/${TAG}_pmc$SUF.md
  rm -rf /tmp/prof
  TRARGS="--steps 30 --warmup 6 --develop 900 --no-cpu-baseline"
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof -- $BENCH $TRARGS > /tmp/prof.log 2>&1
  grep -h "^{" /tmp/prof.log | head -1 > $OUT/${TAG}_bench_profiled$SUF.json
  python $R/scripts/rocpd_summary.py /tmp/prof/*/*.db "$TAG kernel trace ($W): rocprofv3 --kernel-trace --stats -- python bench.py --workload $W $TRARGS" > $OUT/${TAG}_kernel_trace$SUF.md
  head -12 $OUT/${TAG}_kernel_trace$SUF.md
done
cd $R
case " $WORKLOADS " in *" step2d "*)
  timeout 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  tail -c 600 $OUT/${TAG}_bench.json;;
esac

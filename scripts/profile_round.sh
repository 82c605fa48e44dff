#!/bin/bash
# Round profile set, run on the GPU box:  gpurun -- 'bash scripts/profile_round.sh r04a [workload ...]'
# For every workload (default: step2d; others: sedov3d cylinder3d sw2d step2d_aeos) writes
#   gpurun_out/<tag>_{kernel_trace,pmc}[_<workload>].md  and  gpurun_out/<tag>_bench_profiled[_<workload>].json
# plus, for step2d, the plain bench line gpurun_out/<tag>_bench.json. Copy them to profiles/.
# The developed state of the workload (bench.py's default: coarse run, interpolation, re-sharpening) is made once
# and loaded by the profiled passes, so that every dispatch they count runs on the benchmark mesh.
# PMC counters are collected in their own passes, without any trace domain (see MI355X_MICROARCH.md). Both
# summaries carry the fingerprint of the kernel sources they were taken with (bench.py refuses a stale one).
set -u
TAG=${1:-rXX}
shift || true
WORKLOADS=${*:-step2d}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
FP=$(cd $R && python -c "import bench; print(bench.source_fingerprint())")
for W in $WORKLOADS; do
  SUF=""; [ "$W" != step2d ] && SUF="_$W"
  BENCH="python $R/bench.py --workload $W"
  STATE=/tmp/state_$W.npz
  timeout 600 $BENCH --save-state $STATE --steps 3 --warmup 0 --reps 1 --no-cpu-baseline --binding device > /tmp/state_$W.log 2>&1
  PMCARGS="--steps 6 --warmup 3 --load-state $STATE --no-cpu-baseline --binding device"
  rm -rf /tmp/pmc_*
  for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU"; do
    n=$(echo $c | cut -d" " -f1)
    timeout 400 rocprofv3 --pmc $c -d /tmp/pmc_$n -- $BENCH $PMCARGS > /tmp/pmc_$n.log 2>&1
  done
  python $R/scripts/pmc_summary.py "$TAG PMC ($W): rocprofv3 --pmc <counters> -- python bench.py --workload $W --steps 6 --warmup 3 --load-state <the default developed state> --no-cpu-baseline (FETCH_SIZE, WRITE_SIZE and SQ counters in three separate passes)" /tmp/pmc_*/*/*.db > $OUT/${TAG}_pmc$SUF.md
  printf "\nkernel sources: %s\n" "$FP" >> $OUT/${TAG}_pmc$SUF.md
  rm -rf /tmp/prof
  TRARGS="--steps 30 --warmup 6 --load-state $STATE --no-cpu-baseline --binding device"
  timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof -- $BENCH $TRARGS > /tmp/prof.log 2>&1
  grep -h "^{" /tmp/prof.log | head -1 > $OUT/${TAG}_bench_profiled$SUF.json
  python $R/scripts/rocpd_summary.py /tmp/prof/*/*.db "$TAG kernel trace ($W): rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --steps 30 --warmup 6 --load-state <the default developed state> --no-cpu-baseline" > $OUT/${TAG}_kernel_trace$SUF.md
  printf "\nkernel sources: %s\n" "$FP" >> $OUT/${TAG}_kernel_trace$SUF.md
  head -14 $OUT/${TAG}_kernel_trace$SUF.md
done
cd $R
case " $WORKLOADS " in *" step2d "*)
  timeout 500 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
  tail -c 600 $OUT/${TAG}_bench.json;;
esac

#!/bin/bash
# Latency / stall counters of the sweeps (one rocprofv3 --pmc pass per counter group, no trace domains):
#   gpurun -- 'bash scripts/pmc_diagnose.sh <tag> <workload>'  ->  gpurun_out/<tag>_pmc_diagnose_<workload>.md
# VmemLatency = accumulate(SQ_INST_LEVEL_VMEM) / SQ_INSTS_VMEM (cycles from issue to return, averaged over the vector
# memory instructions of the kernel), InstrFetchLatency likewise for instruction fetches, MemUnitStalled (% of cycles the
# memory unit is stalled), L2 hits and misses, write-request stalls at the fabric interface, occupancy.
set -u
TAG=$1; W=$2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
BENCH="python $R/bench.py --workload $W"
STATE=/tmp/state_$W.npz
[ -f $STATE ] || timeout 900 $BENCH --save-state $STATE --steps 3 --warmup 0 --reps 1 --no-cpu-baseline --binding device > /tmp/state.log 2>&1
rm -rf /tmp/pmcd_*
k=0
for c in "VmemLatency" "InstrFetchLatency" "LdsLatency" "MemUnitStalled" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum" "MeanOccupancyPerActiveCU" "VALUBusy" "TCP_PENDING_STALL_CYCLES_sum"; do
  k=$((k+1))
  timeout 400 rocprofv3 --pmc $c -d /tmp/pmcd_$k -- $BENCH --load-state $STATE --steps 6 --warmup 3 --reps 1 --no-cpu-baseline --binding device > /tmp/pmcd_$k.log 2>&1
done
python $R/scripts/pmc_summary.py "$TAG latency / stall counters ($W): one rocprofv3 --pmc pass per group, bench.py --workload $W --steps 6 --warmup 3 --load-state <developed state>" /tmp/pmcd_*/*/*.db > $OUT/${TAG}_pmc_diagnose_$W.md
cat $OUT/${TAG}_pmc_diagnose_$W.md | cut -c1-260 | head -16

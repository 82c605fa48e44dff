#!/bin/bash
# Vector-memory instructions and the compiler's s_waitcnt vmcnt(..) of a kernel, from the device assembly:
#   scripts/isa_loop_waits.sh <demangled-name substring> [-D...]
# gfx9 has one in-order counter for vector loads AND stores: a `vmcnt(0)` inside a column loop means the wave waits for
# everything it has in flight (the stores of the previous column included) before it goes on.
set -u
PAT=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
S=/tmp/ryujin_dev.s
if [ ! -f $S ] || [ -n "$(find $R/ryujin_amd/csrc -newer $S | head -1)" ] || [ $# -gt 0 ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S "$@" -I$R/include -I$R/ryujin_amd/csrc $R/ryujin_amd/csrc/ryujin_hip.hip -o $S 2>/dev/null
fi
grep -n "^_Z.*:  *; @" $S | while IFS=: read L NAME REST; do
  D=$(echo $NAME | c++filt | sed 's/ryujin_hip:://g')
  case "$D" in *"$PAT"*)
    E=$(awk -v s=$L 'NR>s && /^.Lfunc_end/{print NR; exit}' $S)
    echo "== ${D%%(*}  (lines $L-$E)"
    awk -v s=$L -v e=$E 'NR>=s && NR<=e' $S | grep -n "s_waitcnt vmcnt\|global_store\|global_load\|=>This\|in Loop: Header\|s_swappc\|scratch_" | awk '{printf "%s %s %s %s\n", $1, $2, $3, $4}'
  ;; esac
done

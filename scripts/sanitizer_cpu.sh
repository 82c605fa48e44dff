#!/bin/bash
# Sanitizer evidence (the reference builds its Debug configuration with
# -fsanitize=address,undefined,leak, CMakeLists.txt:76,135-142).
#
#   scripts/sanitizer_cpu.sh       g++ -fsanitize=address,undefined builds of libryujin_synth.so and of the oracle; the CPU
#                                  test-suite runs on them (python itself is not instrumented: libasan is preloaded)
# The host side of libryujin_hip.so only runs next to a GPU, and the GPU pool refuses every AddressSanitizer build
# (this file is in .gpurunignore for that reason): scripts/sanitizer_gpu_host.sh runs it under the
# UndefinedBehaviorSanitizer alone.
# Log: gpurun_out/r06_sanitizer_cpu.log (copy to profiles/).
set -u
MODE=cpu
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
D=$R/ryujin_amd/lib/sanitized_$MODE; mkdir -p $D
LOG=$OUT/r06_sanitizer_$MODE.log
CSRC=$R/ryujin_amd/csrc
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
{
if [ $MODE = cpu ]; then
  CXX=g++
  # (libstdc++ with it: python does not link it, and the interceptor of __cxa_throw must find the real one at start-up)
  RT="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)"
else
  CXX=/opt/rocm/lib/llvm/bin/amdclang++
  RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1) $(gcc -print-file-name=libstdc++.so)"
fi
echo "# sanitizer run ($MODE): $CXX $SAN; runtime $RT"
if [ -z "${RUN_ONLY:-}" ]; then
set -x
$CXX $SAN -std=c++17 -fPIC -shared -fopenmp -Wall -I$R/include -I$CSRC $CSRC/offline_synthetic.cc $CSRC/offline_io.cc -o $D/libryujin_synth.so || exit 1
$CXX $SAN -std=c++17 -fPIC -shared -fopenmp -ffp-contract=off -Wall -I$R/include -I$R/oracle $R/oracle/oracle_capi.cc -o $D/libryujin_oracle.so || exit 1
if [ $MODE = gpu ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -std=c++17 -fPIC -shared -ffp-contract=off -I$R/include -I$CSRC $CSRC/ryujin_hip.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $D/libryujin_hip.so || exit 1
fi
set +x
fi
[ -n "${BUILD_ONLY:-}" ] && exit 0
[ $MODE = gpu ] && export RYUJIN_HIP_LIB=$D/libryujin_hip.so
export RYUJIN_SYNTH_LIB=$D/libryujin_synth.so RYUJIN_ORACLE_LIB=$D/libryujin_oracle.so
# leaks: python and the HIP runtime are not ours to check; everything else aborts the test at the first report
rm -f $OUT/sanitizer_report_$MODE.*
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:verify_asan_link_order=0:log_path=$OUT/sanitizer_report_$MODE
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$OUT/sanitizer_report_$MODE
cd $R
if [ $MODE = cpu ]; then
  LD_PRELOAD="$RT" python -m pytest tests -q -m "not gpu" -p no:cacheprovider -x \
     --deselect tests/test_binding_run.py --deselect tests/test_binding_compile.py 2>&1 | tail -15
else
  LD_PRELOAD="$RT" python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x \
     -k "2d_step_geometry or partitioned or time_step or tile_map" 2>&1 | tail -25
fi
for f in $OUT/sanitizer_report_$MODE.*; do [ -f "$f" ] && { echo "## $f"; head -60 "$f"; }; done
echo "# sanitizer report files: $(ls $OUT/sanitizer_report_$MODE.* 2>/dev/null | wc -l)"
} > $LOG 2>&1
tail -6 $LOG

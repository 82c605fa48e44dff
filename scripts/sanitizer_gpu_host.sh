#!/bin/bash
# The HOST side of libryujin_hip.so (layout conversion, exchange lists, stream choreography, Runge-Kutta driver: 3.3 k
# lines that only run next to a GPU) under the UndefinedBehaviorSanitizer: hipcc -fsanitize=undefined -fno-gpu-sanitize
# (device code is not instrumented; the pool refuses AddressSanitizer builds, scripts/sanitizer_cpu.sh has that leg for
# the CPU libraries). Build where hipcc is (BUILD_ONLY=1), run on the GPU box (RUN_ONLY=1):
#   a selection of tests/test_gpu_parity.py + the partitioned and binding tests on the instrumented library.
# Log: gpurun_out/r06_sanitizer_gpu_host.log (copy to profiles/).
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out; mkdir -p $OUT
D=$R/ryujin_amd/lib/sanitized_gpu; mkdir -p $D
LOG=$OUT/r06_sanitizer_gpu_host.log
CSRC=$R/ryujin_amd/csrc
if [ -z "${RUN_ONLY:-}" ]; then
  set -x
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -g -fsanitize=undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer -std=c++17 -fPIC -shared -ffp-contract=off -I$R/include -I$CSRC $CSRC/ryujin_hip.hip -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -o $D/libryujin_hip.so || exit 1
  set +x
fi
[ -n "${BUILD_ONLY:-}" ] && exit 0
{
echo "# host side of libryujin_hip.so under -fsanitize=undefined (no recovery: the first report aborts the test process)"
export RYUJIN_HIP_LIB=$D/libryujin_hip.so
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:log_path=$OUT/sanitizer_report_gpu_host
rm -f $OUT/sanitizer_report_gpu_host.*
cd $R
# (a shared library does not carry the sanitizer runtime: preloaded into the uninstrumented python)
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1) $(gcc -print-file-name=libstdc++.so)"
LD_PRELOAD="$RT" python -m pytest tests/test_gpu_parity.py tests/test_partitioned_vs_oracle.py tests/test_binding_run.py -q -m gpu -p no:cacheprovider -x \
   -k "2d_step_geometry or partitioned or time_step or tile_map or large_meshes or host_mirroring or download_prepared" 2>&1 | tail -25
for f in $OUT/sanitizer_report_gpu_host.*; do [ -f "$f" ] && { echo "## $f"; head -60 "$f"; }; done
echo "# sanitizer report files: $(ls $OUT/sanitizer_report_gpu_host.* 2>/dev/null | wc -l)"
} > $LOG 2>&1
tail -8 $LOG

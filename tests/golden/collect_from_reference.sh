#!/bin/sh
# Generating script for tests/golden/: copies the reference's own golden OUTPUT
# data files (numdiff baselines: numbers only, no source) for the hot path.
# Run in the build container, where /root/reference exists. The files are data
# fixtures (expected outputs of the reference's tests); test INPUTS are restated
# in tests/*.py with a citation of the reference test they come from.
set -e
R=/root/reference/tests
D=$(dirname "$0")
cp $R/euler/riemann_solver.output              $D/euler_riemann_solver.output
cp $R/euler/riemann_solver-iterated-2.output   $D/euler_riemann_solver-iterated-2.output
cp $R/euler/riemann_solver-iterated-10.output  $D/euler_riemann_solver-iterated-10.output
cp $R/euler/riemann_solver-simd.output         $D/euler_riemann_solver-simd.output
cp $R/euler/limiter.output                     $D/euler_limiter.output
cp $R/euler/hyperbolic_system.output           $D/euler_hyperbolic_system.output
cp $R/euler/check-mass-conservation_01.output  $D/euler_check-mass-conservation_01.output
for l in 5 6; do
  cp $R/euler/verification-isentropic_vortex-2d-ssprk33-l$l.output $D/euler_verification-isentropic_vortex-2d-ssprk33-l$l.output
  cp $R/euler/verification-isentropic_vortex-2d-erk33-l$l.output   $D/euler_verification-isentropic_vortex-2d-erk33-l$l.output
done
cp $R/shallow_water/riemann_solver.output      $D/shallow_water_riemann_solver.output
cp $R/common/sparse_matrix_simd.output.sse2    $D/common_sparse_matrix_simd.output.sse2
cp $R/common/sparse_matrix_simd.output.avx2    $D/common_sparse_matrix_simd.output.avx2
cp $R/common/sparse_matrix_simd.output.avx512  $D/common_sparse_matrix_simd.output.avx512
cp "$R/common/sparsity_pattern_simd_01.mpirun=4.output" $D/common_sparsity_pattern_simd_01.mpirun4.output
# Euler with arbitrary equation of state
for f in riemann_solver riemann_solver-strict riemann_solver-strict-NASG limiter limiter-NASG \
         hyperbolic_system equation_of_state_library; do
  cp $R/euler_aeos/$f.output $D/euler_aeos_$f.output
done
for l in 5 6; do
  cp $R/euler_aeos/verification-isentropic_vortex-pge-2d-ssprk33-l$l.output $D/euler_aeos_verification-isentropic_vortex-pge-2d-ssprk33-l$l.output
  cp $R/euler_aeos/verification-isentropic_vortex-pge-2d-erk33-l$l.output   $D/euler_aeos_verification-isentropic_vortex-pge-2d-erk33-l$l.output
done
# scalar conservation
cp $R/scalar_conservation/riemann_solver.output     $D/scalar_conservation_riemann_solver.output
cp $R/scalar_conservation/hyperbolic_system.output  $D/scalar_conservation_hyperbolic_system.output
for s in ssprk22 ssprk33 erk11 erk22 erk33 erk43 erk54; do
  cp $R/scalar_conservation/verification-linear_transport-$s.output $D/scalar_conservation_verification-linear_transport-$s.output
done
# 1-D and shallow-water verification runs (analytic solutions; final time and error norms)
for f in leblanc-1d-erk33-l6 rarefaction-1d-erk33-l6; do
  cp "$R/euler/verification-$f.mpirun=4.output" $D/euler_verification-$f.mpirun4.output
done
for f in leblanc-pge-1d-erk33-l6 leblanc-pge-1d-erk33-l6-strict rarefaction-pge-1d-erk33-l6; do
  cp "$R/euler_aeos/verification-$f.mpirun=4.output" $D/euler_aeos_verification-$f.mpirun4.output
done
for f in paraboloid_1d-erk33-l7 ritter_dam_break-erk33-l7 smooth_vortex-erk33-l6 steady_incline-erk33-l9; do
  cp $R/shallow_water/verification-$f.output $D/shallow_water_verification-$f.output
done
for s in ssprk33 erk33; do
  cp "$R/euler/verification-isentropic_vortex-2d-$s-l7.mpirun=4.output" $D/euler_verification-isentropic_vortex-2d-$s-l7.mpirun4.output
done
for s in ssprk33 erk33; do
  cp "$R/euler_aeos/verification-isentropic_vortex-pge-2d-$s-l7.mpirun=4.output" $D/euler_aeos_verification-isentropic_vortex-pge-2d-$s-l7.mpirun4.output
done
# the reference's second baseline of the wetting/drying run: its own platform spread
cp $R/shallow_water/verification-paraboloid_1d-erk33-l7.output.gcc-13.3-avx2 $D/shallow_water_verification-paraboloid_1d-erk33-l7.output.gcc-13.3-avx2

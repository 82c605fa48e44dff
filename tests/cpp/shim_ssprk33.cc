// Drives the C++ shim the way ryujin's TimeIntegrator::step_ssprk_33 drives HyperbolicModule
// (source/time_integrator.template.h:302-328) on the check-mass-conservation_01 configuration and
// prints "t mean_rho" after every step. Built and run by tests/test_shim_cpp.py.
#include <array>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hyperbolic_module_shim.hpp"
#include "ryujin_synth.h"

using namespace ryujin_hip_shim;

int main(int argc, char **argv)
{
  const int n_steps = argc > 1 ? std::atoi(argv[1]) : 3;
  const bool device_resident = argc > 2 && std::atoi(argv[2]) != 0; /* 1: HyperbolicModule::time_step */
  ryujin_synth_spec spec{};
  spec.dim = 2;
  spec.n_cells[0] = spec.n_cells[1] = 64;
  spec.n_cells[2] = 1;
  spec.upper[0] = spec.upper[1] = 20.;
  for (int f = 0; f < 4; ++f)
    spec.bc[f] = RYUJIN_BC_SLIP;
  spec.n_ranks = 1;
  ryujin_synth *mesh = ryujin_synth_build(&spec);
  if (!mesh) {
    std::fprintf(stderr, "%s\n", ryujin_synth_last_error());
    return 1;
  }
  const ryujin_hip_offline *off = ryujin_synth_offline(mesh);

  ryujin_hip_params params;
  ryujin_hip_default_params(&params, RYUJIN_EQ_EULER, 2);

  try {
    HyperbolicModule hyperbolic_module(*off, params);
    hyperbolic_module.prepare();
    hyperbolic_module.cfl(0.9); /* TimeIntegrator::prepare(): cfl(cfl_max_) */

    /* uniform Mach-3 state rho = 1.4, u = 3, p = 1 (initial_state_uniform.h:36-38) */
    const unsigned n = off->n_relevant;
    std::vector<double> U(4 * (size_t)n);
    for (unsigned i = 0; i < n; ++i) {
      U[4 * i + 0] = 1.4;
      U[4 * i + 1] = 1.4 * 3.;
      U[4 * i + 2] = 0.;
      U[4 * i + 3] = 1. / 0.4 + 0.5 * 1.4 * 9.;
    }
    StateVector state_vector = hyperbolic_module.create_state_vector();
    std::array<StateVector, 3> temp_ = {hyperbolic_module.create_state_vector(),
                                        hyperbolic_module.create_state_vector(),
                                        hyperbolic_module.create_state_vector()};
    hyperbolic_module.upload(state_vector, U.data());

    double t = 0.;
    for (int cycle = 0; cycle < n_steps; ++cycle) {
      if (device_resident) {
        /* the whole Runge-Kutta step inside the library: one host synchronisation */
        t += hyperbolic_module.time_step(RYUJIN_SCHEME_SSPRK_33, state_vector, temp_, t);
      } else {
        /* step_ssprk_33 */
        hyperbolic_module.prepare_state_vector(state_vector, t);
        const double tau = hyperbolic_module.step<0>(state_vector, {}, {}, temp_[0], 0.);
        hyperbolic_module.prepare_state_vector(temp_[0], t + 1.0 * tau);
        hyperbolic_module.step<0>(temp_[0], {}, {}, temp_[1], tau);
        sadd(hyperbolic_module, temp_[1], 1.0 / 4.0, 3.0 / 4.0, state_vector);
        hyperbolic_module.prepare_state_vector(temp_[1], t + 0.5 * tau);
        hyperbolic_module.step<0>(temp_[1], {}, {}, temp_[0], tau);
        sadd(hyperbolic_module, temp_[0], 2.0 / 3.0, 1.0 / 3.0, state_vector);
        state_vector.swap(temp_[0]);
        t += tau;
      }

      hyperbolic_module.download(state_vector, U.data());
      double mass = 0., rho = 0.;
      for (unsigned i = 0; i < off->n_owned; ++i) {
        mass += off->mi[i];
        rho += off->mi[i] * U[4 * i];
      }
      std::printf("%.14e %.14e\n", t, rho / mass);
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    ryujin_synth_free(mesh);
    return 2;
  }
  ryujin_synth_free(mesh);
  return 0;
}

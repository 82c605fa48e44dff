// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// Shallow-water restatement: placeholder until the Euler path is green.
#pragma once
#include <array>
#include <stdexcept>
#include "hyperbolic_module.hpp"
namespace oracle
{
  namespace shallow_water
  {
    struct RiemannSolver {
      explicit RiemannSolver(const ryujin_hip_params &) {}
      double compute(const std::array<double, 3> &, const std::array<double, 3> &, double *) const
      {
        throw std::runtime_error("shallow water oracle not implemented yet");
      }
    };
  } // namespace shallow_water
  template <int dim>
  struct ShallowWaterModule final : ModuleBase {
    ShallowWaterModule(const ryujin_hip_offline &, const ryujin_hip_params &)
    {
      throw std::runtime_error("shallow water oracle not implemented yet");
    }
    int k() const override { return dim + 1; }
    int n_prec() const override { return 2; }
    int n_bounds() const override { return 5; }
    int state_alloc() override { return -1; }
    void state_free(int) override {}
    double *state_U(int) override { return nullptr; }
    double *state_prec(int) override { return nullptr; }
    void prepare_state_vector(int, double, const double *) override {}
    int step(int, int, const int *, const double *, int, double, double, double *) override { return RYUJIN_ERR_UNSUPPORTED; }
    void sadd(int, double, double, int) override {}
    int debug_fetch(int, double *, size_t) override { return RYUJIN_ERR_UNSUPPORTED; }
  };
} // namespace oracle

// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the scalar conservation Description (source/scalar_conservation/):
// flux library (burgers, kpp, and the "function" flux restricted to polynomials in u, with the
// central-difference gradient of dealii::FunctionParser / AutoDerivativeFunction), HyperbolicSystemView,
// RiemannSolver, Indicator, Limiter. Scalar double.
//
// Parity status: function level PINNED by tests/golden/scalar_conservation_{riemann_solver,
// hyperbolic_system}.output; whole-run level PINNED by the seven linear-transport verification outputs
// (tests/scalar_conservation/verification-linear_transport-{ssprk22,ssprk33,erk11,erk22,erk33,erk43,erk54}),
// which also pin every explicit Runge-Kutta scheme of the time integrator.
// Not restated: "random entropies" > 0 (std::random_device in the reference: not reproducible),
// arbitrary muparser expressions.

#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <tuple>

#include "ryujin_hip.h"

namespace oracle
{
  namespace scalar
  {
    /* FluxLibrary: source/scalar_conservation/flux_{burgers,kpp,function}.h */
    struct Flux {
      int kind = RYUJIN_FLUX_BURGERS;
      double poly[3][4] = {};
      double delta = 1.e4 * std::numeric_limits<double>::epsilon(); /* flux.h:33-34 */

      Flux() = default;
      explicit Flux(const ryujin_hip_params &p)
          : kind(p.sc_flux)
      {
        for (int d = 0; d < 3; ++d)
          for (int n = 0; n < 4; ++n)
            poly[d][n] = p.sc_flux_polynomial[d][n];
        if (kind == RYUJIN_FLUX_POLYNOMIAL)
          delta = p.sc_derivative_approximation_delta; /* flux_function.h:40-44 */
        if (kind < RYUJIN_FLUX_BURGERS || kind > RYUJIN_FLUX_POLYNOMIAL)
          throw std::runtime_error("unknown flux");
      }

      double polynomial(const double u, const int direction) const
      {
        const double *c = poly[direction];
        return c[0] + u * (c[1] + u * (c[2] + u * c[3]));
      }

      double value(const double u, const int direction) const
      {
        switch (kind) {
        case RYUJIN_FLUX_BURGERS: /* flux_burgers.h:35-39 */
          return 0.5 * u * u;
        case RYUJIN_FLUX_KPP: /* flux_kpp.h:35-48 */
          if (direction == 0)
            return std::sin(u);
          if (direction == 1)
            return std::cos(u);
          throw std::runtime_error("KPP is only defined in (1 or) 2 space dimensions");
        default:
          return polynomial(u, direction);
        }
      }

      double gradient(const double u, const int direction) const
      {
        switch (kind) {
        case RYUJIN_FLUX_BURGERS:
          return u;
        case RYUJIN_FLUX_KPP:
          if (direction == 0)
            return std::cos(u);
          if (direction == 1)
            return -std::sin(u);
          throw std::runtime_error("KPP is only defined in (1 or) 2 space dimensions");
        default:
          /* dealii::FunctionParser -> AutoDerivativeFunction, formula Euler:
           * (f(u + h) - f(u - h)) / (2 h)  (flux_function.h:80-84) */
          return (polynomial(u + delta, direction) - polynomial(u - delta, direction)) / (2 * delta);
        }
      }
    };


    /* HyperbolicSystemView: source/scalar_conservation/hyperbolic_system.h */
    template <int dim>
    struct View {
      static constexpr int problem_dimension = 1;
      static constexpr int n_precomputed_values = 2 * dim; /* f[dim], df[dim]  (:130-150) */
      using state_type = std::array<double, 1>;
      using vec_type = std::array<double, dim>;
      using precomputed_type = std::array<double, n_precomputed_values>;

      Flux flux;
      explicit View(const ryujin_hip_params &p)
          : flux(p)
      {
        if (flux.kind == RYUJIN_FLUX_KPP && dim == 3)
          throw std::runtime_error("KPP is only defined in (1 or) 2 space dimensions");
      }

      double derivative_approximation_delta() const { return flux.delta; }

      vec_type flux_function(const double u) const
      {
        vec_type r;
        for (int k = 0; k < dim; ++k)
          r[k] = flux.value(u, k);
        return r;
      }
      vec_type flux_gradient_function(const double u) const
      {
        vec_type r;
        for (int k = 0; k < dim; ++k)
          r[k] = flux.gradient(u, k);
        return r;
      }

      /* precomputation_loop body (:283-300) */
      precomputed_type precompute(const double u) const
      {
        const auto f = flux_function(u);
        const auto df = flux_gradient_function(u);
        precomputed_type p;
        for (int k = 0; k < dim; ++k) {
          p[k] = f[k];
          p[dim + k] = df[k];
        }
        return p;
      }

      static vec_type construct_flux_tensor(const precomputed_type &p)
      {
        vec_type r;
        for (int k = 0; k < dim; ++k)
          r[k] = p[k];
        return r;
      }
      static vec_type construct_flux_gradient_tensor(const precomputed_type &p)
      {
        vec_type r;
        for (int k = 0; k < dim; ++k)
          r[k] = p[dim + k];
        return r;
      }

      static double square_entropy(const double u) { return 0.5 * u * u; }
      static double square_entropy_derivative(const double u) { return u; }
      static double kruzkov_entropy(const double k, const double u) { return std::abs(k - u); }
      static double kruzkov_entropy_derivative(const double k, const double u)
      {
        return u >= k ? 1. : -1.;
      }

      /* apply_boundary_conditions (:381-420): Dirichlet only; slip / no_slip / dynamic throw */
      state_type apply_boundary_conditions(const int id, const state_type &U,
                                           const state_type &U_dirichlet) const
      {
        if (id == RYUJIN_BC_DIRICHLET)
          return U_dirichlet;
        if (id == RYUJIN_BC_SLIP || id == RYUJIN_BC_NO_SLIP || id == RYUJIN_BC_DYNAMIC)
          throw std::runtime_error("boundary conditions other than dirichlet / do nothing / periodic are "
                                   "unavailable for scalar conservation equations");
        return U;
      }

      static double dot(const vec_type &a, const vec_type &b)
      {
        double s = 0.;
        for (int k = 0; k < dim; ++k)
          s += a[k] * b[k];
        return s;
      }
    };


    /* RiemannSolver: source/scalar_conservation/riemann_solver.template.h:21-175 */
    struct RiemannTrace {
      double f_i, f_j, df_i, df_j, roe, second, third, k, f_k, left, right;
    };

    template <int dim>
    struct RiemannSolver {
      using V = View<dim>;
      const V &view;
      bool use_greedy_wavespeed, use_averaged_entropy;
      mutable RiemannTrace *trace = nullptr;

      RiemannSolver(const V &view, const ryujin_hip_params &p)
          : view(view)
          , use_greedy_wavespeed(p.sc_use_greedy_wavespeed != 0)
          , use_averaged_entropy(p.sc_use_averaged_entropy != 0)
      {
        if (p.sc_random_entropies != 0)
          throw std::runtime_error("random entropies > 0 draw from std::random_device in the reference "
                                   "and are not reproducible: unsupported");
      }

      double compute(const double u_i, const double u_j, const typename V::precomputed_type &prec_i,
                     const typename V::precomputed_type &prec_j, const typename V::vec_type &n_ij) const
      {
        const double f_i = V::dot(V::construct_flux_tensor(prec_i), n_ij);
        const double f_j = V::dot(V::construct_flux_tensor(prec_j), n_ij);
        const double df_i = V::dot(V::construct_flux_gradient_tensor(prec_i), n_ij);
        const double df_j = V::dot(V::construct_flux_gradient_tensor(prec_j), n_ij);
        const double h2 = 2. * view.derivative_approximation_delta();

        double lambda_max = std::abs(f_i - f_j) / std::max(std::abs(u_i - u_j), h2);
        if (trace) {
          *trace = RiemannTrace{};
          trace->f_i = f_i, trace->f_j = f_j, trace->df_i = df_i, trace->df_j = df_j;
          trace->roe = lambda_max;
        }

        if (use_greedy_wavespeed) {
          const double interpolated = std::abs(0.5 * (df_i + df_j));
          lambda_max = std::abs(u_i - u_j) >= h2 ? lambda_max : interpolated;
          if (trace)
            trace->second = interpolated;
        } else {
          lambda_max = std::max(lambda_max, std::abs(df_i));
          lambda_max = std::max(lambda_max, std::abs(df_j));
          if (trace)
            trace->second = std::abs(df_i), trace->third = std::abs(df_j);
        }

        if (use_averaged_entropy) {
          const double k = 0.5 * (u_i + u_j);
          const double f_k = V::dot(view.flux_function(k), n_ij);
          const double eta_i = V::kruzkov_entropy(k, u_i);
          const double q_i = V::kruzkov_entropy_derivative(k, u_i) * (f_i - f_k);
          const double eta_j = V::kruzkov_entropy(k, u_j);
          const double q_j = V::kruzkov_entropy_derivative(k, u_j) * (f_j - f_k);
          const double a = u_i + u_j - 2. * k;
          const double b = f_j - f_i;
          const double c = eta_i + eta_j;
          const double d = q_j - q_i;
          const double lambda_left = std::abs(d + b) / (std::abs(c + a) + h2);
          const double lambda_right = std::abs(d - b) / (std::abs(c - a) + h2);
          lambda_max = std::max(lambda_max, lambda_left);
          lambda_max = std::max(lambda_max, lambda_right);
          if (trace)
            trace->k = k, trace->f_k = f_k, trace->left = lambda_left, trace->right = lambda_right;
        }
        return lambda_max;
      }
    };


    /* Indicator: source/scalar_conservation/indicator.h:160-205 */
    template <int dim>
    struct Indicator {
      using V = View<dim>;
      double evc_factor;
      double u_i = 0., u_abs_max = 0., left = 0., right = 0.;
      typename V::vec_type f_i;

      explicit Indicator(const ryujin_hip_params &p)
          : evc_factor(p.indicator_evc_factor)
      {
      }

      void reset(const double new_u_i, const typename V::precomputed_type &prec_i)
      {
        u_i = new_u_i;
        u_abs_max = std::abs(u_i);
        f_i = V::construct_flux_tensor(prec_i);
        left = 0.;
        right = 0.;
      }

      void accumulate(const double u_j, const typename V::precomputed_type &prec_j,
                      const typename V::vec_type &c_ij)
      {
        u_abs_max = std::max(u_abs_max, std::abs(u_j));
        const double d_eta_j = V::kruzkov_entropy_derivative(u_i, u_j);
        const auto f_j = V::construct_flux_tensor(prec_j);
        left += d_eta_j * V::dot(f_j, c_ij);
        right += d_eta_j * V::dot(f_i, c_ij);
      }

      double alpha(const double hd_i) const
      {
        const double numerator = left - right;
        const double denominator = std::abs(left) + std::abs(right);
        const double regularization = 100. * std::numeric_limits<double>::min();
        const double quotient =
            std::abs(numerator) / (denominator + std::max(hd_i * std::abs(u_abs_max), regularization));
        return std::min(1., evc_factor * quotient);
      }
    };


    /* Limiter: source/scalar_conservation/limiter.h:190-290, limiter.template.h:15-110 */
    template <int dim>
    struct Limiter {
      using V = View<dim>;
      static constexpr int n_bounds = 2;
      using Bounds = std::array<double, 2>;
      double relaxation_factor;
      bool expensive_bounds_check = false;

      double u_i = 0.;
      typename V::vec_type flux_i;
      Bounds bounds_;
      double u_relaxation_numerator = 0., u_relaxation_denominator = 0.;

      explicit Limiter(const ryujin_hip_params &p)
          : relaxation_factor(p.limiter_relaxation_factor)
      {
      }

      void reset(const double new_u_i, const typename V::vec_type &new_flux_i)
      {
        u_i = new_u_i;
        flux_i = new_flux_i;
        bounds_[0] = std::numeric_limits<double>::max();
        bounds_[1] = std::numeric_limits<double>::lowest();
        u_relaxation_numerator = 0.;
        u_relaxation_denominator = 0.;
      }

      void accumulate(const double u_j, const typename V::vec_type &flux_j,
                      const typename V::vec_type &scaled_c_ij)
      {
        double contracted = 0.;
        for (int d = 0; d < dim; ++d)
          contracted += (flux_j[d] + (-flux_i[d])) * scaled_c_ij[d];
        const double u_ij_bar = 0.5 * (u_i + u_j) - 0.5 * contracted + 0.;
        bounds_[0] = std::min(bounds_[0], u_ij_bar);
        bounds_[1] = std::max(bounds_[1], u_ij_bar);
        const double beta_ij = 1.;
        u_relaxation_numerator += beta_ij * (u_i + u_j);
        u_relaxation_denominator += std::abs(beta_ij);
      }

      Bounds bounds(const double hd_i) const
      {
        auto [u_min, u_max] = bounds_;
        double r_i = std::sqrt(hd_i);
        if constexpr (dim == 2) {
          const double t = std::sqrt(r_i);
          r_i = t * t * t;
        } else if constexpr (dim == 1) {
          r_i = r_i * r_i * r_i;
        }
        r_i *= relaxation_factor;
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double u_relaxation =
            std::abs(u_relaxation_numerator) / (std::abs(u_relaxation_denominator) + eps);
        u_min = std::max(std::min((1. - r_i) * u_min, (1. + r_i) * u_min), u_min - 2. * u_relaxation);
        u_max = std::min(std::max((1. + r_i) * u_max, (1. - r_i) * u_max), u_max + 2. * u_relaxation);
        return {u_min, u_max};
      }

      std::tuple<double, bool> limit(const Bounds &bounds, const double u_U, const double u_P,
                                     const double t_min = 0., const double t_max = 1.) const
      {
        bool success = true;
        double t_r = t_max;
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double relax = 1. + 10000. * eps;
        const double u_min = bounds[0], u_max = bounds[1];

        const double test_max = std::max(0., std::min(u_U - relax * u_max, relax * u_U - u_max));
        const double test_min = std::max(0., std::min(u_min - relax * u_U, relax * u_min - u_U));
        if (!(test_max == 0. && test_min == 0.))
          success = false;

        const double regularization = 100. * std::numeric_limits<double>::min();
        const double denominator = 1. / std::max(regularization, std::abs(u_P) + eps * u_max);
        t_r = u_max < u_U + t_r * u_P ? (u_max - u_U) * denominator : t_r;
        t_r = u_U + t_r * u_P < u_min ? (u_U - u_min) * denominator : t_r;
        t_r = std::min(t_r, t_max);
        t_r = std::max(t_r, t_min);

        if (expensive_bounds_check) {
          const double u_new = u_U + t_r * u_P;
          const double test_new_max =
              std::max(0., std::min(u_new - relax * u_max, relax * u_new - u_max));
          const double test_new_min =
              std::max(0., std::min(u_min - relax * u_new, relax * u_min - u_new));
          if (!(test_new_max == 0. && test_new_min == 0.))
            success = false;
        }
        return {t_r, success};
      }
    };
  } // namespace scalar
} // namespace oracle

// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement (scalar double) of ryujin's Euler "Description":
// HyperbolicSystemView, RiemannSolver, Indicator and Limiter. Every function
// cites the reference file:line it restates. Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Parity status: PINNED against the reference's own golden outputs
// (tests/golden/euler_*.output; see tests/test_oracle_golden.py).
// ryujin::pow is std::pow here (source/simd.template.h:233-272, the non-x86
// branch); x86 reference builds use vcl::pow and differ in the last digits
// (SURVEY.md Appendix E-1).

#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <limits>
#include <vector>

#include "ryujin_hip.h"

namespace oracle
{
  inline double positive_part(double x) { return std::max(0., x); }
  inline double negative_part(double x) { return -std::min(0., x); } /* simd.h: (|x|-x)/2 */

  /* source/newton.h:37-101 */
  inline void quadratic_newton_step(double &p_1, double &p_2, const double phi_p_1,
                                    const double phi_p_2, const double dphi_p_1,
                                    const double dphi_p_2, const double sign = 1.0)
  {
    constexpr double eps = std::numeric_limits<double>::epsilon();
    const double scaling = 1. / (p_2 - p_1 + eps);

    const double dd_11 = dphi_p_1;
    const double dd_12 = (phi_p_2 - phi_p_1) * scaling;
    const double dd_22 = dphi_p_2;

    const double dd_112 = (dd_12 - dd_11) * scaling;
    const double dd_122 = (dd_22 - dd_12) * scaling;

    const double discriminant_1 = std::abs(dphi_p_1 * dphi_p_1 - 4. * phi_p_1 * dd_112);
    const double discriminant_2 = std::abs(dphi_p_2 * dphi_p_2 - 4. * phi_p_2 * dd_122);

    const double denominator_1 = dphi_p_1 + sign * std::sqrt(discriminant_1);
    const double denominator_2 = dphi_p_2 + sign * std::sqrt(discriminant_2);

    double t_1 = p_1 - (std::abs(denominator_1) < eps ? 0. : 2. * phi_p_1 / denominator_1);
    double t_2 = p_2 - (std::abs(denominator_2) < eps ? 0. : 2. * phi_p_2 / denominator_2);

    t_1 = std::max(p_1, t_1);
    t_1 = std::min(p_2, t_1);
    t_2 = std::max(p_1, t_2);
    t_2 = std::min(p_2, t_2);

    p_1 = std::min(t_1, t_2);
    p_2 = std::max(t_1, t_2);
  }

  namespace euler
  {
    template <int dim>
    struct View {
      static constexpr int k = dim + 2;
      using state_type = std::array<double, k>;
      using flux_type = std::array<std::array<double, dim>, k>;

      double gamma, gamma_inverse, gamma_plus_one_inverse, gamma_minus_one_inverse;
      double reference_density, vacuum_state_relaxation_small, vacuum_state_relaxation_large;

      explicit View(const ryujin_hip_params &p)
      {
        gamma = p.gamma;
        /* source/euler/hyperbolic_system.h:690-695 */
        gamma_inverse = 1. / gamma;
        gamma_plus_one_inverse = 1. / (gamma + 1.);
        gamma_minus_one_inverse = 1. / (gamma - 1.);
        reference_density = p.reference_density;
        vacuum_state_relaxation_small = p.vacuum_state_relaxation_small;
        vacuum_state_relaxation_large = p.vacuum_state_relaxation_large;
      }

      static double density(const state_type &U) { return U[0]; }
      static double total_energy(const state_type &U) { return U[1 + dim]; }
      static double momentum_norm_square(const state_type &U)
      {
        double s = 0.;
        for (int d = 0; d < dim; ++d)
          s += U[1 + d] * U[1 + d];
        return s;
      }

      /* hyperbolic_system.h:750-759 */
      double filter_vacuum_density(const double rho) const
      {
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double rho_cutoff_large = reference_density * vacuum_state_relaxation_large * eps;
        return std::abs(rho) < rho_cutoff_large ? 0. : rho;
      }

      /* hyperbolic_system.h:783-792 */
      static double internal_energy(const state_type &U)
      {
        const double rho_inverse = 1. / density(U);
        return total_energy(U) - 0.5 * momentum_norm_square(U) * rho_inverse;
      }

      /* hyperbolic_system.h:797-819 */
      static state_type internal_energy_derivative(const state_type &U)
      {
        const double rho_inverse = 1. / density(U);
        state_type result;
        double u2 = 0.;
        for (int d = 0; d < dim; ++d) {
          const double u = U[1 + d] * rho_inverse;
          u2 += u * u;
          result[1 + d] = -u;
        }
        result[0] = 0.5 * u2;
        result[dim + 1] = 1.;
        return result;
      }

      /* hyperbolic_system.h:824-828 */
      double pressure(const state_type &U) const { return (gamma - 1.) * internal_energy(U); }

      /* hyperbolic_system.h:833-839 */
      double speed_of_sound(const state_type &U) const
      {
        const double rho_inverse = 1. / density(U);
        const double p = pressure(U);
        return std::sqrt(gamma * p * rho_inverse);
      }

      /* hyperbolic_system.h:844-850 */
      double specific_entropy(const state_type &U) const
      {
        const double rho_inverse = 1. / density(U);
        return internal_energy(U) * std::pow(rho_inverse, gamma);
      }

      /* hyperbolic_system.h:855-865 */
      double harten_entropy(const state_type &U) const
      {
        const double rho_rho_e = density(U) * total_energy(U) - 0.5 * momentum_norm_square(U);
        return std::pow(rho_rho_e, gamma_plus_one_inverse);
      }

      /* hyperbolic_system.h:870-902 */
      state_type harten_entropy_derivative(const state_type &U) const
      {
        const double rho = density(U);
        const double E = total_energy(U);
        const double rho_rho_e = rho * E - 0.5 * momentum_norm_square(U);
        const double factor =
            gamma_plus_one_inverse * std::pow(rho_rho_e, -gamma * gamma_plus_one_inverse);
        state_type result;
        result[0] = factor * E;
        for (int d = 0; d < dim; ++d)
          result[1 + d] = -factor * U[1 + d];
        result[dim + 1] = factor * rho;
        return result;
      }

      /* hyperbolic_system.h:907-913 */
      double mathematical_entropy(const state_type &U) const
      {
        return std::pow(pressure(U), gamma_inverse);
      }

      /* hyperbolic_system.h:918-950 */
      state_type mathematical_entropy_derivative(const state_type &U) const
      {
        const double rho_inverse = 1. / density(U);
        const double p = pressure(U);
        const double factor = (gamma - 1.0) * gamma_inverse * std::pow(p, gamma_inverse - 1.);
        state_type result;
        double u2 = 0.;
        for (int d = 0; d < dim; ++d) {
          const double u = U[1 + d] * rho_inverse;
          u2 += u * u;
          result[1 + d] = -factor * u;
        }
        result[0] = factor * 0.5 * u2;
        result[dim + 1] = factor;
        return result;
      }

      /* hyperbolic_system.h:955-979 */
      bool is_admissible(const state_type &U) const
      {
        return density(U) > 0. && internal_energy(U) > 0. && specific_entropy(U) > 0.;
      }

      /* hyperbolic_system.h:1164-1181 */
      flux_type f(const state_type &U) const
      {
        const double rho_inverse = 1. / density(U);
        const double p = pressure(U);
        const double E = total_energy(U);
        flux_type result;
        for (int d = 0; d < dim; ++d)
          result[0][d] = U[1 + d];
        for (int i = 0; i < dim; ++i) {
          const double s = U[1 + i] * rho_inverse;
          for (int d = 0; d < dim; ++d)
            result[1 + i][d] = U[1 + d] * s;
          result[1 + i][i] += p;
        }
        const double s = rho_inverse * (E + p);
        for (int d = 0; d < dim; ++d)
          result[dim + 1][d] = U[1 + d] * s;
        return result;
      }

      /* flux_divergence = -contract(add(flux_i, flux_j), c_ij)
       * hyperbolic_system.h:1208-1216, convenience_macros.h:79-102 */
      static state_type flux_divergence(const flux_type &flux_i, const flux_type &flux_j,
                                        const std::array<double, dim> &c_ij)
      {
        state_type result;
        for (int q = 0; q < k; ++q) {
          double s = 0.;
          for (int d = 0; d < dim; ++d)
            s += (flux_i[q][d] + flux_j[q][d]) * c_ij[d];
          result[q] = -s;
        }
        return result;
      }

      /* hyperbolic_system.h:1040-1093 */
      template <int component>
      state_type prescribe_riemann_characteristic(const state_type &U, const state_type &U_bar,
                                                  const std::array<double, dim> &normal) const
      {
        const double rho = density(U);
        const double a = speed_of_sound(U);
        double mn = 0.;
        for (int d = 0; d < dim; ++d)
          mn += U[1 + d] * normal[d];
        const double vn = mn / rho;

        const double rho_bar = density(U_bar);
        const double a_bar = speed_of_sound(U_bar);
        double mn_bar = 0.;
        for (int d = 0; d < dim; ++d)
          mn_bar += U_bar[1 + d] * normal[d];
        const double vn_bar = mn_bar / rho_bar;

        const double R_1 =
            component == 1 ? vn_bar - 2. * a_bar / (gamma - 1.) : vn - 2. * a / (gamma - 1.);
        const double R_2 =
            component == 2 ? vn_bar + 2. * a_bar / (gamma - 1.) : vn + 2. * a / (gamma - 1.);

        const double p = pressure(U);
        const double s = p / std::pow(rho, gamma);

        std::array<double, dim> vperp;
        for (int d = 0; d < dim; ++d)
          vperp[d] = U[1 + d] / rho - vn * normal[d];

        const double vn_new = 0.5 * (R_1 + R_2);

        const double tmp = ((gamma - 1.) / 4.) * (R_2 - R_1);
        double rho_new = 1. / (gamma * s) * (tmp * tmp);
        rho_new = std::pow(rho_new, 1. / (gamma - 1.));

        const double p_new = s * std::pow(rho_new, gamma);

        state_type U_new;
        U_new[0] = rho_new;
        double vperp2 = 0.;
        for (int d = 0; d < dim; ++d) {
          U_new[1 + d] = rho_new * (vn_new * normal[d] + vperp[d]);
          vperp2 += vperp[d] * vperp[d];
        }
        U_new[1 + dim] = p_new / (gamma - 1.) + 0.5 * rho_new * (vn_new * vn_new + vperp2);
        return U_new;
      }

      /* hyperbolic_system.h:1099-1159 */
      state_type apply_boundary_conditions(int id, const state_type &U,
                                           const std::array<double, dim> &normal,
                                           const state_type &U_dirichlet) const
      {
        state_type result = U;
        if (id == RYUJIN_BC_DIRICHLET) {
          result = U_dirichlet;
        } else if (id == RYUJIN_BC_SLIP) {
          double mn = 0.;
          for (int d = 0; d < dim; ++d)
            mn += U[1 + d] * normal[d];
          for (int d = 0; d < dim; ++d)
            result[1 + d] = U[1 + d] - 1. * mn * normal[d];
        } else if (id == RYUJIN_BC_NO_SLIP) {
          for (int d = 0; d < dim; ++d)
            result[1 + d] = 0.;
        } else if (id == RYUJIN_BC_DYNAMIC) {
          const double rho = density(U);
          const double a = speed_of_sound(U);
          double mn = 0.;
          for (int d = 0; d < dim; ++d)
            mn += U[1 + d] * normal[d];
          const double vn = mn / rho;
          if (vn < -a)
            result = U_dirichlet;
          if (vn >= -a && vn <= 0.)
            result = prescribe_riemann_characteristic<2>(U_dirichlet, U, normal);
          if (vn > 0. && vn <= a)
            result = prescribe_riemann_characteristic<1>(U, U_dirichlet, normal);
        }
        return result;
      }
    };


    /* ---------------------------------------------------------------------
     * RiemannSolver: source/euler/riemann_solver.template.h
     * riemann data = {rho, u, p, a}
     * ------------------------------------------------------------------- */

    struct RiemannTrace {
      double p_star_two_rarefaction = 0., p_star_failsafe = 0., p_star_tilde = 0.,
             phi_p_star_tilde = 0., lambda_max = 0.;
      int converged_after = -1; /* iteration index at which the tolerance break fired */
      double p_1_start = 0., p_2_start = 0., gap_start = 0., lambda_max_start = 0.;
      struct Iter {
        double phi_p_1, phi_p_2, dphi_p_1, dphi_p_2, p_1, p_2, gap, lambda_max;
      };
      std::vector<Iter> iterations;
    };

    using primitive_type = std::array<double, 4>;

    struct RiemannSolver {
      double gamma, gamma_inverse, gamma_minus_one_inverse, gamma_plus_one_inverse;
      unsigned int newton_max_iterations;
      double newton_tolerance;

      explicit RiemannSolver(const ryujin_hip_params &p)
      {
        gamma = p.gamma;
        gamma_inverse = 1. / gamma;
        gamma_minus_one_inverse = 1. / (gamma - 1.);
        gamma_plus_one_inverse = 1. / (gamma + 1.);
        newton_max_iterations = (unsigned)p.riemann_newton_max_iterations;
        newton_tolerance = p.riemann_newton_tolerance;
      }

      /* riemann_solver.template.h:21-46 */
      double f(const primitive_type &rd, const double p_star) const
      {
        const auto &[rho, u, p, a] = rd;
        (void)u;
        const double Az = 2. / (rho * (gamma + 1.));
        const double Bz = (gamma - 1.) / (gamma + 1.) * p;
        const double radicand = Az / (p_star + Bz);
        const double true_value = (p_star - p) * std::sqrt(radicand);

        const double exponent = 0.5 * (gamma - 1.) / gamma;
        const double factor = std::pow(p_star / p, exponent) - 1.;
        const double false_value = 2. * a * factor / (gamma - 1.);
        return p_star >= p ? true_value : false_value;
      }

      /* riemann_solver.template.h:49-84 */
      double df(const primitive_type &rd, const double p_star) const
      {
        const auto &[rho, u, p, a] = rd;
        (void)u;
        const double radicand_inverse = 0.5 * rho * ((gamma + 1.) * p_star + (gamma - 1.) * p);
        const double denominator = (p_star + (gamma - 1.) * gamma_plus_one_inverse * p);
        const double true_value =
            (denominator - 0.5 * (p_star - p)) / (denominator * std::sqrt(radicand_inverse));

        const double exponent = (-1. - gamma) * 0.5 * gamma_inverse;
        const double factor =
            (gamma - 1.) * 0.5 * gamma_inverse * std::pow(p_star / p, exponent) / p;
        const double false_value = factor * 2. * a * gamma_minus_one_inverse;
        return p_star >= p ? true_value : false_value;
      }

      /* :87-107 */
      double phi(const primitive_type &rd_i, const primitive_type &rd_j, const double p_in) const
      {
        return f(rd_i, p_in) + f(rd_j, p_in) + rd_j[1] - rd_i[1];
      }
      double dphi(const primitive_type &rd_i, const primitive_type &rd_j, const double p) const
      {
        return df(rd_i, p) + df(rd_j, p);
      }

      /* :122-149 */
      double phi_of_p_max(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, a_j] = rd_j;
        (void)a_i;
        (void)a_j;
        const double p_max = std::max(p_i, p_j);
        const double radicand_inverse_i =
            0.5 * rho_i * ((gamma + 1.) * p_max + (gamma - 1.) * p_i);
        const double value_i = (p_max - p_i) / std::sqrt(radicand_inverse_i);
        const double radicand_inverse_j =
            0.5 * rho_j * ((gamma + 1.) * p_max + (gamma - 1.) * p_j);
        const double value_j = (p_max - p_j) / std::sqrt(radicand_inverse_j);
        return value_i + value_j + u_j - u_i;
      }

      /* :164-181 */
      double lambda1_minus(const primitive_type &rd, const double p_star) const
      {
        const double factor = (gamma + 1.0) * 0.5 * gamma_inverse;
        const auto &[rho, u, p, a] = rd;
        (void)rho;
        const double inv_p = 1.0 / p;
        const double tmp = positive_part((p_star - p) * inv_p);
        return u - a * std::sqrt(1.0 + factor * tmp);
      }

      /* :189-205 */
      double lambda3_plus(const primitive_type &rd, const double p_star) const
      {
        const double factor = (gamma + 1.0) * 0.5 * gamma_inverse;
        const auto &[rho, u, p, a] = rd;
        (void)rho;
        const double inv_p = 1.0 / p;
        const double tmp = positive_part((p_star - p) * inv_p);
        return u + a * std::sqrt(1.0 + factor * tmp);
      }

      /* :217-238 */
      std::array<double, 2> compute_gap(const primitive_type &rd_i, const primitive_type &rd_j,
                                        const double p_1, const double p_2) const
      {
        const double nu_11 = lambda1_minus(rd_i, p_2 /*SIC!*/);
        const double nu_12 = lambda1_minus(rd_i, p_1 /*SIC!*/);
        const double nu_31 = lambda3_plus(rd_j, p_1);
        const double nu_32 = lambda3_plus(rd_j, p_2);
        const double lambda_max = std::max(positive_part(nu_32), negative_part(nu_11));
        const double gap = std::max(std::abs(nu_32 - nu_31), std::abs(nu_12 - nu_11));
        return {{gap, lambda_max}};
      }

      /* :252-263 */
      double compute_lambda(const primitive_type &rd_i, const primitive_type &rd_j,
                            const double p_star) const
      {
        const double nu_11 = lambda1_minus(rd_i, p_star);
        const double nu_32 = lambda3_plus(rd_j, p_star);
        return std::max(positive_part(nu_32), negative_part(nu_11));
      }

      /* :274-319 */
      double p_star_two_rarefaction(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, a_j] = rd_j;
        (void)rho_i;
        (void)rho_j;
        const double inv_p_j = 1. / p_j;
        const double factor = (gamma - 1.) * 0.5;
        const double numerator = positive_part(a_i + a_j - factor * (u_j - u_i));
        const double denominator = a_i * std::pow(p_i * inv_p_j, -factor * gamma_inverse) + a_j;
        const double exponent = 2.0 * gamma * gamma_minus_one_inverse;
        return p_j * std::pow(numerator / denominator, exponent);
      }

      /* :330-374 */
      double p_star_failsafe(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, a_j] = rd_j;
        (void)a_i;
        (void)a_j;
        const double p_max = std::max(p_i, p_j);

        double radicand_i = 2. * p_max;
        radicand_i /= rho_i * ((gamma + 1.) * p_max + (gamma - 1.) * p_i);
        const double x_i = std::sqrt(radicand_i);

        double radicand_j = 2. * p_max;
        radicand_j /= rho_j * ((gamma + 1.) * p_max + (gamma - 1.) * p_j);
        const double x_j = std::sqrt(radicand_j);

        const double a = x_i + x_j;
        const double b = u_j - u_i;
        const double c = -p_i * x_i - p_j * x_j;

        const double base = (-b + std::sqrt(b * b - 4. * a * c)) / (2. * a);
        return base * base;
      }

      /* :377-403 */
      template <int dim>
      primitive_type riemann_data_from_state(const std::array<double, dim + 2> &U,
                                             const std::array<double, dim> &n_ij) const
      {
        const double rho = U[0];
        const double rho_inverse = 1.0 / rho;
        double proj_m = 0.;
        for (int d = 0; d < dim; ++d)
          proj_m += n_ij[d] * U[1 + d];
        double perp2 = 0.;
        for (int d = 0; d < dim; ++d) {
          const double perp = U[1 + d] - proj_m * n_ij[d];
          perp2 += perp * perp;
        }
        const double E = U[1 + dim] - 0.5 * perp2 * rho_inverse;
        /* 1-D view: pressure, speed of sound of {rho, proj_m, E} */
        const double rho_e = E - 0.5 * (proj_m * proj_m) * (1. / rho);
        const double p = (gamma - 1.) * rho_e;
        const double a = std::sqrt(gamma * p * (1. / rho));
        return {{rho, proj_m * rho_inverse, p, a}};
      }

      /* :406-582 */
      double compute(const primitive_type &rd_i, const primitive_type &rd_j,
                     RiemannTrace *trace = nullptr) const
      {
        const double p_i = rd_i[2], p_j = rd_j[2];
        const double p_max = std::max(p_i, p_j);

        const double rarefaction = p_star_two_rarefaction(rd_i, rd_j);
        const double failsafe = p_star_failsafe(rd_i, rd_j);
        const double p_star_tilde = std::min(rarefaction, failsafe);

        const double phi_p_max = phi_of_p_max(rd_i, rd_j);

        double p_2 = phi_p_max < 0. ? p_star_tilde : std::min(p_max, p_star_tilde);

        if (trace) {
          trace->p_star_two_rarefaction = rarefaction;
          trace->p_star_failsafe = failsafe;
          trace->p_star_tilde = p_2;
          trace->phi_p_star_tilde = phi(rd_i, rd_j, p_2);
        }

        if (newton_max_iterations == 0) {
          const double lambda_max = compute_lambda(rd_i, rd_j, p_2);
          if (trace)
            trace->lambda_max = lambda_max;
          return lambda_max;
        }

        const double p_min = std::min(p_i, p_j);
        double p_1 = phi_p_max < 0. ? p_max : p_min;
        p_1 = p_1 <= p_2 ? p_1 : p_2;

        auto [gap, lambda_max] = compute_gap(rd_i, rd_j, p_1, p_2);
        if (trace) {
          trace->p_1_start = p_1;
          trace->p_2_start = p_2;
          trace->gap_start = gap;
          trace->lambda_max_start = lambda_max;
        }

        for (unsigned int i = 0; i < newton_max_iterations; ++i) {
          if (std::max(0., gap - newton_tolerance) == 0.) {
            if (trace)
              trace->converged_after = (int)i;
            break;
          }
          const double phi_p_1 = phi(rd_i, rd_j, p_1);
          const double phi_p_2 = phi(rd_i, rd_j, p_2);
          const double dphi_p_1 = dphi(rd_i, rd_j, p_1);
          const double dphi_p_2 = dphi(rd_i, rd_j, p_2);

          quadratic_newton_step(p_1, p_2, phi_p_1, phi_p_2, dphi_p_1, dphi_p_2);

          auto [gap_new, lambda_max_new] = compute_gap(rd_i, rd_j, p_1, p_2);
          gap = gap_new;
          lambda_max = lambda_max_new;
          if (trace)
            trace->iterations.push_back(
                {phi_p_1, phi_p_2, dphi_p_1, dphi_p_2, p_1, p_2, gap, lambda_max});
        }
        if (trace)
          trace->lambda_max = lambda_max;
        return lambda_max;
      }

      /* :585-597 */
      template <int dim>
      double compute(const std::array<double, dim + 2> &U_i, const std::array<double, dim + 2> &U_j,
                     const std::array<double, dim> &n_ij) const
      {
        return compute(riemann_data_from_state<dim>(U_i, n_ij),
                       riemann_data_from_state<dim>(U_j, n_ij));
      }
    };


    /* ---------------------------------------------------------------------
     * Indicator: source/euler/indicator.h:187-258
     * ------------------------------------------------------------------- */
    template <int dim>
    struct Indicator {
      using V = View<dim>;
      using state_type = typename V::state_type;
      using flux_type = typename V::flux_type;
      const V &view;
      double evc_factor;

      double rho_i_inverse = 0., eta_i = 0.;
      flux_type f_i;
      state_type d_eta_i;
      double left = 0.;
      state_type right;

      Indicator(const V &view, const ryujin_hip_params &p)
          : view(view)
          , evc_factor(p.indicator_evc_factor)
      {
      }

      void reset(const state_type &U_i, const double new_eta_i)
      {
        rho_i_inverse = 1. / U_i[0];
        eta_i = new_eta_i;
        d_eta_i = view.harten_entropy_derivative(U_i);
        d_eta_i[0] -= eta_i * rho_i_inverse;
        f_i = view.f(U_i);
        left = 0.;
        right.fill(0.);
      }

      void accumulate(const state_type &U_j, const double eta_j, const std::array<double, dim> &c_ij)
      {
        const double rho_j_inverse = 1. / U_j[0];
        const auto f_j = view.f(U_j);
        double m_j_c = 0.;
        for (int d = 0; d < dim; ++d)
          m_j_c += U_j[1 + d] * c_ij[d];
        const double entropy_flux = (eta_j * rho_j_inverse - eta_i * rho_i_inverse) * m_j_c;
        left += entropy_flux;
        for (int q = 0; q < V::k; ++q) {
          double component = 0.;
          for (int d = 0; d < dim; ++d)
            component += (f_j[q][d] - f_i[q][d]) * c_ij[d];
          right[q] += component;
        }
      }

      double alpha(const double hd_i) const
      {
        double numerator = left;
        double denominator = std::abs(left);
        for (int q = 0; q < V::k; ++q) {
          numerator -= d_eta_i[q] * right[q];
          denominator += std::abs(d_eta_i[q] * right[q]);
        }
        const double quotient = std::abs(numerator) / (denominator + hd_i * std::abs(eta_i));
        return std::min(1., evc_factor * quotient);
      }
    };


    /* ---------------------------------------------------------------------
     * Limiter: source/euler/limiter.h:255-363, limiter.template.h:15-327
     * ------------------------------------------------------------------- */

    struct LimiterTrace {
      double t_l_start = 0., t_r_start = 0.;
      bool density_violation_low_order = false, density_violation_high_order = false;
      bool entropy_violation_low_order = false, entropy_violation_high_order = false;
      struct Iter {
        int kind; /* 0 shortcut t_l==t_r, 1 break within tolerance, 2 newton step */
        double psi_l, psi_r, dpsi_l, dpsi_r, t_l, t_r;
      };
      std::vector<Iter> iterations;
    };

    template <int dim>
    struct Limiter {
      using V = View<dim>;
      using state_type = typename V::state_type;
      static constexpr int n_bounds = 3;
      using Bounds = std::array<double, n_bounds>;

      const V &view;
      double newton_tolerance, relaxation_factor;
      unsigned int newton_max_iterations;
      bool expensive_bounds_check = false;

      state_type U_i;
      Bounds bounds_;
      double rho_relaxation_numerator = 0., rho_relaxation_denominator = 0., s_interp_max = 0.;

      Limiter(const V &view, const ryujin_hip_params &p)
          : view(view)
          , newton_tolerance(p.limiter_newton_tolerance)
          , relaxation_factor(p.limiter_relaxation_factor)
          , newton_max_iterations((unsigned)p.limiter_newton_max_iterations)
      {
      }

      /* limiter.h:255-276 */
      void reset(const state_type &new_U_i)
      {
        U_i = new_U_i;
        bounds_[0] = std::numeric_limits<double>::max();
        bounds_[1] = 0.;
        bounds_[2] = std::numeric_limits<double>::max();
        rho_relaxation_numerator = 0.;
        rho_relaxation_denominator = 0.;
        s_interp_max = 0.;
      }

      /* limiter.h:279-327 (affine_shift == 0 for Euler) */
      void accumulate(const state_type &U_j, const double s_j,
                      const std::array<double, dim> &scaled_c_ij)
      {
        auto &[rho_min, rho_max, s_min] = bounds_;
        const double rho_i = U_i[0];
        const double rho_j = U_j[0];
        double dm_c = 0.;
        for (int d = 0; d < dim; ++d)
          dm_c += (U_i[1 + d] - U_j[1 + d]) * scaled_c_ij[d];
        const double rho_affine_shift = 0.;
        const double rho_ij_bar = 0.5 * (rho_i + rho_j + dm_c) + rho_affine_shift;
        rho_min = std::min(rho_min, rho_ij_bar);
        rho_max = std::max(rho_max, rho_ij_bar);
        s_min = std::min(s_min, s_j);

        const double beta_ij = 1.;
        rho_relaxation_numerator += beta_ij * (rho_i + rho_j);
        rho_relaxation_denominator += std::abs(beta_ij);

        state_type U_avg;
        for (int q = 0; q < V::k; ++q)
          U_avg[q] = (U_i[q] + U_j[q]) * .5;
        const double s_interp = view.specific_entropy(U_avg);
        s_interp_max = std::max(s_interp_max, s_interp);
      }

      /* limiter.h:330-363 */
      Bounds bounds(const double hd_i) const
      {
        auto relaxed_bounds = bounds_;
        auto &[rho_min, rho_max, s_min] = relaxed_bounds;

        double r_i = std::sqrt(hd_i);
        if constexpr (dim == 2) {
          const double t = std::sqrt(r_i);
          r_i = t * t * t;
        } else if constexpr (dim == 1) {
          r_i = r_i * r_i * r_i;
        }
        r_i *= relaxation_factor;

        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double rho_relaxation =
            std::abs(rho_relaxation_numerator) / (std::abs(rho_relaxation_denominator) + eps);
        const double relaxation = (2. * relaxation_factor) * rho_relaxation;

        rho_min = std::max((1. - r_i) * rho_min, rho_min - relaxation);
        rho_max = std::min((1. + r_i) * rho_max, rho_max + relaxation);

        const double entropy_relaxation = relaxation_factor * (s_interp_max - s_min);
        s_min = std::max((1. - r_i) * s_min, s_min - entropy_relaxation);
        return relaxed_bounds;
      }

      /* limiter.template.h:15-327 */
      std::pair<double, bool> limit(const Bounds &bounds, const state_type &U, const state_type &P,
                                    const double t_min = 0., const double t_max = 1.,
                                    LimiterTrace *trace = nullptr) const
      {
        bool success = true;
        double t_r = t_max;

        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double small = view.vacuum_state_relaxation_small;
        const double large = view.vacuum_state_relaxation_large;
        const double relax_small = 1. + small * eps;
        const double relax = 1. + large * eps;

        {
          const double rho_U = U[0];
          const double rho_P = P[0];
          const double rho_min = bounds[0];
          const double rho_max = bounds[1];

          const double test_min =
              view.filter_vacuum_density(std::max(0., rho_U - relax * rho_max));
          const double test_max =
              view.filter_vacuum_density(std::max(0., rho_min - relax * rho_U));
          if (!(test_min == 0. && test_max == 0.)) {
            success = false;
            if (trace)
              trace->density_violation_low_order = true;
          }

          const double denominator = 1. / (std::abs(rho_P) + eps * rho_max);

          t_r = rho_max < rho_U + t_r * rho_P ? (rho_max - rho_U) * denominator : t_r;
          t_r = rho_U + t_r * rho_P < rho_min ? (rho_U - rho_min) * denominator : t_r;

          t_r = std::min(t_r, t_max);
          t_r = std::max(t_r, t_min);

          if (expensive_bounds_check) {
            const double rho_new = U[0] + t_r * P[0];
            const double test_new_min =
                view.filter_vacuum_density(std::max(0., rho_new - relax * rho_max));
            const double test_new_max =
                view.filter_vacuum_density(std::max(0., rho_min - relax * rho_new));
            if (!(test_new_min == 0. && test_new_max == 0.)) {
              success = false;
              if (trace)
                trace->density_violation_high_order = true;
            }
          }
        }

        double t_l = t_min;

        const double gamma = view.gamma;
        const double gp1 = gamma + 1.;

        {
          const double s_min = bounds[2];
          if (trace) {
            trace->t_l_start = t_l;
            trace->t_r_start = t_r;
          }

          for (unsigned int n = 0; n < newton_max_iterations; ++n) {
            state_type U_r;
            for (int q = 0; q < V::k; ++q)
              U_r[q] = U[q] + t_r * P[q];
            const double rho_r = U_r[0];
            const double rho_r_gamma = std::pow(rho_r, gamma);
            const double rho_e_r = V::internal_energy(U_r);

            const double psi_r = relax_small * rho_r * rho_e_r - s_min * rho_r * rho_r_gamma;

            if (!expensive_bounds_check) {
              t_l = psi_r > 0. ? t_r : t_l;
              if (t_l == t_r) {
                if (trace)
                  trace->iterations.push_back({0, 0., psi_r, 0., 0., t_l, t_r});
                break;
              }
            }

            state_type U_l;
            for (int q = 0; q < V::k; ++q)
              U_l[q] = U[q] + t_l * P[q];
            const double rho_l = U_l[0];
            const double rho_l_gamma = std::pow(rho_l, gamma);
            const double rho_e_l = V::internal_energy(U_l);

            const double psi_l = relax_small * rho_l * rho_e_l - s_min * rho_l * rho_l_gamma;

            const double lower_bound = (1. - relax) * s_min * rho_l * rho_l_gamma;
            if (n == 0 && !(std::min(0., psi_l - lower_bound) == 0.)) {
              success = false;
              if (trace)
                trace->entropy_violation_low_order = true;
            }

            if (expensive_bounds_check)
              t_l = psi_r > 0. ? t_r : t_l;

            if (std::max(0., t_r - t_l - newton_tolerance) == 0.) {
              if (trace)
                trace->iterations.push_back({1, psi_l, psi_r, 0., 0., t_l, t_r});
              break;
            }

            const double drho = P[0];
            const auto de_l = V::internal_energy_derivative(U_l);
            const auto de_r = V::internal_energy_derivative(U_r);
            double drho_e_l = 0., drho_e_r = 0.;
            for (int q = 0; q < V::k; ++q) {
              drho_e_l += de_l[q] * P[q];
              drho_e_r += de_r[q] * P[q];
            }
            const double dpsi_l =
                rho_l * drho_e_l + (rho_e_l - gp1 * s_min * rho_l_gamma) * drho;
            const double dpsi_r =
                rho_r * drho_e_r + (rho_e_r - gp1 * s_min * rho_r_gamma) * drho;

            quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);

            if (trace)
              trace->iterations.push_back({2, psi_l, psi_r, dpsi_l, dpsi_r, t_l, t_r});
          }

          if (expensive_bounds_check) {
            state_type U_new;
            for (int q = 0; q < V::k; ++q)
              U_new[q] = U[q] + t_l * P[q];
            const double rho_new = U_new[0];
            const double rho_new_gamma = std::pow(rho_new, gamma);
            const double rho_e_new = V::internal_energy(U_new);
            const double psi_new =
                relax_small * rho_new * rho_e_new - s_min * rho_new * rho_new_gamma;
            const double lower_bound = (1. - relax) * s_min * rho_new * rho_new_gamma;
            const bool e_valid = std::min(0., rho_e_new) == 0.;
            const bool psi_valid = std::min(0., psi_new - lower_bound) == 0.;
            if (!e_valid || !psi_valid) {
              success = false;
              if (trace)
                trace->entropy_violation_high_order = true;
            }
          }
        }

        return {t_l, success};
      }
    };

  } // namespace euler
} // namespace oracle

// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// CPU restatement of the "Euler with arbitrary equation of state" Description
// (source/euler_aeos/): equation-of-state library (closed-form members), HyperbolicSystemView,
// RiemannSolver, Indicator, Limiter. Scalar double, ryujin::pow == std::pow.
//
// Parity status: function level PINNED by tests/golden/euler_aeos_*.output (the outputs of
// tests/euler_aeos/{riemann_solver,riemann_solver-strict,riemann_solver-strict-NASG,limiter,
// limiter-NASG,hyperbolic_system,equation_of_state_library}.cc); whole-step level PINNED by the
// isentropic-vortex "pge" verification outputs (tests/euler_aeos/verification-isentropic_vortex-pge-2d-*).

#pragma once

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "euler.hpp" /* positive_part, negative_part, quadratic_newton_step */
#include "ryujin_hip.h"

namespace oracle
{
  namespace aeos
  {
    /* EquationOfStateLibrary: source/euler_aeos/equation_of_state*.h */
    struct EquationOfState {
      int kind = RYUJIN_EOS_POLYTROPIC_GAS;
      double gamma = 1.4, b = 0., q = 0., pinf = 0., vdw_a = 0., R = 287.052874;
      double capA = 0., capB = 0., R1 = 1., R2 = 1., omega = 1., rho_0 = 1., q_0 = 0., jwl_cv = 1.;
      /* the NASG interpolation parameters of the surrogate (equation_of_state.h:40-60, :200-230) */
      double interpolation_b = 0., interpolation_pinfty = 0., interpolation_q = 0.;

      EquationOfState() = default;
      explicit EquationOfState(const ryujin_hip_params &p)
          : kind(p.eos)
          , gamma(p.gamma)
          , b(p.eos_covolume_b)
          , q(p.eos_q)
          , pinf(p.eos_pinf)
          , vdw_a(p.eos_vdw_a)
          , R(p.eos_gas_constant_R)
          , capA(p.jwl_A)
          , capB(p.jwl_B)
          , R1(p.jwl_R1)
          , R2(p.jwl_R2)
          , omega(p.jwl_omega)
          , rho_0(p.jwl_rho_0)
          , q_0(p.jwl_q_0)
          , jwl_cv(p.jwl_cv)
      {
        switch (kind) {
        case RYUJIN_EOS_POLYTROPIC_GAS:
        case RYUJIN_EOS_JONES_WILKINS_LEE:
          break;
        case RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS: /* ...noble_abel_stiffened_gas.h:52-56 */
          interpolation_b = b;
          interpolation_pinfty = pinf;
          interpolation_q = q;
          break;
        case RYUJIN_EOS_VAN_DER_WAALS: /* ...van_der_waals.h:46-52 */
          interpolation_b = b;
          if (b > 0.)
            interpolation_pinfty = vdw_a / (b * b);
          break;
        default:
          throw std::runtime_error("unknown equation of state");
        }
      }

      double jwl_cold(double rho) const
      {
        const double ratio = rho / rho_0;
        const double first_term = capA * (1. - omega / R1 * ratio) * std::exp(-R1 * 1. / ratio);
        const double second_term = capB * (1. - omega / R2 * ratio) * std::exp(-R2 * 1. / ratio);
        return first_term + second_term;
      }

      double pressure(double rho, double e) const
      {
        switch (kind) {
        case RYUJIN_EOS_POLYTROPIC_GAS: /* :50-53 */
          return (gamma - 1.) * rho * e;
        case RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS: /* :65-69 */
          return (gamma - 1.) * rho * (e - q) / (1. - b * rho) - gamma * pinf;
        case RYUJIN_EOS_VAN_DER_WAALS: { /* :62-68 */
          const double intermolecular = vdw_a * rho * rho;
          const double numerator = rho * e + intermolecular;
          const double covolume = 1. - b * rho;
          return (gamma - 1.) * numerator / covolume - intermolecular;
        }
        default: { /* Jones-Wilkins-Lee :77-87 */
          const double ratio = rho / rho_0;
          const double first_term = capA * (1. - omega / R1 * ratio) * std::exp(-R1 * 1. / ratio);
          const double second_term = capB * (1. - omega / R2 * ratio) * std::exp(-R2 * 1. / ratio);
          return first_term + second_term + omega * rho * (e + q_0);
        }
        }
      }

      double specific_internal_energy(double rho, double p) const
      {
        switch (kind) {
        case RYUJIN_EOS_POLYTROPIC_GAS:
          return p / (rho * (gamma - 1.));
        case RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS: {
          const double numerator = (p + gamma * pinf) * (1. - b * rho);
          const double denominator = rho * (gamma - 1.);
          return q + numerator / denominator;
        }
        case RYUJIN_EOS_VAN_DER_WAALS: {
          const double intermolecular = vdw_a * rho * rho;
          const double covolume = 1. - b * rho;
          const double numerator = (p + intermolecular) * covolume;
          const double denominator = rho * (gamma - 1.);
          return numerator / denominator - vdw_a * rho;
        }
        default: {
          const double ratio = rho / rho_0;
          const double first_term = capA * (1. - omega / R1 * ratio) * std::exp(-R1 * 1. / ratio);
          const double second_term = capB * (1. - omega / R2 * ratio) * std::exp(-R2 * 1. / ratio);
          return (p - first_term - second_term) / (rho * omega);
        }
        }
      }

      double temperature(double rho, double e) const
      {
        const double cv = R / (gamma - 1.);
        switch (kind) {
        case RYUJIN_EOS_POLYTROPIC_GAS:
          return e / cv;
        case RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS:
          return (e - q - pinf * (1. / rho - b)) / cv;
        case RYUJIN_EOS_VAN_DER_WAALS:
          return (e + vdw_a * rho) / cv;
        default: {
          const double ratio = rho / rho_0;
          const double first_term = capA / R1 * std::exp(-R1 * 1. / ratio);
          const double second_term = capB / R2 * std::exp(-R2 * 1. / ratio);
          return (e + q_0 - 1. / rho_0 * (first_term + second_term)) / jwl_cv;
        }
        }
      }

      double speed_of_sound(double rho, double e) const
      {
        switch (kind) {
        case RYUJIN_EOS_POLYTROPIC_GAS:
          return std::sqrt(gamma * (gamma - 1.) * e);
        case RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS: {
          const double covolume = 1. - b * rho;
          double numerator = (rho * (e - q) - pinf * covolume) / rho;
          numerator *= gamma * (gamma - 1.);
          return std::sqrt(numerator) / covolume;
        }
        case RYUJIN_EOS_VAN_DER_WAALS: {
          const double covolume = 1. - b * rho;
          const double numerator = gamma * (gamma - 1.) * (e + vdw_a * rho);
          return std::sqrt(numerator / (covolume * covolume) - 2. * vdw_a * rho);
        }
        default: {
          const double t1 = omega * rho / (R1 * rho_0);
          const double factor1 = omega * (1. - t1) * (1. + 1. / t1) - t1;
          const double first_term = capA / rho * factor1 * std::exp(-1. / t1 / omega);
          const double t2 = omega * rho / (R2 * rho_0);
          const double factor2 = omega * (1. - t2) * (1. + 1. / t2) - t2;
          const double second_term = capB / rho * factor2 * std::exp(-1. / t2 / omega);
          const double third_term = omega * (omega + 1.) * e;
          return std::sqrt(first_term + second_term + third_term);
        }
        }
      }
    };


    /* HyperbolicSystemView: source/euler_aeos/hyperbolic_system.h */
    template <int dim>
    struct View {
      static constexpr int problem_dimension = dim + 2;
      static constexpr int n_precomputed_values = 4; /* p, surrogate gamma(_min), s, eta  (:364-379) */
      using state_type = std::array<double, problem_dimension>;
      using vec_type = std::array<double, dim>;
      using flux_type = std::array<vec_type, problem_dimension>;
      using precomputed_type = std::array<double, n_precomputed_values>;

      EquationOfState eos;
      double reference_density, vacuum_small, vacuum_large;
      bool compute_strict_bounds;

      explicit View(const ryujin_hip_params &p)
          : eos(p)
          , reference_density(p.reference_density)
          , vacuum_small(p.vacuum_state_relaxation_small)
          , vacuum_large(p.vacuum_state_relaxation_large)
          , compute_strict_bounds(p.compute_strict_bounds != 0)
      {
      }

      double b() const { return eos.interpolation_b; }
      double pinf() const { return eos.interpolation_pinfty; }
      double q() const { return eos.interpolation_q; }

      static double density(const state_type &U) { return U[0]; }
      static vec_type momentum(const state_type &U)
      {
        vec_type m;
        for (int d = 0; d < dim; ++d)
          m[d] = U[1 + d];
        return m;
      }
      static double total_energy(const state_type &U) { return U[1 + dim]; }
      static double norm_square(const vec_type &m)
      {
        double s = 0.;
        for (int d = 0; d < dim; ++d)
          s += m[d] * m[d];
        return s;
      }

      /* :991-998 */
      double filter_vacuum_density(const double rho) const
      {
        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double rho_cutoff_large = reference_density * vacuum_large * eps;
        return std::abs(rho) < rho_cutoff_large ? 0. : rho;
      }

      /* :1033-1043 */
      static double internal_energy(const state_type &U)
      {
        const double rho_inverse = 1. / density(U);
        const auto m = momentum(U);
        const double E = total_energy(U);
        return E - 0.5 * norm_square(m) * rho_inverse;
      }

      /* :1049-1068 */
      static state_type internal_energy_derivative(const state_type &U)
      {
        const double rho_inverse = 1. / density(U);
        const auto m = momentum(U);
        state_type result;
        double u2 = 0.;
        for (int d = 0; d < dim; ++d) {
          const double u = m[d] * rho_inverse;
          u2 += u * u;
          result[1 + d] = -u;
        }
        result[0] = 0.5 * u2;
        result[dim + 1] = 1.;
        return result;
      }

      /* :1073-1088 */
      double surrogate_specific_entropy(const state_type &U, const double gamma_min) const
      {
        const double rho = density(U);
        const double rho_inverse = 1. / rho;
        const double covolume = 1. - b() * rho;
        const double shift = internal_energy(U) - rho * q() - pinf() * covolume;
        return shift * std::pow(rho_inverse - b(), gamma_min) / covolume;
      }

      /* :1093-1116 */
      double surrogate_harten_entropy(const state_type &U, const double gamma_min) const
      {
        const double rho = density(U);
        const auto m = momentum(U);
        const double E = total_energy(U);
        const double rho_rho_e_q = rho * E - 0.5 * norm_square(m) - rho * rho * q();
        const double exponent = 1. / (gamma_min + 1.);
        const double covolume = 1. - b() * rho;
        const double covolume_term = std::pow(covolume, gamma_min - 1.);
        const double rho_pinfcov = rho * pinf() * covolume;
        return std::pow((rho_rho_e_q - rho_pinfcov) * covolume_term, exponent);
      }

      /* :1121-1176 */
      state_type surrogate_harten_entropy_derivative(const state_type &U, const double eta,
                                                     const double gamma_min) const
      {
        const double rho = density(U);
        const auto m = momentum(U);
        const double E = total_energy(U);
        const double covolume = 1. - b() * rho;
        const double covolume_inverse = 1. / covolume;
        const double shift =
            rho * E - 0.5 * norm_square(m) - rho * rho * q() - rho * pinf() * covolume;
        const double factor = std::pow(eta * covolume_inverse, -gamma_min) *
                              (covolume_inverse * covolume_inverse) / (gamma_min + 1.);
        state_type result;
        const double first_term = E - 2. * rho * q() - pinf() * (1. - 2. * b() * rho);
        const double second_term = -(gamma_min - 1.) * shift * b();
        result[0] = factor * (covolume * first_term + second_term);
        for (int d = 0; d < dim; ++d)
          result[1 + d] = -factor * covolume * m[d];
        result[dim + 1] = factor * covolume * rho;
        return result;
      }

      /* :1181-1196 */
      double surrogate_gamma(const state_type &U, const double p) const
      {
        const double rho = density(U);
        const double rho_e = internal_energy(U);
        const double covolume = 1. - b() * rho;
        const double numerator = (p + pinf()) * covolume;
        const double denominator = rho_e - rho * q() - covolume * pinf();
        return 1. + numerator / denominator;
      }

      /* :1201-1214 */
      double surrogate_pressure(const state_type &U, const double gamma) const
      {
        const double rho = density(U);
        const double rho_e = internal_energy(U);
        const double covolume = 1. - b() * rho;
        return (gamma - 1.) * (rho_e - rho * q()) / covolume - gamma * pinf();
      }

      /* :1219-1250 */
      bool is_admissible(const state_type &U) const
      {
        const double rho = density(U);
        const double rho_e = internal_energy(U);
        const double covolume = 1. - b() * rho;
        const double shift = rho_e - rho * q() - pinf() * covolume;
        return rho > 0. && shift > 0.;
      }

      /* the EOS call of precomputation cycle 0 (:925-933) */
      double eos_pressure_of_state(const state_type &U) const
      {
        const double rho_i = density(U);
        const double e_i = internal_energy(U) / rho_i;
        return eos.pressure(rho_i, e_i);
      }

      /* :1314-1377; dynamic is __builtin_trap() in the reference: rejected by create() */
      state_type apply_boundary_conditions(const int id, const state_type &U, const vec_type &normal,
                                           const state_type &U_dirichlet) const
      {
        state_type result = U;
        if (id == RYUJIN_BC_DIRICHLET) {
          result = U_dirichlet;
        } else if (id == RYUJIN_BC_SLIP) {
          auto m = momentum(U);
          double mn = 0.;
          for (int d = 0; d < dim; ++d)
            mn += m[d] * normal[d];
          for (int d = 0; d < dim; ++d)
            result[1 + d] = m[d] - 1. * mn * normal[d];
        } else if (id == RYUJIN_BC_NO_SLIP) {
          for (int d = 0; d < dim; ++d)
            result[1 + d] = 0.;
        } else if (id == RYUJIN_BC_DYNAMIC) {
          throw std::runtime_error("euler_aeos: dynamic boundary conditions are not implemented "
                                   "in the reference (hyperbolic_system.h:1337)");
        }
        return result;
      }

      /* :1382-1400 */
      static flux_type f(const state_type &U, const double p)
      {
        const double rho_inverse = 1. / density(U);
        const auto m = momentum(U);
        const double E = total_energy(U);
        flux_type result;
        result[0] = m;
        for (int i = 0; i < dim; ++i) {
          for (int d = 0; d < dim; ++d)
            result[1 + i][d] = m[d] * (m[i] * rho_inverse);
          result[1 + i][i] += p;
        }
        for (int d = 0; d < dim; ++d)
          result[dim + 1][d] = m[d] * (rho_inverse * (E + p));
        return result;
      }

      /* :1437-1445: -contract(add(flux_i, flux_j), c_ij) */
      static state_type flux_divergence(const flux_type &flux_i, const flux_type &flux_j,
                                        const vec_type &c_ij)
      {
        state_type result;
        for (int k = 0; k < problem_dimension; ++k) {
          double s = 0.;
          for (int d = 0; d < dim; ++d)
            s += (flux_i[k][d] + flux_j[k][d]) * c_ij[d];
          result[k] = -s;
        }
        return result;
      }

      /* from_initial_state (:1470-1486) + from_primitive_state (:1491-1512): (rho, u, p) -> U */
      state_type from_initial_state(const double rho, const vec_type &u, const double p) const
      {
        const double e = eos.specific_internal_energy(rho, p);
        state_type U;
        U[0] = rho;
        double u2 = 0.;
        for (int d = 0; d < dim; ++d) {
          U[1 + d] = rho * u[d];
          u2 += u[d] * u[d];
        }
        U[dim + 1] = rho * e + 0.5 * rho * u2;
        return U;
      }
    };


    /* RiemannSolver: source/euler_aeos/riemann_solver.template.h */
    struct RiemannTrace {
      double rs_p_1 = 0., rs_p_2 = 0., ss_p_1 = 0., ss_p_2 = 0., interpolated = 0., p_star = 0.,
             phi_p_star = 0.;
    };

    struct RiemannSolver {
      using primitive_type = std::array<double, 5>; /* rho, u, p, gamma, a */
      double b, pinf;
      bool strict;
      mutable RiemannTrace *trace = nullptr;

      RiemannSolver(double interpolation_b, double interpolation_pinfty, bool compute_strict_bounds)
          : b(interpolation_b)
          , pinf(interpolation_pinfty)
          , strict(compute_strict_bounds)
      {
      }

      /* :21-36 */
      static double c(const double gamma)
      {
        constexpr double slope = -0.34976871477801828189920753948709;
        const double first_radicand = (3. * gamma + 11.) / (6. * gamma + 6.);
        const double second_radicand = 5. / 6. + slope * (gamma - 3.);
        double radicand = std::min(first_radicand, second_radicand);
        radicand = std::min(1., radicand);
        radicand = std::max(1. / 2., radicand);
        return std::sqrt(radicand);
      }

      /* :39-50 */
      double alpha(const double rho, const double gamma, const double a) const
      {
        const double numerator = 2. * a * (1. - b * rho);
        const double denominator = gamma - 1.;
        return numerator / denominator;
      }

      /* :53-120 */
      double p_star_RS_full(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, gamma_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, gamma_j, a_j] = rd_j;
        const double alpha_i = alpha(rho_i, gamma_i, a_i);
        const double alpha_j = alpha(rho_j, gamma_j, a_j);
        const double p_min = std::min(p_i, p_j);
        const double p_max = std::max(p_i, p_j);
        const double gamma_min = p_i < p_j ? gamma_i : gamma_j;
        const double alpha_min = p_i < p_j ? alpha_i : alpha_j;
        const double alpha_hat_min = c(gamma_min) * alpha_min;
        const double alpha_max = p_i >= p_j ? alpha_i : alpha_j;
        const double gamma_m = std::min(gamma_i, gamma_j);
        const double gamma_M = std::max(gamma_i, gamma_j);
        const double numerator = positive_part(alpha_hat_min + alpha_max - (u_j - u_i));
        const double p_ratio = (p_min + pinf) / (p_max + pinf);
        const double r_exponent = (gamma_M - gamma_min) / (2. * gamma_min * gamma_M);
        const double first_exponent = (gamma_M - 1.) / (2. * gamma_M);
        const double first_exponent_inverse = 1. / first_exponent;
        const double first_denom =
            alpha_hat_min * std::pow(p_ratio, r_exponent - first_exponent) + alpha_max;
        const double p_1_tilde =
            (p_max + pinf) * std::pow(numerator / first_denom, first_exponent_inverse) - pinf;
        const double second_exponent = (gamma_m - 1.) / (2. * gamma_m);
        const double second_exponent_inverse = 1. / second_exponent;
        const double second_denom = alpha_hat_min * std::pow(p_ratio, -second_exponent) +
                                    alpha_max * std::pow(p_ratio, r_exponent);
        const double p_2_tilde =
            (p_max + pinf) * std::pow(numerator / second_denom, second_exponent_inverse) - pinf;
        if (trace) {
          trace->rs_p_1 = p_1_tilde;
          trace->rs_p_2 = p_2_tilde;
        }
        return std::min(p_1_tilde, p_2_tilde);
      }

      /* :161-198 */
      double p_star_failsafe(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, gamma_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, gamma_j, a_j] = rd_j;
        const double p_max = std::max(p_i, p_j) + pinf;
        double radicand_i = 2. * (1. - b * rho_i) * p_max;
        radicand_i /= rho_i * ((gamma_i + 1.) * p_max + (gamma_i - 1.) * (p_i + pinf));
        const double x_i = std::sqrt(radicand_i);
        double radicand_j = 2. * (1. - b * rho_j) * p_max;
        radicand_j /= rho_j * ((gamma_j + 1.) * p_max + (gamma_j - 1.) * (p_j + pinf));
        const double x_j = std::sqrt(radicand_j);
        const double a = x_i + x_j;
        const double bb = u_j - u_i;
        const double cc = -(p_i + pinf) * x_i - (p_j + pinf) * x_j;
        const double base = (-bb + std::sqrt(bb * bb - 4. * a * cc)) / (2. * a);
        const double p_2_tilde = base * base - pinf;
        if (trace)
          trace->ss_p_2 = p_2_tilde;
        return p_2_tilde;
      }

      /* :123-158 */
      double p_star_SS_full(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, gamma_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, gamma_j, a_j] = rd_j;
        const double gamma_m = std::min(gamma_i, gamma_j);
        const double alpha_hat_i = c(gamma_i) * alpha(rho_i, gamma_i, a_i);
        const double alpha_hat_j = c(gamma_j) * alpha(rho_j, gamma_j, a_j);
        const double exponent = (gamma_m - 1.) / (2. * gamma_m);
        const double exponent_inverse = 1. / exponent;
        const double numerator = positive_part(alpha_hat_i + alpha_hat_j - (u_j - u_i));
        const double denominator =
            alpha_hat_i * std::pow((p_i + pinf) / (p_j + pinf), -exponent) + alpha_hat_j;
        const double p_1_tilde =
            (p_j + pinf) * std::pow(numerator / denominator, exponent_inverse) - pinf;
        if (trace)
          trace->ss_p_1 = p_1_tilde;
        const double p_2_tilde = p_star_failsafe(rd_i, rd_j);
        return std::min(p_1_tilde, p_2_tilde);
      }

      /* :201-255 */
      double p_star_interpolated(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, gamma_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, gamma_j, a_j] = rd_j;
        const double alpha_i = alpha(rho_i, gamma_i, a_i);
        const double alpha_j = alpha(rho_j, gamma_j, a_j);
        const double p_min = std::min(p_i, p_j) + pinf;
        const double p_max = std::max(p_i, p_j) + pinf;
        const double gamma_min = p_i < p_j ? gamma_i : gamma_j;
        const double alpha_min = p_i < p_j ? alpha_i : alpha_j;
        const double alpha_hat_min = c(gamma_min) * alpha_min;
        const double gamma_max = p_i >= p_j ? gamma_i : gamma_j;
        const double alpha_max = p_i >= p_j ? alpha_i : alpha_j;
        const double alpha_hat_max = c(gamma_max) * alpha_max;
        const double gamma_m = std::min(gamma_i, gamma_j);
        const double gamma_M = std::max(gamma_i, gamma_j);
        const double p_ratio = p_min / p_max;
        const double r_exponent = (gamma_M - gamma_min) / (2. * gamma_min * gamma_M);
        const double exponent = (gamma_m - 1.) / (2. * gamma_m);
        const double exponent_inverse = 1. / exponent;
        const double numerator =
            positive_part(alpha_hat_min + /*SIC!*/ alpha_max - (u_j - u_i));
        const double denominator = alpha_hat_min * std::pow(p_ratio, -exponent) +
                                   alpha_hat_max * std::pow(p_ratio, r_exponent);
        const double p_tilde = p_max * std::pow(numerator / denominator, exponent_inverse) - pinf;
        if (trace)
          trace->interpolated = p_tilde;
        return p_tilde;
      }

      /* :258-291 */
      double f(const primitive_type &rd, const double p_star) const
      {
        const auto &[rho, u, p, gamma, a] = rd;
        const double one_minus_b_rho = 1. - b * rho;
        const double gamma_minus_one = gamma - 1.;
        const double Az = 2. * one_minus_b_rho / (rho * (gamma + 1.));
        const double Bz = gamma_minus_one / (gamma + 1.) * (p + pinf);
        const double radicand = Az / (p_star + pinf + Bz);
        const double true_value = (p_star - p) * std::sqrt(radicand);
        const double exponent = 0.5 * gamma_minus_one / gamma;
        const double ratio = (p_star + pinf) / (p + pinf);
        const double factor = std::pow(ratio, exponent) - 1.;
        const double false_value = 2. * a * one_minus_b_rho * factor / gamma_minus_one;
        return p_star >= p ? true_value : false_value;
      }

      /* :294-304 */
      double phi(const primitive_type &rd_i, const primitive_type &rd_j, const double p_in) const
      {
        return f(rd_i, p_in) + f(rd_j, p_in) + rd_j[1] - rd_i[1];
      }

      /* :307-339 */
      double phi_of_p_max(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const auto &[rho_i, u_i, p_i, gamma_i, a_i] = rd_i;
        const auto &[rho_j, u_j, p_j, gamma_j, a_j] = rd_j;
        const double p_max = std::max(p_i, p_j) + pinf;
        const double radicand_inverse_i = 0.5 * rho_i / (1. - b * rho_i) *
                                          ((gamma_i + 1.) * p_max + (gamma_i - 1.) * (p_i + pinf));
        const double value_i = (p_max - p_i) / std::sqrt(radicand_inverse_i);
        const double radicand_inverse_j = 0.5 * rho_j / (1. - b * rho_j) *
                                          ((gamma_j + 1.) * p_max + (gamma_j - 1.) * (p_j + pinf));
        const double value_j = (p_max - p_j) / std::sqrt(radicand_inverse_j);
        return value_i + value_j + u_j - u_i;
      }

      /* :342-378 */
      double lambda1_minus(const primitive_type &rd, const double p_star) const
      {
        const auto &[rho, u, p, gamma, a] = rd;
        const double factor = 0.5 * (gamma + 1.) / gamma;
        const double tmp = positive_part((p_star - p) / (p + pinf));
        return u - a * std::sqrt(1. + factor * tmp);
      }
      double lambda3_plus(const primitive_type &rd, const double p_star) const
      {
        const auto &[rho, u, p, gamma, a] = rd;
        const double factor = 0.5 * (gamma + 1.) / gamma;
        const double tmp = positive_part((p_star - p) / (p + pinf));
        return u + a * std::sqrt(1. + factor * tmp);
      }

      /* :381-392 */
      double compute_lambda(const primitive_type &rd_i, const primitive_type &rd_j,
                            const double p_star) const
      {
        const double nu_11 = lambda1_minus(rd_i, p_star);
        const double nu_32 = lambda3_plus(rd_j, p_star);
        return std::max(positive_part(nu_32), negative_part(nu_11));
      }

      /* :395-440 */
      template <int dim>
      primitive_type riemann_data_from_state(const View<dim> &view,
                                             const typename View<dim>::state_type &U, const double p,
                                             const std::array<double, dim> &n_ij) const
      {
        const double rho = View<dim>::density(U);
        const double rho_inverse = 1.0 / rho;
        const auto m = View<dim>::momentum(U);
        double proj_m = 0.;
        for (int d = 0; d < dim; ++d)
          proj_m += n_ij[d] * m[d];
        const double gamma = view.surrogate_gamma(U, p);
        const double x = 1. - b * rho;
        const double a = std::sqrt(gamma * (p + pinf) / (rho * x));
        return {{rho, proj_m * rho_inverse, p, gamma, a}};
      }

      /* :443-560 */
      double compute(const primitive_type &rd_i, const primitive_type &rd_j) const
      {
        const double p_max = std::max(rd_i[2], rd_j[2]) + pinf;
        const double phi_p_max = phi_of_p_max(rd_i, rd_j);

        if (!strict) {
          if (trace) { /* the reference's DEBUG_RIEMANN_SOLVER block evaluates both full estimates */
            const double p_star_RS = p_star_RS_full(rd_i, rd_j);
            const double p_star_SS = p_star_SS_full(rd_i, rd_j);
            (void)p_star_RS;
            (void)p_star_SS;
          }
          const double p_star_tilde = p_star_interpolated(rd_i, rd_j);
          const double p_star_backup = p_star_failsafe(rd_i, rd_j);
          const double p_2 = phi_p_max < 0. ? std::min(p_star_tilde, p_star_backup)
                                            : std::min(p_max, p_star_tilde);
          if (trace) {
            trace->p_star = p_2;
            trace->phi_p_star = phi(rd_i, rd_j, p_2);
          }
          return compute_lambda(rd_i, rd_j, p_2);
        }

        const double p_star_RS = p_star_RS_full(rd_i, rd_j);
        const double p_star_SS = p_star_SS_full(rd_i, rd_j);
        const double p_2 = phi_p_max < 0. ? p_star_SS : std::min(p_max, p_star_RS);
        if (trace) {
          trace->p_star = p_2;
          trace->phi_p_star = phi(rd_i, rd_j, p_2);
        }
        return compute_lambda(rd_i, rd_j, p_2);
      }

      /* :563-582: p_i, p_j are the precomputed EOS pressures */
      template <int dim>
      double compute(const View<dim> &view, const typename View<dim>::state_type &U_i, const double p_i,
                     const typename View<dim>::state_type &U_j, const double p_j,
                     const std::array<double, dim> &n_ij) const
      {
        const auto rd_i = riemann_data_from_state<dim>(view, U_i, p_i, n_ij);
        const auto rd_j = riemann_data_from_state<dim>(view, U_j, p_j, n_ij);
        return compute(rd_i, rd_j);
      }
    };


    /* Indicator: source/euler_aeos/indicator.h:187-262 */
    template <int dim>
    struct Indicator {
      using V = View<dim>;
      using state_type = typename V::state_type;
      using vec_type = typename V::vec_type;
      const V &view;
      double evc_factor;

      double rho_i_inverse = 0., eta_i = 0., gamma_min = 0., left = 0.;
      typename V::flux_type f_i;
      state_type d_eta_i, right;

      Indicator(const V &view, const ryujin_hip_params &p)
          : view(view)
          , evc_factor(p.indicator_evc_factor)
      {
      }

      void reset(const state_type &U_i, const typename V::precomputed_type &prec_i)
      {
        gamma_min = prec_i[1];
        const double rho_i = V::density(U_i);
        rho_i_inverse = 1. / rho_i;
        eta_i = prec_i[3];
        d_eta_i = view.surrogate_harten_entropy_derivative(U_i, eta_i, gamma_min);
        d_eta_i[0] -= eta_i * rho_i_inverse;
        const double surrogate_p_i = view.surrogate_pressure(U_i, gamma_min);
        f_i = V::f(U_i, surrogate_p_i);
        left = 0.;
        right.fill(0.);
      }

      void accumulate(const state_type &U_j, const vec_type &c_ij)
      {
        const double eta_j = view.surrogate_harten_entropy(U_j, gamma_min);
        const double rho_j = V::density(U_j);
        const double rho_j_inverse = 1. / rho_j;
        const auto m_j = V::momentum(U_j);
        const double surrogate_p_j = view.surrogate_pressure(U_j, gamma_min);
        const auto f_j = V::f(U_j, surrogate_p_j);
        double m_c = 0.;
        for (int d = 0; d < dim; ++d)
          m_c += m_j[d] * c_ij[d];
        const double entropy_flux = (eta_j * rho_j_inverse - eta_i * rho_i_inverse) * m_c;
        left += entropy_flux;
        for (int k = 0; k < V::problem_dimension; ++k) {
          double component = 0.;
          for (int d = 0; d < dim; ++d)
            component += (f_j[k][d] - f_i[k][d]) * c_ij[d];
          right[k] += component;
        }
      }

      double alpha(const double hd_i) const
      {
        double numerator = left;
        double denominator = std::abs(left);
        for (int k = 0; k < V::problem_dimension; ++k) {
          numerator -= d_eta_i[k] * right[k];
          denominator += std::abs(d_eta_i[k] * right[k]);
        }
        const double quotient = std::abs(numerator) / (denominator + hd_i * std::abs(eta_i));
        return std::min(1., evc_factor * quotient);
      }
    };


    /* Limiter: source/euler_aeos/limiter.h:255-455, limiter.template.h:15-360 */
    struct LimiterTrace {
      double t_l_start = 0., t_r_start = 0.;
      struct Iter {
        double psi_l, psi_r, dpsi_l, dpsi_r, t_l, t_r;
        bool newton;
      };
      std::vector<Iter> iters;
    };

    template <int dim>
    struct Limiter {
      using V = View<dim>;
      using state_type = typename V::state_type;
      using vec_type = typename V::vec_type;
      using flux_type = typename V::flux_type;
      static constexpr int n_bounds = 4; /* rho_min, rho_max, s_min, gamma_min */
      using Bounds = std::array<double, n_bounds>;

      const V &view;
      double relaxation_factor, newton_tolerance;
      int newton_max_iterations;
      bool expensive_bounds_check = false;
      LimiterTrace *trace = nullptr;

      state_type U_i;
      flux_type flux_i;
      Bounds bounds_;
      double rho_relaxation_numerator = 0., rho_relaxation_denominator = 0., s_interp_max = 0.;

      Limiter(const V &view, const ryujin_hip_params &p)
          : view(view)
          , relaxation_factor(p.limiter_relaxation_factor)
          , newton_tolerance(p.limiter_newton_tolerance)
          , newton_max_iterations(p.limiter_newton_max_iterations)
      {
      }

      /* limiter.h:258-284 */
      void reset(const state_type &new_U_i, const flux_type &new_flux_i, const double gamma_min_i)
      {
        U_i = new_U_i;
        flux_i = new_flux_i;
        bounds_[0] = std::numeric_limits<double>::max();
        bounds_[1] = 0.;
        bounds_[2] = std::numeric_limits<double>::max();
        bounds_[3] = gamma_min_i;
        rho_relaxation_numerator = 0.;
        rho_relaxation_denominator = 0.;
        s_interp_max = 0.;
      }

      /* limiter.h:287-353; s_j_precomputed is only read with compute_strict_bounds == false */
      void accumulate(const state_type &U_j, const flux_type &flux_j, const vec_type &scaled_c_ij,
                      const double s_j_precomputed)
      {
        auto &[rho_min, rho_max, s_min, gamma_min] = bounds_;
        const double rho_i = V::density(U_i);
        const double rho_j = V::density(U_j);

        state_type U_ij_bar, U_avg;
        for (int k = 0; k < V::problem_dimension; ++k) {
          double contracted = 0.;
          for (int d = 0; d < dim; ++d)
            contracted += (flux_j[k][d] + (-flux_i[k][d])) * scaled_c_ij[d];
          U_ij_bar[k] = 0.5 * (U_i[k] + U_j[k]) - 0.5 * contracted + 0.;
          U_avg[k] = (U_i[k] + U_j[k]) * .5;
        }
        const double rho_ij_bar = V::density(U_ij_bar);

        rho_min = std::min(rho_min, rho_ij_bar);
        rho_max = std::max(rho_max, rho_ij_bar);

        const double beta_ij = 1.;
        rho_relaxation_numerator += beta_ij * (rho_i + rho_j);
        rho_relaxation_denominator += std::abs(beta_ij);

        if (view.compute_strict_bounds) {
          const double s_j = view.surrogate_specific_entropy(U_j, gamma_min);
          const double s_ij_bar = view.surrogate_specific_entropy(U_ij_bar, gamma_min);
          const double s_interp = view.surrogate_specific_entropy(U_avg, gamma_min);
          s_min = std::min(s_min, s_j);
          s_min = std::min(s_min, s_ij_bar);
          s_interp_max = std::max(s_interp_max, s_interp);
        } else {
          const double s_j = s_j_precomputed;
          const double s_ij_bar = view.surrogate_specific_entropy(U_ij_bar, gamma_min);
          s_min = std::min(s_min, s_j);
          s_min = std::min(s_min, s_ij_bar);
          s_interp_max = std::max(s_interp_max, s_ij_bar);
        }
      }

      /* limiter.h:356-410 */
      Bounds bounds(const double hd_i) const
      {
        auto relaxed_bounds = bounds_;
        auto &[rho_min, rho_max, s_min, gamma_min] = relaxed_bounds;

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

        const double numerator = (gamma_min + 1.) * rho_max;
        const double denominator = gamma_min - 1. + 2. * view.b() * rho_max;
        const double upper_bound = numerator / denominator;
        rho_max = std::min(upper_bound, rho_max);

        return relaxed_bounds;
      }

      /* limiter.template.h:15-360 */
      std::tuple<double, bool> limit(const Bounds &bounds, const state_type &U, const state_type &P,
                                     const double t_min = 0., const double t_max = 1.)
      {
        bool success = true;
        double t_r = t_max;

        constexpr double eps = std::numeric_limits<double>::epsilon();
        const double relax_small = 1. + view.vacuum_small * eps;
        const double relax = 1. + view.vacuum_large * eps;

        auto axpy = [](const state_type &a, double t, const state_type &b) {
          state_type r;
          for (int k = 0; k < V::problem_dimension; ++k)
            r[k] = a[k] + t * b[k];
          return r;
        };
        auto dot = [](const state_type &a, const state_type &b) {
          double s = 0.;
          for (int k = 0; k < V::problem_dimension; ++k)
            s += a[k] * b[k];
          return s;
        };

        {
          const double rho_U = V::density(U);
          const double rho_P = V::density(P);
          const double rho_min = bounds[0];
          const double rho_max = bounds[1];

          const double test_min =
              view.filter_vacuum_density(std::max(0., rho_U - relax * rho_max));
          const double test_max =
              view.filter_vacuum_density(std::max(0., rho_min - relax * rho_U));
          if (!(test_min == 0. && test_max == 0.))
            success = false;

          const double denominator = 1. / (std::abs(rho_P) + eps * rho_max);
          t_r = rho_max < rho_U + t_r * rho_P ? (rho_max - rho_U) * denominator : t_r;
          t_r = rho_U + t_r * rho_P < rho_min ? (rho_U - rho_min) * denominator : t_r;
          t_r = std::min(t_r, t_max);
          t_r = std::max(t_r, t_min);

          if (expensive_bounds_check) {
            const double rho_new = V::density(axpy(U, t_r, P));
            const double test_new_min =
                view.filter_vacuum_density(std::max(0., rho_new - relax * rho_max));
            const double test_new_max =
                view.filter_vacuum_density(std::max(0., rho_min - relax * rho_new));
            if (!(test_new_min == 0. && test_new_max == 0.))
              success = false;
          }
        }

        double t_l = t_min;
        const double gamma = bounds[3];
        const double gm1 = gamma - 1.;
        const double b = view.b(), pinf = view.pinf(), q = view.q();

        {
          const double s_min = bounds[2];
          if (trace) {
            trace->t_l_start = t_l;
            trace->t_r_start = t_r;
            trace->iters.clear();
          }

          for (int n = 0; n < newton_max_iterations; ++n) {
            const auto U_r = axpy(U, t_r, P);
            const double rho_r = V::density(U_r);
            const double rho_r_gamma = std::pow(rho_r, gamma);
            const double covolume_r = 1. - b * rho_r;
            const double rho_e_r = V::internal_energy(U_r);
            const double shift_r = rho_e_r - rho_r * q - pinf * covolume_r;
            double psi_r = relax_small * rho_r * shift_r -
                           s_min * rho_r * rho_r_gamma * std::pow(covolume_r, -gm1);

            if (!expensive_bounds_check) {
              t_l = psi_r > 0. ? t_r : t_l;
              if (t_l == t_r)
                break;
            }

            const auto U_l = axpy(U, t_l, P);
            const double rho_l = V::density(U_l);
            const double rho_l_gamma = std::pow(rho_l, gamma);
            const double covolume_l = 1. - b * rho_l;
            const double rho_e_l = V::internal_energy(U_l);
            const double shift_l = rho_e_l - rho_l * q - pinf * covolume_l;
            double psi_l = relax_small * rho_l * shift_l -
                           s_min * rho_l * rho_l_gamma * std::pow(covolume_l, -gm1);

            const double lower_bound =
                (1. - relax) * s_min * rho_l * rho_l_gamma * std::pow(covolume_l, -gm1);
            if (n == 0 && !(std::min(0., psi_l - lower_bound) == 0.))
              success = false;

            if (expensive_bounds_check)
              t_l = psi_r > 0. ? t_r : t_l;

            if (std::max(0., t_r - t_l - newton_tolerance) == 0.) {
              if (trace)
                trace->iters.push_back({psi_l, psi_r, 0., 0., t_l, t_r, false});
              break;
            }

            const double drho = V::density(P);
            const double drho_e_l = dot(V::internal_energy_derivative(U_l), P);
            const double drho_e_r = dot(V::internal_energy_derivative(U_r), P);
            const double q_pinf_term_l = 2. * rho_l * q + pinf * (1. - 2. * b * rho_l);
            const double q_pinf_term_r = 2. * rho_r * q + pinf * (1. - 2. * b * rho_r);
            const double extra_term_l =
                s_min * std::pow(rho_l / covolume_l, gamma) * (covolume_l + gamma - b * rho_l);
            const double extra_term_r =
                s_min * std::pow(rho_r / covolume_r, gamma) * (covolume_r + gamma - b * rho_r);
            const double dpsi_l = rho_l * drho_e_l + (rho_e_l - q_pinf_term_l - extra_term_l) * drho;
            const double dpsi_r = rho_r * drho_e_r + (rho_e_r - q_pinf_term_r - extra_term_r) * drho;

            quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
            if (trace)
              trace->iters.push_back({psi_l, psi_r, dpsi_l, dpsi_r, t_l, t_r, true});
          }

          if (expensive_bounds_check) {
            const auto U_new = axpy(U, t_l, P);
            const double rho_new = V::density(U_new);
            const double covolume_new = 1. - b * rho_new;
            const double rho_new_gamma = std::pow(rho_new, gamma);
            const double rho_e_new = V::internal_energy(U_new);
            const double shift_new = rho_e_new - rho_new * q - pinf * covolume_new;
            const double psi_new = relax_small * rho_new * shift_new -
                                   s_min * rho_new * rho_new_gamma * std::pow(covolume_new, -gm1);
            const double lower_bound =
                (1. - relax) * s_min * rho_new * rho_new_gamma * std::pow(covolume_new, -gm1);
            const bool e_valid = std::min(0., shift_new) == 0.;
            const bool psi_valid = std::min(0., psi_new - lower_bound) == 0.;
            if (!e_valid || !psi_valid)
              success = false;
          }
        }
        return {t_l, success};
      }
    };
  } // namespace aeos
} // namespace oracle

// Device-side "Euler with arbitrary equation of state" Description.
//
// Restates (operation order preserved):
//   EquationOfStateLibrary  source/euler_aeos/equation_of_state_{polytropic_gas,noble_abel_stiffened_gas,
//                           van_der_waals,jones_wilkins_lee}.h  (pressure only: the hot path calls nothing else)
//   HyperbolicSystemView    source/euler_aeos/hyperbolic_system.h:843-1445
//   RiemannSolver           source/euler_aeos/riemann_solver.template.h:21-582
//   Indicator               source/euler_aeos/indicator.h:187-262
//   Limiter                 source/euler_aeos/limiter.h:255-410, limiter.template.h:15-360
// Precomputed values per DoF: (p, gamma_min, s, eta) -- hyperbolic_system.h:364-379.

#pragma once

#include "euler_device.hpp"
#include "ryujin_hip.h"

namespace ryujin_hip
{
  struct EulerAeosParams {
    int eos, strict;
    double gamma, eos_b, eos_q, eos_pinf, vdw_a;
    double jwl_A, jwl_B, jwl_R1, jwl_R2, jwl_omega, jwl_rho_0, jwl_q_0;
    double b, pinf, q; /* NASG interpolation of the surrogate: eos_interpolation_{b,pinfty,q}() */
    double reference_density, vacuum_small, vacuum_large;
    double evc_factor;
    double lim_newton_tolerance, lim_relaxation_factor;
    int lim_newton_max_iterations;
  };

  template <int DIM>
  struct EulerAeos {
    static constexpr int DIMENSION = DIM;
    static constexpr int K = DIM + 2;
    static constexpr int NB = 4;    /* rho_min, rho_max, s_min, gamma_min: limiter.h:111 */
    static constexpr bool kFusablePrecompute = false; /* two precomputation cycles, the second over the stencil */
    /* steps 6/7 may form a limited row's update as V_i - sum (1 - l_ij) lambda P_ij (kernels_limiter.hpp): another
     * rounding of the reference's sum. Not where l = 0 has to return the low-order update EXACTLY (a dry node) */
    static constexpr bool kLimitedUpdateFromV = true;
    static constexpr int NPREC = 4; /* p, gamma_min, s, eta */
    using Params = EulerAeosParams;

    struct Prec {
      double p, gamma_min, s, eta;
    };
    static RYUJIN_DEV Prec load_prec(const double *__restrict__ prec, const uint32_t i)
    {
      const double2 a = reinterpret_cast<const double2 *>(prec)[2 * (size_t)i];
      const double2 b = reinterpret_cast<const double2 *>(prec)[2 * (size_t)i + 1];
      return {a.x, a.y, b.x, b.y};
    }
    static RYUJIN_DEV void store_prec(double *__restrict__ prec, const uint32_t i, const Prec &v)
    {
      double2 a, b;
      a.x = v.p;
      a.y = v.gamma_min;
      b.x = v.s;
      b.y = v.eta;
      reinterpret_cast<double2 *>(prec)[2 * (size_t)i] = a;
      reinterpret_cast<double2 *>(prec)[2 * (size_t)i + 1] = b;
    }

    static RYUJIN_DEV double momentum_norm_square(const double (&U)[K])
    {
      double m2 = U[1] * U[1];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        m2 += U[1 + d] * U[1 + d];
      return m2;
    }

    /* hyperbolic_system.h:1033-1043 */
    static RYUJIN_DEV double internal_energy(const double (&U)[K])
    {
      const double rho_inverse = 1. / U[0];
      return U[1 + DIM] - 0.5 * momentum_norm_square(U) * rho_inverse;
    }

    /* EquationOfState::pressure(rho, e) */
    static RYUJIN_DEV double eos_pressure(const Params &P, const double rho, const double e)
    {
      switch (P.eos) {
      case RYUJIN_EOS_POLYTROPIC_GAS:
        return (P.gamma - 1.) * rho * e;
      case RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS:
        return (P.gamma - 1.) * rho * (e - P.eos_q) / (1. - P.eos_b * rho) - P.gamma * P.eos_pinf;
      case RYUJIN_EOS_VAN_DER_WAALS: {
        const double intermolecular = P.vdw_a * rho * rho;
        const double numerator = rho * e + intermolecular;
        const double covolume = 1. - P.eos_b * rho;
        return (P.gamma - 1.) * numerator / covolume - intermolecular;
      }
      default: {
        const double ratio = rho / P.jwl_rho_0;
        const double first_term =
            P.jwl_A * (1. - P.jwl_omega / P.jwl_R1 * ratio) * exp(-P.jwl_R1 * 1. / ratio);
        const double second_term =
            P.jwl_B * (1. - P.jwl_omega / P.jwl_R2 * ratio) * exp(-P.jwl_R2 * 1. / ratio);
        return first_term + second_term + P.jwl_omega * rho * (e + P.jwl_q_0);
      }
      }
    }

    /* hyperbolic_system.h:1181-1196 */
    static RYUJIN_DEV double surrogate_gamma(const Params &P, const double (&U)[K], const double p)
    {
      const double rho = U[0];
      const double rho_e = internal_energy(U);
      const double covolume = 1. - P.b * rho;
      const double numerator = (p + P.pinf) * covolume;
      const double denominator = rho_e - rho * P.q - covolume * P.pinf;
      return 1. + numerator / denominator;
    }

    /* hyperbolic_system.h:1201-1214 */
    static RYUJIN_DEV double surrogate_pressure(const Params &P, const double (&U)[K], const double gamma)
    {
      const double rho = U[0];
      const double rho_e = internal_energy(U);
      const double covolume = 1. - P.b * rho;
      return (gamma - 1.) * (rho_e - rho * P.q) / covolume - gamma * P.pinf;
    }

    /* (1 - b rho)^x. Without a covolume (b == 0 is a kernel argument: a scalar branch) the base is exactly 1
     * and pow(1, x) == 1 exactly: the polytropic gas skips one of its two powers per entropy evaluation */
    static RYUJIN_DEV double covolume_pow(const Params &P, const double covolume, const double exponent)
    {
      return P.b == 0. ? 1. : dev_pow(covolume, exponent);
    }

    /* hyperbolic_system.h:1073-1088 */
    static RYUJIN_DEV double surrogate_specific_entropy(const Params &P, const double (&U)[K],
                                                        const double gamma_min)
    {
      const double rho = U[0];
      const double rho_inverse = 1. / rho;
      const double covolume = 1. - P.b * rho;
      const double shift = internal_energy(U) - rho * P.q - P.pinf * covolume;
      return shift * dev_pow(rho_inverse - P.b, gamma_min) / covolume;
    }

    /* hyperbolic_system.h:1093-1116 */
    static RYUJIN_DEV double surrogate_harten_entropy(const Params &P, const double (&U)[K],
                                                      const double gamma_min)
    {
      const double rho = U[0];
      const double E = U[1 + DIM];
      const double rho_rho_e_q = rho * E - 0.5 * momentum_norm_square(U) - rho * rho * P.q;
      const double exponent = 1. / (gamma_min + 1.);
      const double covolume = 1. - P.b * rho;
      const double covolume_term = covolume_pow(P, covolume, gamma_min - 1.);
      const double rho_pinfcov = rho * P.pinf * covolume;
      return dev_pow((rho_rho_e_q - rho_pinfcov) * covolume_term, exponent);
    }

    /* hyperbolic_system.h:1121-1176 */
    static RYUJIN_DEV void surrogate_harten_entropy_derivative(const Params &P, const double (&U)[K],
                                                               const double eta, const double gamma_min,
                                                               double (&result)[K])
    {
      const double rho = U[0];
      const double E = U[1 + DIM];
      const double covolume = 1. - P.b * rho;
      const double covolume_inverse = 1. / covolume;
      const double shift =
          rho * E - 0.5 * momentum_norm_square(U) - rho * rho * P.q - rho * P.pinf * covolume;
      const double factor = dev_pow(eta * covolume_inverse, -gamma_min) *
                            (covolume_inverse * covolume_inverse) / (gamma_min + 1.);
      const double first_term = E - 2. * rho * P.q - P.pinf * (1. - 2. * P.b * rho);
      const double second_term = -(gamma_min - 1.) * shift * P.b;
      result[0] = factor * (covolume * first_term + second_term);
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        result[1 + d] = -factor * covolume * U[1 + d];
      result[DIM + 1] = factor * covolume * rho;
    }

    /* f(U, p): hyperbolic_system.h:1382-1400 */
    static RYUJIN_DEV void flux(const double (&U)[K], const double p, double (&f)[K][DIM])
    {
      const double rho_inverse = 1. / U[0];
      const double E = U[1 + DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        f[0][d] = U[1 + d];
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const double s = U[1 + i] * rho_inverse;
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          f[1 + i][d] = U[1 + d] * s;
        f[1 + i][i] += p;
      }
      const double s = rho_inverse * (E + p);
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        f[DIM + 1][d] = U[1 + d] * s;
    }

    static RYUJIN_DEV void flux_divergence(const double (&fi)[K][DIM], const double (&fj)[K][DIM],
                                           const double (&c)[DIM], double (&out)[K])
    {
#pragma unroll
      for (int q = 0; q < K; ++q) {
        double s = (fi[q][0] + fj[q][0]) * c[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          s += (fi[q][d] + fj[q][d]) * c[d];
        out[q] = -s;
      }
    }

    /* hyperbolic_system.h:991-998 */
    static RYUJIN_DEV double filter_vacuum_density(const Params &P, const double rho)
    {
      const double rho_cutoff_large = P.reference_density * P.vacuum_large * DBL_EPSILON;
      return fabs(rho) < rho_cutoff_large ? 0. : rho;
    }

    /* precomputation cycle 0 (hyperbolic_system.h:919-938) */
    static RYUJIN_DEV Prec precompute_cycle0(const Params &P, const double (&U)[K])
    {
      const double rho_i = U[0];
      const double e_i = internal_energy(U) / rho_i;
      const double p_i = eos_pressure(P, rho_i, e_i);
      return {p_i, surrogate_gamma(P, U, p_i), 0., 0.};
    }

    /* Indicator: indicator.h:187-262 */
    struct Indicator {
      double rho_i_inverse, eta_i, gamma_min, left;
      double d_eta_i[K], f_i[K][DIM], right[K];

      RYUJIN_DEV void reset(const Params &P, const double (&U_i)[K], const Prec &prec_i)
      {
        gamma_min = prec_i.gamma_min;
        rho_i_inverse = 1. / U_i[0];
        eta_i = prec_i.eta;
        surrogate_harten_entropy_derivative(P, U_i, eta_i, gamma_min, d_eta_i);
        d_eta_i[0] -= eta_i * rho_i_inverse;
        const double surrogate_p_i = surrogate_pressure(P, U_i, gamma_min);
        flux(U_i, surrogate_p_i, f_i);
        left = 0.;
#pragma unroll
        for (int q = 0; q < K; ++q)
          right[q] = 0.;
      }

      RYUJIN_DEV void accumulate(const Params &P, const double (&U_j)[K], const double (&c_ij)[DIM])
      {
        const double eta_j = surrogate_harten_entropy(P, U_j, gamma_min);
        const double rho_j_inverse = 1. / U_j[0];
        const double surrogate_p_j = surrogate_pressure(P, U_j, gamma_min);
        double f_j[K][DIM];
        flux(U_j, surrogate_p_j, f_j);
        double m_j_c = U_j[1] * c_ij[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          m_j_c += U_j[1 + d] * c_ij[d];
        const double entropy_flux = (eta_j * rho_j_inverse - eta_i * rho_i_inverse) * m_j_c;
        left += entropy_flux;
#pragma unroll
        for (int q = 0; q < K; ++q) {
          double component = (f_j[q][0] - f_i[q][0]) * c_ij[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            component += (f_j[q][d] - f_i[q][d]) * c_ij[d];
          right[q] += component;
        }
      }

      RYUJIN_DEV double alpha(const Params &P, const double hd_i) const
      {
        double numerator = left;
        double denominator = fabs(left);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          numerator -= d_eta_i[q] * right[q];
          denominator += fabs(d_eta_i[q] * right[q]);
        }
        const double quotient = fabs(numerator) / (denominator + hd_i * fabs(eta_i));
        return fmin(1., P.evc_factor * quotient);
      }
    };

    /* ------------------------------------------------------------------ Riemann solver */

    /* alpha = rs_alpha(rho, gamma, a) (:39-50) and alpha_hat = rs_c(gamma) alpha (:21-36) depend on one state only:
     * carried along instead of being re-derived inside every pressure estimate */
    struct RiemannData {
      double rho, u, p, gamma, a, alpha, alpha_hat;
    };

    /* riemann_solver.template.h:21-36 */
    static RYUJIN_DEV double rs_c(const double gamma)
    {
      constexpr double slope = -0.34976871477801828189920753948709;
      const double first_radicand = (3. * gamma + 11.) / (6. * gamma + 6.);
      const double second_radicand = 5. / 6. + slope * (gamma - 3.);
      double radicand = fmin(first_radicand, second_radicand);
      radicand = fmin(1., radicand);
      radicand = fmax(1. / 2., radicand);
      return sqrt(radicand);
    }

    /* :39-50 */
    static RYUJIN_DEV double rs_alpha(const Params &P, const double rho, const double gamma, const double a)
    {
      const double numerator = 2. * a * (1. - P.b * rho);
      const double denominator = gamma - 1.;
      return numerator / denominator;
    }

    /* :395-440 */
    static RYUJIN_DEV RiemannData riemann_data_from_state(const Params &P, const double (&U)[K],
                                                          const double p, const double (&n)[DIM])
    {
      const double rho = U[0];
      const double rho_inverse = 1.0 / rho;
      double proj_m = n[0] * U[1];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        proj_m += n[d] * U[1 + d];
      const double gamma = surrogate_gamma(P, U, p);
      const double x = 1. - P.b * rho;
      const double a = sqrt(gamma * (p + P.pinf) / (rho * x));
      return make_riemann_data(P, rho, proj_m * rho_inverse, p, gamma, a);
    }

    static RYUJIN_DEV RiemannData make_riemann_data(const Params &P, const double rho, const double u,
                                                    const double p, const double gamma, const double a)
    {
      const double alpha = rs_alpha(P, rho, gamma, a);
      return {rho, u, p, gamma, a, alpha, rs_c(gamma) * alpha};
    }

    /* ---- per-node Riemann record (as Euler<DIM>::riemann_record): everything riemann_data_from_state derives
     * from one state and its EOS pressure except the normal velocity -- rho, p, gamma, a, alpha, alpha_hat -- and
     * the velocity vector; per pair this removes 2 x (3 divisions + 1 square root) of the state conversion and
     * the 2-3 x (division, square root, division) of rs_alpha / rs_c. The normal velocity becomes v . n instead
     * of (m . n) / rho: a few ulp, inside the 1e-12 contract on d_ij (pinned on random pairs and in every sweep
     * comparison, as for Euler). record = (rho, p, gamma, a, alpha, alpha_hat, v[DIM]) padded to even. */
    static constexpr int RS = (6 + DIM + 1) / 2 * 2;

    static RYUJIN_DEV void riemann_record(const Params &P, const double (&U)[K], const double p,
                                          double (&rec)[RS])
    {
      const double rho = U[0];
      const double rho_inverse = 1.0 / rho;
      const double gamma = surrogate_gamma(P, U, p);
      const double x = 1. - P.b * rho;
      const double a = sqrt(gamma * (p + P.pinf) / (rho * x));
      const double alpha = rs_alpha(P, rho, gamma, a);
      rec[0] = rho;
      rec[1] = p;
      rec[2] = gamma;
      rec[3] = a;
      rec[4] = alpha;
      rec[5] = rs_c(gamma) * alpha;
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        rec[6 + d] = U[1 + d] * rho_inverse;
#pragma unroll
      for (int d = 6 + DIM; d < RS; ++d)
        rec[d] = 0.;
    }

    static RYUJIN_DEV double dij_from_records(const Params &P, const double (&ri)[RS], const double (&rj)[RS],
                                              const double (&c)[DIM])
    {
      double norm2 = c[0] * c[0];
      double vc_i = ri[6] * c[0], vc_j = rj[6] * c[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        norm2 += c[d] * c[d];
        vc_i += ri[6 + d] * c[d];
        vc_j += rj[6 + d] * c[d];
      }
      const double norm = sqrt(norm2);
      const double inverse_norm = 1. / norm;
      const RiemannData rd_i{ri[0], vc_i * inverse_norm, ri[1], ri[2], ri[3], ri[4], ri[5]};
      const RiemannData rd_j{rj[0], vc_j * inverse_norm, rj[1], rj[2], rj[3], rj[4], rj[5]};
      return norm * riemann_compute(P, rd_i, rd_j);
    }

    /* :161-198 */
    static RYUJIN_DEV double p_star_failsafe(const Params &P, const RiemannData &i, const RiemannData &j)
    {
      const double p_max = fmax(i.p, j.p) + P.pinf;
      double radicand_i = 2. * (1. - P.b * i.rho) * p_max;
      radicand_i /= i.rho * ((i.gamma + 1.) * p_max + (i.gamma - 1.) * (i.p + P.pinf));
      const double x_i = sqrt(radicand_i);
      double radicand_j = 2. * (1. - P.b * j.rho) * p_max;
      radicand_j /= j.rho * ((j.gamma + 1.) * p_max + (j.gamma - 1.) * (j.p + P.pinf));
      const double x_j = sqrt(radicand_j);
      const double a = x_i + x_j;
      const double b = j.u - i.u;
      const double c = -(i.p + P.pinf) * x_i - (j.p + P.pinf) * x_j;
      const double base = (-b + sqrt(b * b - 4. * a * c)) / (2. * a);
      return base * base - P.pinf;
    }

    /* :53-120 */
    static RYUJIN_DEV double p_star_RS_full(const Params &P, const RiemannData &i, const RiemannData &j)
    {
      const double alpha_i = i.alpha;
      const double alpha_j = j.alpha;
      const double p_min = fmin(i.p, j.p);
      const double p_max = fmax(i.p, j.p);
      const double gamma_min = i.p < j.p ? i.gamma : j.gamma;
      const double alpha_hat_min = i.p < j.p ? i.alpha_hat : j.alpha_hat; /* rs_c(gamma_min) alpha_min */
      const double alpha_max = i.p >= j.p ? alpha_i : alpha_j;
      const double gamma_m = fmin(i.gamma, j.gamma);
      const double gamma_M = fmax(i.gamma, j.gamma);
      const double numerator = positive_part(alpha_hat_min + alpha_max - (j.u - i.u));
      const double p_ratio = (p_min + P.pinf) / (p_max + P.pinf);
      const double r_exponent = (gamma_M - gamma_min) / (2. * gamma_min * gamma_M);
      const double first_exponent = (gamma_M - 1.) / (2. * gamma_M);
      const double first_exponent_inverse = 1. / first_exponent;
      const double first_denom =
          alpha_hat_min * dev_pow(p_ratio, r_exponent - first_exponent) + alpha_max;
      const double p_1_tilde =
          (p_max + P.pinf) * dev_pow(numerator / first_denom, first_exponent_inverse) - P.pinf;
      const double second_exponent = (gamma_m - 1.) / (2. * gamma_m);
      const double second_exponent_inverse = 1. / second_exponent;
      const double second_denom = alpha_hat_min * dev_pow(p_ratio, -second_exponent) +
                                  alpha_max * dev_pow(p_ratio, r_exponent);
      const double p_2_tilde =
          (p_max + P.pinf) * dev_pow(numerator / second_denom, second_exponent_inverse) - P.pinf;
      return fmin(p_1_tilde, p_2_tilde);
    }

    /* :123-158 */
    static RYUJIN_DEV double p_star_SS_full(const Params &P, const RiemannData &i, const RiemannData &j)
    {
      const double gamma_m = fmin(i.gamma, j.gamma);
      const double alpha_hat_i = i.alpha_hat;
      const double alpha_hat_j = j.alpha_hat;
      const double exponent = (gamma_m - 1.) / (2. * gamma_m);
      const double exponent_inverse = 1. / exponent;
      const double numerator = positive_part(alpha_hat_i + alpha_hat_j - (j.u - i.u));
      const double denominator =
          alpha_hat_i * dev_pow((i.p + P.pinf) / (j.p + P.pinf), -exponent) + alpha_hat_j;
      const double p_1_tilde =
          (j.p + P.pinf) * dev_pow(numerator / denominator, exponent_inverse) - P.pinf;
      const double p_2_tilde = p_star_failsafe(P, i, j);
      return fmin(p_1_tilde, p_2_tilde);
    }

    /* :201-255 */
    static RYUJIN_DEV double p_star_interpolated(const Params &P, const RiemannData &i, const RiemannData &j)
    {
      const double alpha_i = i.alpha;
      const double alpha_j = j.alpha;
      const double p_min = fmin(i.p, j.p) + P.pinf;
      const double p_max = fmax(i.p, j.p) + P.pinf;
      const double gamma_min = i.p < j.p ? i.gamma : j.gamma;
      const double alpha_hat_min = i.p < j.p ? i.alpha_hat : j.alpha_hat; /* rs_c(gamma_min) alpha_min */
      const double alpha_max = i.p >= j.p ? alpha_i : alpha_j;
      const double alpha_hat_max = i.p >= j.p ? i.alpha_hat : j.alpha_hat; /* rs_c(gamma_max) alpha_max */
      const double gamma_m = fmin(i.gamma, j.gamma);
      const double gamma_M = fmax(i.gamma, j.gamma);
      const double p_ratio = p_min / p_max;
      const double r_exponent = (gamma_M - gamma_min) / (2. * gamma_min * gamma_M);
      const double exponent = (gamma_m - 1.) / (2. * gamma_m);
      const double exponent_inverse = 1. / exponent;
      const double numerator = positive_part(alpha_hat_min + /*SIC!*/ alpha_max - (j.u - i.u));
      const double denominator = alpha_hat_min * dev_pow(p_ratio, -exponent) +
                                 alpha_hat_max * dev_pow(p_ratio, r_exponent);
      return p_max * dev_pow(numerator / denominator, exponent_inverse) - P.pinf;
    }

    /* :307-339 */
    static RYUJIN_DEV double phi_of_p_max(const Params &P, const RiemannData &i, const RiemannData &j)
    {
      const double p_max = fmax(i.p, j.p) + P.pinf;
      const double radicand_inverse_i = 0.5 * i.rho / (1. - P.b * i.rho) *
                                        ((i.gamma + 1.) * p_max + (i.gamma - 1.) * (i.p + P.pinf));
      const double value_i = (p_max - i.p) / sqrt(radicand_inverse_i);
      const double radicand_inverse_j = 0.5 * j.rho / (1. - P.b * j.rho) *
                                        ((j.gamma + 1.) * p_max + (j.gamma - 1.) * (j.p + P.pinf));
      const double value_j = (p_max - j.p) / sqrt(radicand_inverse_j);
      return value_i + value_j + j.u - i.u;
    }

    /* :342-392 */
    static RYUJIN_DEV double compute_lambda(const Params &P, const RiemannData &i, const RiemannData &j,
                                            const double p_star)
    {
      const double factor_i = 0.5 * (i.gamma + 1.) / i.gamma;
      const double tmp_i = positive_part((p_star - i.p) / (i.p + P.pinf));
      const double nu_11 = i.u - i.a * sqrt(1. + factor_i * tmp_i);
      const double factor_j = 0.5 * (j.gamma + 1.) / j.gamma;
      const double tmp_j = positive_part((p_star - j.p) / (j.p + P.pinf));
      const double nu_32 = j.u + j.a * sqrt(1. + factor_j * tmp_j);
      return fmax(positive_part(nu_32), negative_part(nu_11));
    }

    /* :443-560 */
    static RYUJIN_DEV double riemann_compute(const Params &P, const RiemannData &i, const RiemannData &j)
    {
      const double p_max = fmax(i.p, j.p) + P.pinf;
      const double phi_p_max = phi_of_p_max(P, i, j);
      if (!P.strict) {
        const double p_star_tilde = p_star_interpolated(P, i, j);
        const double p_star_backup = p_star_failsafe(P, i, j);
        const double p_2 =
            phi_p_max < 0. ? fmin(p_star_tilde, p_star_backup) : fmin(p_max, p_star_tilde);
        return compute_lambda(P, i, j, p_2);
      }
      /* The reference evaluates both estimates (seven powers) and selects by the sign of phi(p_max). Where the
       * sign is the same over the wave only the selected estimate is evaluated: same value, two or five powers less. */
      const bool two_shocks = phi_p_max < 0.;
      double p_2;
      if (!__any(!two_shocks)) {
        p_2 = p_star_SS_full(P, i, j);
      } else if (!__any(two_shocks)) {
        p_2 = fmin(p_max, p_star_RS_full(P, i, j));
      } else {
        const double p_star_RS = p_star_RS_full(P, i, j);
        const double p_star_SS = p_star_SS_full(P, i, j);
        p_2 = two_shocks ? p_star_SS : fmin(p_max, p_star_RS);
      }
      return compute_lambda(P, i, j, p_2);
    }

    /* d_ij = |c| lambda_max(U_i, U_j, c/|c|) with the precomputed EOS pressures (:563-582) */
    static RYUJIN_DEV double dij_from_states(const Params &P, const double (&U_i)[K], const double p_i,
                                             const double (&U_j)[K], const double p_j,
                                             const double (&c)[DIM])
    {
      double norm2 = c[0] * c[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        norm2 += c[d] * c[d];
      const double norm = sqrt(norm2);
      double n[DIM];
      const double inverse_norm = 1. / norm; /* dealii::Tensor / scalar multiplies by the inverse */
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        n[d] = c[d] * inverse_norm;
      const RiemannData rd_i = riemann_data_from_state(P, U_i, p_i, n);
      const RiemannData rd_j = riemann_data_from_state(P, U_j, p_j, n);
      return norm * riemann_compute(P, rd_i, rd_j);
    }

    /* ------------------------------------------------------------------ Limiter::limit */

    /* psi(U) = relax_small rho (rho e - rho q - pinf (1 - b rho)) - s_min rho rho^gamma (1 - b rho)^(1-gamma)
     * (limiter.template.h:190-200); also returns the pieces the Newton step needs */
    struct Psi {
      double psi, rho, rho_gamma, covolume, rho_e;
    };
    static RYUJIN_DEV Psi psi_of(const Params &P, const double (&V)[K], const double s_min,
                                 const double gamma, const double relax_small)
    {
      Psi r;
      r.rho = V[0];
      r.rho_gamma = dev_pow(r.rho, gamma);
      r.covolume = 1. - P.b * r.rho;
      r.rho_e = internal_energy(V);
      const double shift = r.rho_e - r.rho * P.q - P.pinf * r.covolume;
      r.psi = relax_small * r.rho * shift -
              s_min * r.rho * r.rho_gamma * covolume_pow(P, r.covolume, -(gamma - 1.));
      return r;
    }

    static RYUJIN_DEV double density_clip(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                          const double (&Pij)[K], bool &success)
    {
      const double rho_min = bnd[0], rho_max = bnd[1];
      constexpr double t_min = 0., t_max = 1.;
      constexpr double eps = DBL_EPSILON;
      const double relax = 1. + P.vacuum_large * eps;
      double t_r = t_max;
      const double rho_U = U[0];
      const double rho_P = Pij[0];
      const double test_min = filter_vacuum_density(P, fmax(0., rho_U - relax * rho_max));
      const double test_max = filter_vacuum_density(P, fmax(0., rho_min - relax * rho_U));
      if (!(test_min == 0. && test_max == 0.))
        success = false;
      /* (the division only where a lane clips, as Euler<DIM>::first_psi_r: t_r = t_max stays untouched otherwise) */
      if (rho_max < rho_U + t_r * rho_P || rho_U + t_r * rho_P < rho_min) {
        const double denominator = 1. / (fabs(rho_P) + eps * rho_max);
        t_r = rho_max < rho_U + t_r * rho_P ? (rho_max - rho_U) * denominator : t_r;
        t_r = rho_U + t_r * rho_P < rho_min ? (rho_U - rho_min) * denominator : t_r;
        t_r = fmin(t_r, t_max);
        t_r = fmax(t_r, t_min);
      }
      return t_r;
    }

    /* wave-uniform fast part, see Euler<DIM>::limit_fast */
    static RYUJIN_DEV double limit_fast(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                        const double (&Pij)[K], bool &success, bool &undecided)
    {
      constexpr double t_min = 0.;
      success = true;
      undecided = false;
      const double t_r = density_clip(P, bnd, U, Pij, success);
      if (P.lim_newton_max_iterations <= 0)
        return t_min;
      const double relax_small = 1. + P.vacuum_small * DBL_EPSILON;
      double U_r[K];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_r[q] = U[q] + t_r * Pij[q];
      const Psi r = psi_of(P, U_r, bnd[2], bnd[3], relax_small);
      if (r.psi > 0.)
        return t_r;
      if (t_r == t_min)
        return t_min;
      undecided = true;
      return t_min;
    }

    static RYUJIN_DEV double internal_energy_derivative_dot(const double (&V)[K], const double (&Pij)[K])
    {
      const double rho_inverse = 1. / V[0];
      double u[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        u[d] = V[1 + d] * rho_inverse;
      double u2 = u[0] * u[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        u2 += u[d] * u[d];
      double s = (0.5 * u2) * Pij[0];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        s += (-u[d]) * Pij[1 + d];
      s += 1. * Pij[1 + DIM];
      return s;
    }

    /* limiter.template.h:15-360, production control flow */
    static RYUJIN_DEV double limit(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                   const double (&Pij)[K], bool &success)
    {
      constexpr double t_min = 0.;
      constexpr double eps = DBL_EPSILON;
      success = true;
      double t_r = density_clip(P, bnd, U, Pij, success);
      const double relax_small = 1. + P.vacuum_small * eps;
      const double relax = 1. + P.vacuum_large * eps;
      double t_l = t_min;
      const double s_min = bnd[2];
      const double gamma = bnd[3];
      const double gm1 = gamma - 1.;

      for (int n = 0; n < P.lim_newton_max_iterations; ++n) {
        double U_r[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_r[q] = U[q] + t_r * Pij[q];
        const Psi R = psi_of(P, U_r, s_min, gamma, relax_small);

        t_l = R.psi > 0. ? t_r : t_l;
        if (t_l == t_r)
          break;

        double U_l[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_l[q] = U[q] + t_l * Pij[q];
        const Psi L = psi_of(P, U_l, s_min, gamma, relax_small);

        const double lower_bound =
            (1. - relax) * s_min * L.rho * L.rho_gamma * covolume_pow(P, L.covolume, -gm1);
        if (n == 0 && !(fmin(0., L.psi - lower_bound) == 0.))
          success = false;

        if (fmax(0., t_r - t_l - P.lim_newton_tolerance) == 0.)
          break;

        const double drho = Pij[0];
        const double drho_e_l = internal_energy_derivative_dot(U_l, Pij);
        const double drho_e_r = internal_energy_derivative_dot(U_r, Pij);
        const double q_pinf_term_l = 2. * L.rho * P.q + P.pinf * (1. - 2. * P.b * L.rho);
        const double q_pinf_term_r = 2. * R.rho * P.q + P.pinf * (1. - 2. * P.b * R.rho);
        const double extra_term_l =
            s_min * dev_pow(L.rho / L.covolume, gamma) * (L.covolume + gamma - P.b * L.rho);
        const double extra_term_r =
            s_min * dev_pow(R.rho / R.covolume, gamma) * (R.covolume + gamma - P.b * R.rho);
        const double dpsi_l = L.rho * drho_e_l + (L.rho_e - q_pinf_term_l - extra_term_l) * drho;
        const double dpsi_r = R.rho * drho_e_r + (R.rho_e - q_pinf_term_r - extra_term_r) * drho;
        double psi_l = L.psi, psi_r = R.psi;
        quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
      }
      return t_l;
    }

    /* View::is_admissible (euler_aeos/hyperbolic_system.h): rho > 0 and rho e - rho q - pinf (1 - b rho) > 0 */
    static RYUJIN_DEV bool is_admissible(const Params &P, const double (&U)[K])
    {
      const double rho = U[0];
      const double shift = internal_energy(U) - rho * P.q - P.pinf * (1. - P.b * rho);
      return rho > 0. && shift > 0.;
    }

    /* limiter.template.h in the EXPENSIVE_BOUNDS_CHECK control flow (ryujin_hip_params::debug_expensive_bounds_check):
     * the density behind the clip, no "psi_r > 0" shortcut in front of psi_l, the final check of the limited state.
     * The same t_l as limit(); `success` has more ways to be false. Debug kernel only (k_check_limiter). */
    static RYUJIN_DEV double limit_checked(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                           const double (&Pij)[K], bool &success)
    {
      constexpr double t_min = 0., t_max = 1.;
      constexpr double eps = DBL_EPSILON;
      success = true;
      const double rho_min = bnd[0], rho_max = bnd[1];
      const double relax_small = 1. + P.vacuum_small * eps;
      const double relax = 1. + P.vacuum_large * eps;
      double t_r = t_max;
      {
        const double rho_U = U[0];
        const double rho_P = Pij[0];
        const double test_min = filter_vacuum_density(P, fmax(0., rho_U - relax * rho_max));
        const double test_max = filter_vacuum_density(P, fmax(0., rho_min - relax * rho_U));
        if (!(test_min == 0. && test_max == 0.))
          success = false;
        const double denominator = 1. / (fabs(rho_P) + eps * rho_max);
        t_r = rho_max < rho_U + t_r * rho_P ? (rho_max - rho_U) * denominator : t_r;
        t_r = rho_U + t_r * rho_P < rho_min ? (rho_U - rho_min) * denominator : t_r;
        t_r = fmin(t_r, t_max);
        t_r = fmax(t_r, t_min);
        const double rho_new = U[0] + t_r * Pij[0];
        const double test_new_min = filter_vacuum_density(P, fmax(0., rho_new - relax * rho_max));
        const double test_new_max = filter_vacuum_density(P, fmax(0., rho_min - relax * rho_new));
        if (!(test_new_min == 0. && test_new_max == 0.))
          success = false;
      }
      double t_l = t_min;
      const double s_min = bnd[2];
      const double gamma = bnd[3];
      const double gm1 = gamma - 1.;
      for (int n = 0; n < P.lim_newton_max_iterations; ++n) {
        double U_r[K], U_l[K];
#pragma unroll
        for (int q = 0; q < K; ++q) {
          U_r[q] = U[q] + t_r * Pij[q];
          U_l[q] = U[q] + t_l * Pij[q];
        }
        const Psi R = psi_of(P, U_r, s_min, gamma, relax_small);
        const Psi L = psi_of(P, U_l, s_min, gamma, relax_small);
        const double lower_bound =
            (1. - relax) * s_min * L.rho * L.rho_gamma * covolume_pow(P, L.covolume, -gm1);
        if (n == 0 && !(fmin(0., L.psi - lower_bound) == 0.))
          success = false;
        t_l = R.psi > 0. ? t_r : t_l;
        if (fmax(0., t_r - t_l - P.lim_newton_tolerance) == 0.)
          break;
        const double drho = Pij[0];
        const double drho_e_l = internal_energy_derivative_dot(U_l, Pij);
        const double drho_e_r = internal_energy_derivative_dot(U_r, Pij);
        const double q_pinf_term_l = 2. * L.rho * P.q + P.pinf * (1. - 2. * P.b * L.rho);
        const double q_pinf_term_r = 2. * R.rho * P.q + P.pinf * (1. - 2. * P.b * R.rho);
        const double extra_term_l =
            s_min * dev_pow(L.rho / L.covolume, gamma) * (L.covolume + gamma - P.b * L.rho);
        const double extra_term_r =
            s_min * dev_pow(R.rho / R.covolume, gamma) * (R.covolume + gamma - P.b * R.rho);
        const double dpsi_l = L.rho * drho_e_l + (L.rho_e - q_pinf_term_l - extra_term_l) * drho;
        const double dpsi_r = R.rho * drho_e_r + (R.rho_e - q_pinf_term_r - extra_term_r) * drho;
        double psi_l = L.psi, psi_r = R.psi;
        quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
      }
      {
        double U_new[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_new[q] = U[q] + t_l * Pij[q];
        const Psi N = psi_of(P, U_new, s_min, gamma, relax_small);
        const double shift_new = N.rho_e - N.rho * P.q - P.pinf * N.covolume;
        const double lower_bound =
            (1. - relax) * s_min * N.rho * N.rho_gamma * covolume_pow(P, N.covolume, -gm1);
        const bool e_valid = fmin(0., shift_new) == 0.;
        const bool psi_valid = fmin(0., N.psi - lower_bound) == 0.;
        if (!e_valid || !psi_valid)
          success = false;
      }
      return t_l;
    }

    /* hyperbolic_system.h:1314-1377. `dynamic` is __builtin_trap() in the reference; create() rejects it. */
    static RYUJIN_DEV void apply_boundary_conditions(const Params &, const int id, const double (&U)[K],
                                                     const double (&normal)[DIM], const double (&U_D)[K],
                                                     double (&result)[K])
    {
#pragma unroll
      for (int q = 0; q < K; ++q)
        result[q] = U[q];
      if (id == RYUJIN_BC_DIRICHLET) {
#pragma unroll
        for (int q = 0; q < K; ++q)
          result[q] = U_D[q];
      } else if (id == RYUJIN_BC_SLIP) {
        double mn = U[1] * normal[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          mn += U[1 + d] * normal[d];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          result[1 + d] = U[1 + d] - 1. * mn * normal[d];
      } else if (id == RYUJIN_BC_NO_SLIP) {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          result[1 + d] = 0.;
      }
    }
  };
} // namespace ryujin_hip

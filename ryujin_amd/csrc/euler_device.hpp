// Device-side Euler "Description": pointwise arithmetic inlined into the sweep kernels.
//
// Restates (operation order preserved, so that results agree with the reference's scalar
// path to the last bits up to pow()):
//   HyperbolicSystemView  source/euler/hyperbolic_system.h:750-1216
//   RiemannSolver         source/euler/riemann_solver.template.h:21-597
//   Indicator             source/euler/indicator.h:187-258
//   Limiter               source/euler/limiter.h:255-363, limiter.template.h:15-327
//   quadratic_newton_step source/newton.h:37-101
// All loops run over compile-time bounds and are fully unrolled: states live in registers.

#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>


#ifndef RYUJIN_LIMIT_GUARD_CLIP
#define RYUJIN_LIMIT_GUARD_CLIP 1 /* the limiter's density clip (one FP64 division) only where a lane needs it */
#endif

namespace ryujin_hip
{
  struct EulerParams {
    double gamma, gamma_inverse, gamma_plus_one_inverse, gamma_minus_one_inverse;
    double reference_density, vacuum_small, vacuum_large;
    double evc_factor;
    double lim_newton_tolerance, lim_relaxation_factor;
    int lim_newton_max_iterations;
    int riemann_newton_max_iterations;
    double riemann_newton_tolerance;
    /* 2 gamma / (gamma - 1), the exponent of p_star_two_rarefaction, when it is a small integer
     * (gamma = 7/5: 7, 5/3: 5, 2: 4, 3: 3): evaluated by multiplications; 0: general pow */
    int rarefaction_power;
  };

#define RYUJIN_DEV __device__ __forceinline__


  /*
   * ryujin::pow (source/simd.template.h:196-272) on the device.
   *
   * ocml's pow() costs ~230 VALU instructions (special cases + extended-precision log); the hot
   * path only ever raises positive, finite, normal numbers to a real exponent, so we evaluate
   * x^y = exp(y * log x) directly:
   *   log x = e ln2 + 2 atanh(s),  s = (m-1)/(m+1),  x = m 2^e,  m in [sqrt(1/2), sqrt(2))
   *           (odd series in s up to s^21: truncation 8e-18),
   *   exp z = 2^n exp(r),  r = z - n ln2 (ln2 split hi/lo), |r| <= 0.35, Taylor to r^13 (2e-16
   *           relative truncation at the interval ends, far below the y*log(x) rounding).
   * Error: a few ulp for |y log x| <~ 10 (dominated by the rounding of the product y*log x), i.e.
   * ~1e-15 relative -- two orders of magnitude inside the function-level parity tolerance
   * (the reference's own pow variants already differ in the last digits).
   * Everything else (x <= 0, subnormal, inf, nan) is forwarded to ocml's pow().
   */
  /* Self-contained: the special cases (zero, negative base, inf, nan, subnormal, overflow) are resolved with a
   * handful of selects instead of forwarding to ocml's pow(). Inlined ocml pow() costs every kernel that calls
   * pow its register high-water mark (~100 VGPRs on top of the state live at the call site), although the
   * hot path never takes it. */
  RYUJIN_DEV double dev_pow(const double x, const double y)
  {
    double ax = fabs(x);
    long long bits = __double_as_longlong(ax);
    int biased = (int)((bits >> 52) & 0x7ff);
    int e = -1023;
    if (__builtin_expect(biased == 0, 0)) { /* subnormal (or zero, resolved below): renormalise */
      ax *= 18014398509481984.; /* 2^54 */
      bits = __double_as_longlong(ax);
      biased = (int)((bits >> 52) & 0x7ff);
      e = -1023 - 54;
    }
    e += biased;
    /* ax = m * 2^e, m in [sqrt(1/2), sqrt(2)) */
    double m = __longlong_as_double((bits & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
    if (m > 1.4142135623730951) {
      m *= 0.5;
      e += 1;
    }
    const double s = (m - 1.) / (m + 1.);
    const double s2 = s * s;
    double p = 2. / 21.;
    p = __builtin_fma(p, s2, 2. / 19.);
    p = __builtin_fma(p, s2, 2. / 17.);
    p = __builtin_fma(p, s2, 2. / 15.);
    p = __builtin_fma(p, s2, 2. / 13.);
    p = __builtin_fma(p, s2, 2. / 11.);
    p = __builtin_fma(p, s2, 2. / 9.);
    p = __builtin_fma(p, s2, 2. / 7.);
    p = __builtin_fma(p, s2, 2. / 5.);
    p = __builtin_fma(p, s2, 2. / 3.);
    /* log m = 2s + s^3 p */
    const double log_m = __builtin_fma(s * s2, p, 2. * s);
    constexpr double ln2_hi = 6.93147180369123816490e-01; /* 0x3fe62e42fee00000 */
    constexpr double ln2_lo = 1.90821492927058770002e-10; /* 0x3dea39ef35793c76 */
    const double ed = (double)e;
    const double log_x = __builtin_fma(ed, ln2_hi, __builtin_fma(ed, ln2_lo, log_m));

    /* beyond +-800 the result is inf / 0 anyway; the clamp keeps n inside the range of ldexp */
    const double z = fmin(fmax(y * log_x, -800.), 800.);
    const double n = __builtin_rint(z * 1.44269504088896338700e+00);
    double r = __builtin_fma(-n, ln2_hi, z);
    r = __builtin_fma(-n, ln2_lo, r);
    double q = 1. / 6227020800.;               /* 1/13! */
    q = __builtin_fma(q, r, 1. / 479001600.);  /* 1/12! */
    q = __builtin_fma(q, r, 1. / 39916800.);
    q = __builtin_fma(q, r, 1. / 3628800.);
    q = __builtin_fma(q, r, 1. / 362880.);
    q = __builtin_fma(q, r, 1. / 40320.);
    q = __builtin_fma(q, r, 1. / 5040.);
    q = __builtin_fma(q, r, 1. / 720.);
    q = __builtin_fma(q, r, 1. / 120.);
    q = __builtin_fma(q, r, 1. / 24.);
    q = __builtin_fma(q, r, 1. / 6.);
    q = __builtin_fma(q, r, 0.5);
    q = __builtin_fma(q, r, 1.);
    q = __builtin_fma(q, r, 1.);
    double result = ldexp(q, (int)n);

    /* special cases as std::pow */
    if (__builtin_expect(!(ax > 0.) || biased == 0x7ff || x < 0., 0)) {
      if (ax == 0.)
        result = y > 0. ? 0. : __builtin_inf();
      else if (ax == __builtin_inf())
        result = y > 0. ? __builtin_inf() : 0.;
      if (x < 0.) {
        const double half = 0.5 * y;
        if (y != __builtin_rint(y))
          result = __builtin_nan("");
        else if (half != __builtin_rint(half))
          result = -result; /* odd integer exponent */
      }
      if (x != x || y != y)
        result = __builtin_nan("");
    }
    if (__builtin_expect(y == 0. || x == 1., 0))
      result = 1.;
    return result;
  }
  RYUJIN_DEV double positive_part(double x) { return fmax(0., x); }
  RYUJIN_DEV double negative_part(double x) { return -fmin(0., x); }

  /* source/newton.h:37-101 */
  RYUJIN_DEV void quadratic_newton_step(double &p_1, double &p_2, const double phi_p_1,
                                        const double phi_p_2, const double dphi_p_1,
                                        const double dphi_p_2, const double sign)
  {
    constexpr double eps = DBL_EPSILON;
    const double scaling = 1. / (p_2 - p_1 + eps);

    const double dd_11 = dphi_p_1;
    const double dd_12 = (phi_p_2 - phi_p_1) * scaling;
    const double dd_22 = dphi_p_2;

    const double dd_112 = (dd_12 - dd_11) * scaling;
    const double dd_122 = (dd_22 - dd_12) * scaling;

    const double discriminant_1 = fabs(dphi_p_1 * dphi_p_1 - 4. * phi_p_1 * dd_112);
    const double discriminant_2 = fabs(dphi_p_2 * dphi_p_2 - 4. * phi_p_2 * dd_122);

    const double denominator_1 = dphi_p_1 + sign * sqrt(discriminant_1);
    const double denominator_2 = dphi_p_2 + sign * sqrt(discriminant_2);

    double t_1 = p_1 - (fabs(denominator_1) < eps ? 0. : 2. * phi_p_1 / denominator_1);
    double t_2 = p_2 - (fabs(denominator_2) < eps ? 0. : 2. * phi_p_2 / denominator_2);

    t_1 = fmax(p_1, t_1);
    t_1 = fmin(p_2, t_1);
    t_2 = fmax(p_1, t_2);
    t_2 = fmin(p_2, t_2);

    p_1 = fmin(t_1, t_2);
    p_2 = fmax(t_1, t_2);
  }


  template <int DIM>
  struct Euler {
    static constexpr int DIMENSION = DIM;
    static constexpr int K = DIM + 2;
    static constexpr int NB = 3;
    using Params = EulerParams;

    /* hyperbolic_system.h:783-792 */
    static RYUJIN_DEV double internal_energy(const double (&U)[K])
    {
      const double rho_inverse = 1. / U[0];
      double m2 = U[1] * U[1];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        m2 += U[1 + d] * U[1 + d];
      return U[1 + DIM] - 0.5 * m2 * rho_inverse;
    }

    static RYUJIN_DEV double momentum_norm_square(const double (&U)[K])
    {
      double m2 = U[1] * U[1];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        m2 += U[1 + d] * U[1 + d];
      return m2;
    }

    /* hyperbolic_system.h:844-850 */
    static RYUJIN_DEV double specific_entropy(const EulerParams &P, const double (&U)[K])
    {
      const double rho_inverse = 1. / U[0];
      return internal_energy(U) * dev_pow(rho_inverse, P.gamma);
    }

    /* hyperbolic_system.h:855-865 */
    static RYUJIN_DEV double harten_entropy(const EulerParams &P, const double (&U)[K])
    {
      const double rho_rho_e = U[0] * U[1 + DIM] - 0.5 * momentum_norm_square(U);
      return dev_pow(rho_rho_e, P.gamma_plus_one_inverse);
    }

    /* hyperbolic_system.h:870-902 */
    static RYUJIN_DEV void harten_entropy_derivative(const EulerParams &P, const double (&U)[K],
                                                     double (&result)[K])
    {
      const double rho = U[0];
      const double E = U[1 + DIM];
      const double rho_rho_e = rho * E - 0.5 * momentum_norm_square(U);
      const double factor =
          P.gamma_plus_one_inverse * dev_pow(rho_rho_e, -P.gamma * P.gamma_plus_one_inverse);
      result[0] = factor * E;
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        result[1 + d] = -factor * U[1 + d];
      result[DIM + 1] = factor * rho;
    }

    /* f(U): hyperbolic_system.h:1164-1181 */
    static RYUJIN_DEV void flux(const EulerParams &P, const double (&U)[K], double (&f)[K][DIM])
    {
      const double rho_inverse = 1. / U[0];
      const double p = (P.gamma - 1.) * internal_energy(U);
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

    /* -contract(add(flux_i, flux_j), c_ij): hyperbolic_system.h:1208-1216 */
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

    /* hyperbolic_system.h:750-759 */
    static RYUJIN_DEV double filter_vacuum_density(const EulerParams &P, const double rho)
    {
      const double rho_cutoff_large = P.reference_density * P.vacuum_large * DBL_EPSILON;
      return fabs(rho) < rho_cutoff_large ? 0. : rho;
    }

    /* precomputed values (s_i, eta_i): hyperbolic_system.h:702-737 */
    static RYUJIN_DEV double2 precompute(const EulerParams &P, const double (&U)[K])
    {
      double2 out;
      out.x = specific_entropy(P, U);
      out.y = harten_entropy(P, U);
      return out;
    }

    /* the row's own column contributes exact zeros to the indicator sums (eta_j / rho_j - eta_i / rho_i and
     * f_j - f_i vanish bit for bit for j = i): the sweep skips it */
    static constexpr bool kIndicatorDiagonalIsZero = true;
    /* precompute() and riemann_record() are functions of the row's state alone: the last sweep of a step can
     * leave them behind for the next prepare_state_vector() (FusedPrecompute) */
    static constexpr bool kFusablePrecompute = true;
    /* steps 6/7 may form a limited row's update as V_i - sum (1 - l_ij) lambda P_ij (kernels_limiter.hpp): another
     * rounding of the reference's sum. Not where l = 0 has to return the low-order update EXACTLY (a dry node) */
    static constexpr bool kLimitedUpdateFromV = true;

    /* Indicator (entropy-viscosity commutator): indicator.h:187-258 */
    struct Indicator {
      double rho_i_inverse, eta_i, left;
      double d_eta_i[K], f_i[K][DIM], right[K];

      RYUJIN_DEV void reset(const EulerParams &P, const double (&U_i)[K], const double2 prec_i)
      {
        rho_i_inverse = 1. / U_i[0];
        eta_i = prec_i.y;
        harten_entropy_derivative(P, U_i, d_eta_i);
        d_eta_i[0] -= eta_i * rho_i_inverse;
        flux(P, U_i, f_i);
        left = 0.;
#pragma unroll
        for (int q = 0; q < K; ++q)
          right[q] = 0.;
      }

      RYUJIN_DEV void accumulate(const EulerParams &P, const double (&U_j)[K], const double2 prec_j,
                                 const double (&c_ij)[DIM])
      {
        const double eta_j = prec_j.y;
        const double rho_j_inverse = 1. / U_j[0];
        double f_j[K][DIM];
        flux(P, U_j, f_j);
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

      /* the same two from the per-node records (below): the record holds (rho, m, E), eta, 1 / rho and p of its
       * node, computed by the expressions flux() and accumulate() use -- the same bits without a division or an
       * internal-energy evaluation per neighbour, and one gather per neighbour instead of three */
      template <int RS_>
      RYUJIN_DEV void reset_record(const EulerParams &P, const double (&rec)[RS_])
      {
        double U_i[K];
        Euler::state_of_record(rec, U_i);
        rho_i_inverse = rec[kRecRinv];
        eta_i = rec[kRecEta];
        harten_entropy_derivative(P, U_i, d_eta_i);
        d_eta_i[0] -= eta_i * rho_i_inverse;
        Euler::flux_of_record(rec, f_i);
        left = 0.;
#pragma unroll
        for (int q = 0; q < K; ++q)
          right[q] = 0.;
      }

      template <int RS_>
      RYUJIN_DEV void accumulate_record(const double (&rec_j)[RS_], const double (&c_ij)[DIM])
      {
        const double eta_j = rec_j[kRecEta];
        const double rho_j_inverse = rec_j[kRecRinv];
        double f_j[K][DIM];
        Euler::flux_of_record(rec_j, f_j);
        double m_j_c = rec_j[kRecM] * c_ij[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          m_j_c += rec_j[kRecM + d] * c_ij[d];
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

      RYUJIN_DEV double alpha(const EulerParams &P, const double hd_i) const
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

    struct RiemannData {
      double rho, u, p, a;
    };

    /* riemann_solver.template.h:377-403 */
    static RYUJIN_DEV RiemannData riemann_data_from_state(const EulerParams &P, const double (&U)[K],
                                                          const double (&n)[DIM])
    {
      const double rho = U[0];
      const double rho_inverse = 1.0 / rho;
      double proj_m = n[0] * U[1];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        proj_m += n[d] * U[1 + d];
      double perp2;
      {
        const double perp = U[1] - proj_m * n[0];
        perp2 = perp * perp;
      }
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        const double perp = U[1 + d] - proj_m * n[d];
        perp2 += perp * perp;
      }
      const double E = U[1 + DIM] - 0.5 * perp2 * rho_inverse;
      const double rho_e = E - 0.5 * (proj_m * proj_m) * (1. / rho);
      const double p = (P.gamma - 1.) * rho_e;
      const double a = sqrt(P.gamma * p * (1. / rho));
      return {rho, proj_m * rho_inverse, p, a};
    }

    /* :21-46 */
    static RYUJIN_DEV double rs_f(const EulerParams &P, const RiemannData &rd, const double p_star)
    {
      const double Az = 2. / (rd.rho * (P.gamma + 1.));
      const double Bz = (P.gamma - 1.) / (P.gamma + 1.) * rd.p;
      const double radicand = Az / (p_star + Bz);
      const double true_value = (p_star - rd.p) * sqrt(radicand);
      const double exponent = 0.5 * (P.gamma - 1.) / P.gamma;
      const double factor = dev_pow(p_star / rd.p, exponent) - 1.;
      const double false_value = 2. * rd.a * factor / (P.gamma - 1.);
      return p_star >= rd.p ? true_value : false_value;
    }

    /* :49-84 */
    static RYUJIN_DEV double rs_df(const EulerParams &P, const RiemannData &rd, const double p_star)
    {
      const double radicand_inverse =
          0.5 * rd.rho * ((P.gamma + 1.) * p_star + (P.gamma - 1.) * rd.p);
      const double denominator = (p_star + (P.gamma - 1.) * P.gamma_plus_one_inverse * rd.p);
      const double true_value =
          (denominator - 0.5 * (p_star - rd.p)) / (denominator * sqrt(radicand_inverse));
      const double exponent = (-1. - P.gamma) * 0.5 * P.gamma_inverse;
      const double factor =
          (P.gamma - 1.) * 0.5 * P.gamma_inverse * dev_pow(p_star / rd.p, exponent) / rd.p;
      const double false_value = factor * 2. * rd.a * P.gamma_minus_one_inverse;
      return p_star >= rd.p ? true_value : false_value;
    }

    /* :164-205 */
    static RYUJIN_DEV double lambda1_minus(const EulerParams &P, const RiemannData &rd,
                                           const double p_star)
    {
      const double factor = (P.gamma + 1.0) * 0.5 * P.gamma_inverse;
      const double inv_p = 1.0 / rd.p;
      const double tmp = positive_part((p_star - rd.p) * inv_p);
      return rd.u - rd.a * sqrt(1.0 + factor * tmp);
    }
    static RYUJIN_DEV double lambda3_plus(const EulerParams &P, const RiemannData &rd,
                                          const double p_star)
    {
      const double factor = (P.gamma + 1.0) * 0.5 * P.gamma_inverse;
      const double inv_p = 1.0 / rd.p;
      const double tmp = positive_part((p_star - rd.p) * inv_p);
      return rd.u + rd.a * sqrt(1.0 + factor * tmp);
    }

    /* :217-238 */
    static RYUJIN_DEV void compute_gap(const EulerParams &P, const RiemannData &rd_i,
                                       const RiemannData &rd_j, const double p_1, const double p_2,
                                       double &gap, double &lambda_max)
    {
      const double nu_11 = lambda1_minus(P, rd_i, p_2 /*SIC!*/);
      const double nu_12 = lambda1_minus(P, rd_i, p_1 /*SIC!*/);
      const double nu_31 = lambda3_plus(P, rd_j, p_1);
      const double nu_32 = lambda3_plus(P, rd_j, p_2);
      lambda_max = fmax(positive_part(nu_32), negative_part(nu_11));
      gap = fmax(fabs(nu_32 - nu_31), fabs(nu_12 - nu_11));
    }

    /* :406-582 */
    static RYUJIN_DEV double riemann_compute(const EulerParams &P, const RiemannData &rd_i,
                                             const RiemannData &rd_j)
    {
      const double p_max = fmax(rd_i.p, rd_j.p);

      /* p_star_two_rarefaction :274-319 */
      double rarefaction;
      {
        const double inv_p_j = 1. / rd_j.p;
        const double factor = (P.gamma - 1.) * 0.5;
        const double numerator = positive_part(rd_i.a + rd_j.a - factor * (rd_j.u - rd_i.u));
        const double denominator =
            rd_i.a * dev_pow(rd_i.p * inv_p_j, -factor * P.gamma_inverse) + rd_j.a;
        const double exponent = 2.0 * P.gamma * P.gamma_minus_one_inverse;
        rarefaction = rd_j.p * dev_pow(numerator / denominator, exponent);
      }

      /* p_star_failsafe :330-374 */
      double failsafe;
      {
        double radicand_i = 2. * p_max;
        radicand_i /= rd_i.rho * ((P.gamma + 1.) * p_max + (P.gamma - 1.) * rd_i.p);
        const double x_i = sqrt(radicand_i);
        double radicand_j = 2. * p_max;
        radicand_j /= rd_j.rho * ((P.gamma + 1.) * p_max + (P.gamma - 1.) * rd_j.p);
        const double x_j = sqrt(radicand_j);
        const double a = x_i + x_j;
        const double b = rd_j.u - rd_i.u;
        const double c = -rd_i.p * x_i - rd_j.p * x_j;
        const double base = (-b + sqrt(b * b - 4. * a * c)) / (2. * a);
        failsafe = base * base;
      }
      const double p_star_tilde = fmin(rarefaction, failsafe);

      /* phi_of_p_max :122-149 */
      double phi_p_max;
      {
        const double radicand_inverse_i =
            0.5 * rd_i.rho * ((P.gamma + 1.) * p_max + (P.gamma - 1.) * rd_i.p);
        const double value_i = (p_max - rd_i.p) / sqrt(radicand_inverse_i);
        const double radicand_inverse_j =
            0.5 * rd_j.rho * ((P.gamma + 1.) * p_max + (P.gamma - 1.) * rd_j.p);
        const double value_j = (p_max - rd_j.p) / sqrt(radicand_inverse_j);
        phi_p_max = value_i + value_j + rd_j.u - rd_i.u;
      }

      double p_2 = phi_p_max < 0. ? p_star_tilde : fmin(p_max, p_star_tilde);

      if (P.riemann_newton_max_iterations == 0) {
        /* compute_lambda :252-263 */
        const double nu_11 = lambda1_minus(P, rd_i, p_2);
        const double nu_32 = lambda3_plus(P, rd_j, p_2);
        return fmax(positive_part(nu_32), negative_part(nu_11));
      }

      const double p_min = fmin(rd_i.p, rd_j.p);
      double p_1 = phi_p_max < 0. ? p_max : p_min;
      p_1 = p_1 <= p_2 ? p_1 : p_2;

      double gap, lambda_max;
      compute_gap(P, rd_i, rd_j, p_1, p_2, gap, lambda_max);

      for (int it = 0; it < P.riemann_newton_max_iterations; ++it) {
        if (fmax(0., gap - P.riemann_newton_tolerance) == 0.)
          break;
        const double phi_p_1 = rs_f(P, rd_i, p_1) + rs_f(P, rd_j, p_1) + rd_j.u - rd_i.u;
        const double phi_p_2 = rs_f(P, rd_i, p_2) + rs_f(P, rd_j, p_2) + rd_j.u - rd_i.u;
        const double dphi_p_1 = rs_df(P, rd_i, p_1) + rs_df(P, rd_j, p_1);
        const double dphi_p_2 = rs_df(P, rd_i, p_2) + rs_df(P, rd_j, p_2);
        quadratic_newton_step(p_1, p_2, phi_p_1, phi_p_2, dphi_p_1, dphi_p_2, 1.0);
        compute_gap(P, rd_i, rd_j, p_1, p_2, gap, lambda_max);
      }
      return lambda_max;
    }

    /* d_ij = |c| * lambda_max(U_i, U_j, c/|c|): hyperbolic_module.template.h:402-406 */
    static RYUJIN_DEV double dij_from_states(const EulerParams &P, const double (&U_i)[K],
                                             const double (&U_j)[K], const double (&c)[DIM])
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
      const RiemannData rd_i = riemann_data_from_state(P, U_i, n);
      const RiemannData rd_j = riemann_data_from_state(P, U_j, n);
      return norm * riemann_compute(P, rd_i, rd_j);
    }

    /* ------------------------------------------------------------------ Riemann solver, node records
     *
     * The Riemann sweep is bound by the latency of its FP64 division / sqrt / pow chains, not by bandwidth
     * (DESIGN.md section 3). Everything riemann_data_from_state (:377-403) derives from ONE state except the
     * normal velocity does not depend on the direction n_ij: |m|^2 = (m.n)^2 + |m - (m.n) n|^2 for |n| = 1,
     * hence p = (gamma - 1)(E - |m|^2 / (2 rho)) and a = sqrt(gamma p / rho). A row recomputes these for
     * itself and every neighbour, ~4 (2-D) to ~13 (3-D) times per node. They are therefore computed ONCE per
     * node next to the precomputed values (k_precompute_records) together with p^((gamma-1)/(2 gamma)), which turns
     *   (p_i / p_j)^(-(gamma-1)/(2 gamma))  into  pw_j / pw_i        (p_star_two_rarefaction, :296-313),
     * and the second pow of that formula has the exponent 2 gamma / (gamma - 1) = 7 for gamma = 7/5.
     * phi(p_max) (:134-148) shares its square roots with p_star_failsafe (:348-368):
     *   1 / sqrt(rho/2 ((gamma+1) p_max + (gamma-1) p_Z)) = x_Z / sqrt(p_max).
     * Per (i,j) pair this leaves 5 divisions and 7 square roots (before: 2 pow, 11 divisions, 10 square
     * roots). The results differ from the reference's operation order by a few ulp (1e-15 relative): inside
     * the 1e-12 contract on d_ij, checked against the oracle on 2 x 200 k random state pairs
     * (tests/test_gpu_device_functions.py) and in every sweep comparison. With Newton iterations of the
     * Riemann solver switched on (non-default) the reference path (riemann_compute) is used on the same data.
     *
     * record = (rho, p, a, pw, a / pw, 1 / p, v[DIM]) padded to an even number of doubles in 1-D and 2-D (64 B).
     * In 3-D the record is (rho, p, a, pw, a / pw, 1 / p, m[DIM], E, eta, 1 / rho), 96 B: the Riemann data of the
     * node AND what the indicator needs of it (the state, the Harten entropy, 1 / rho), so that step 2 gathers ONE
     * record per neighbour instead of the state (48 B), the precomputed values (16 B) and a Riemann record (80 B)
     * from three arrays; the velocity is formed as m (1 / rho), the product the short record stores. Measured on
     * the C4 share (profiles/r04g_*): step 2 1.434 -> 1.354 ms at 3 waves per SIMD. In 2-D the same layout trades
     * 112 B of gathers for 96 and loses (0.240 -> 0.259 ms on C2: the sweep is not bound by its gather bytes and
     * the longer record costs registers, 128 -> 166), so the short record stays there. */
    static constexpr bool kRecordHoldsState = DIM == 3;
    static constexpr int kRecRho = 0, kRecP = 1, kRecA = 2, kRecPw = 3, kRecApw = 4, kRecPinv = 5, kRecM = 6,
                         kRecE = 6 + DIM, kRecEta = 7 + DIM, kRecRinv = 8 + DIM;
    static constexpr int RS = ((kRecordHoldsState ? 9 : 6) + DIM + 1) / 2 * 2;

    static RYUJIN_DEV double record_velocity(const double (&rec)[RS], const int d)
    {
      if constexpr (kRecordHoldsState)
        return rec[kRecM + d] * rec[kRecRinv];
      else
        return rec[kRecM + d];
    }

    static RYUJIN_DEV void state_of_record(const double (&rec)[RS], double (&U)[K])
    {
      U[0] = rec[kRecRho];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        U[1 + d] = rec[kRecM + d];
      U[1 + DIM] = rec[kRecE];
    }

    /* flux() of the record's state: 1 / rho and p are the values flux() computes (:1164-1181) */
    static RYUJIN_DEV void flux_of_record(const double (&rec)[RS], double (&f)[K][DIM])
    {
      const double rho_inverse = rec[kRecRinv];
      const double p = rec[kRecP];
      const double E = rec[kRecE];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        f[0][d] = rec[kRecM + d];
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const double s = rec[kRecM + i] * rho_inverse;
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          f[1 + i][d] = rec[kRecM + d] * s;
        f[1 + i][i] += p;
      }
      const double s = rho_inverse * (E + p);
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        f[DIM + 1][d] = rec[kRecM + d] * s;
    }

    /* the record of a node with state U and Harten entropy eta */
    static RYUJIN_DEV void node_record(const EulerParams &P, const double (&U)[K], const double2 prec,
                                       double (&rec)[RS])
    {
      const double rho = U[0];
      const double rho_inverse = 1. / rho;
      const double p = (P.gamma - 1.) * internal_energy(U);
      const double a = sqrt(P.gamma * p * rho_inverse);
      const double pw = dev_pow(p, (P.gamma - 1.) * 0.5 * P.gamma_inverse);
      rec[kRecRho] = rho;
      rec[kRecP] = p;
      rec[kRecA] = a;
      rec[kRecPw] = pw;
      rec[kRecApw] = a / pw;
      rec[kRecPinv] = 1. / p;
      if constexpr (kRecordHoldsState) {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          rec[kRecM + d] = U[1 + d];
        rec[kRecE] = U[1 + DIM];
        rec[kRecEta] = prec.y;
        rec[kRecRinv] = rho_inverse;
#pragma unroll
        for (int d = 9 + DIM; d < RS; ++d)
          rec[d] = 0.;
      } else {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          rec[kRecM + d] = U[1 + d] * rho_inverse;
#pragma unroll
        for (int d = 6 + DIM; d < RS; ++d)
          rec[d] = 0.;
      }
    }

    static RYUJIN_DEV void riemann_record(const EulerParams &P, const double (&U)[K], double (&rec)[RS])
    {
      node_record(P, U, precompute(P, U), rec);
    }

    /* (tests: the record of a node given by its Riemann data -- density, pressure, speed of sound, velocity) */
    static RYUJIN_DEV void riemann_record_from_primitive(const EulerParams &P, const double rho, const double p,
                                                         const double a, const double (&v)[DIM],
                                                         double (&rec)[RS])
    {
      const double pw = dev_pow(p, (P.gamma - 1.) * 0.5 * P.gamma_inverse);
#pragma unroll
      for (int d = 0; d < RS; ++d)
        rec[d] = 0.;
      rec[kRecRho] = rho;
      rec[kRecP] = p;
      rec[kRecA] = a;
      rec[kRecPw] = pw;
      rec[kRecApw] = a / pw;
      rec[kRecPinv] = 1. / p;
      if constexpr (kRecordHoldsState)
        rec[kRecRinv] = 1.; /* (m := v, 1 / rho := 1: the velocity the Riemann solver forms is v itself) */
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        rec[kRecM + d] = v[d];
    }

    /* GENERAL = false: Newton iterations off and integral rarefaction exponent (the defaults with
     * gamma = 7/5), known on the host -- the kernel then contains neither pow nor the Newton loop */
    template <bool GENERAL = true>
    static RYUJIN_DEV double dij_from_records(const EulerParams &P, const double (&ri)[RS],
                                              const double (&rj)[RS], const double (&c)[DIM])
    {
      double norm2 = c[0] * c[0];
      double vc_i = record_velocity(ri, 0) * c[0], vc_j = record_velocity(rj, 0) * c[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        norm2 += c[d] * c[d];
        vc_i += record_velocity(ri, d) * c[d];
        vc_j += record_velocity(rj, d) * c[d];
      }
      const double norm = sqrt(norm2);
      const double inverse_norm = 1. / norm;
      const double u_i = vc_i * inverse_norm, u_j = vc_j * inverse_norm;

      if constexpr (GENERAL) {
        if (P.riemann_newton_max_iterations != 0) {
          const RiemannData rd_i{ri[0], u_i, ri[1], ri[2]}, rd_j{rj[0], u_j, rj[1], rj[2]};
          return norm * riemann_compute(P, rd_i, rd_j);
        }
      }

      const double rho_i = ri[0], p_i = ri[1], a_i = ri[2];
      const double rho_j = rj[0], p_j = rj[1], a_j = rj[2];
      const double p_max = fmax(p_i, p_j);
      const double du = u_j - u_i;

      /* p_star_two_rarefaction :274-319 */
      double rarefaction;
      {
        const double factor = (P.gamma - 1.) * 0.5;
        const double numerator = positive_part(a_i + a_j - factor * du);
        const double denominator = ri[4] * rj[3] + a_j; /* a_i (p_i/p_j)^(-(gamma-1)/(2 gamma)) + a_j */
        const double x = numerator / denominator;
        double power;
        if (!GENERAL || P.rarefaction_power > 0) {
          power = 1.;
          double b = x;
          for (int n = P.rarefaction_power; n; n >>= 1) { /* wave-uniform: at most 4 squarings */
            if (n & 1)
              power *= b;
            b *= b;
          }
        } else {
          power = dev_pow(x, 2.0 * P.gamma * P.gamma_minus_one_inverse);
        }
        rarefaction = p_j * power;
      }

      /* p_star_failsafe :330-374 and phi_of_p_max :122-149 on shared square roots */
      const double gp1 = P.gamma + 1., gm1 = P.gamma - 1.;
      const double x_i = sqrt(2. * p_max / (rho_i * (gp1 * p_max + gm1 * p_i)));
      const double x_j = sqrt(2. * p_max / (rho_j * (gp1 * p_max + gm1 * p_j)));
      double failsafe;
      {
        const double a = x_i + x_j;
        const double b = du;
        const double cc = -p_i * x_i - p_j * x_j;
        const double base = (-b + sqrt(b * b - 4. * a * cc)) / (2. * a);
        failsafe = base * base;
      }
      const double p_star_tilde = fmin(rarefaction, failsafe);
      const double phi_p_max = ((p_max - p_i) * x_i + (p_max - p_j) * x_j) * (1. / sqrt(p_max)) + du;
      const double p_2 = phi_p_max < 0. ? p_star_tilde : fmin(p_max, p_star_tilde);

      /* compute_lambda :252-263 with lambda1_minus / lambda3_plus :164-205 */
      const double f = (P.gamma + 1.0) * 0.5 * P.gamma_inverse;
      const double nu_11 = u_i - a_i * sqrt(1.0 + f * positive_part((p_2 - p_i) * ri[5]));
      const double nu_32 = u_j + a_j * sqrt(1.0 + f * positive_part((p_2 - p_j) * rj[5]));
      return norm * fmax(positive_part(nu_32), negative_part(nu_11));
    }

    /* ------------------------------------------------------------------ Limiter::limit */

    /* The common case of limit() (limiter.template.h:40-108 + the first psi_r test :173-216):
     * density clip, then "psi_r > 0 => return t_r". Only 1 in 25-50 (i,j) pairs needs the Newton
     * iteration (reference comment :197-201); with 64 lanes per wave nearly every wave would drag
     * the masked-off lanes through it. The kernels therefore run this wave-uniform fast part over
     * all columns first and finish the few undecided (row, col) pairs with limit() afterwards.
     * Returns the limiter value if decided, otherwise sets undecided = true. */
    /* density clip (limiter.template.h:40-108) and psi_r of the first Newton iteration (:173-182) */
    static RYUJIN_DEV double first_psi_r(const EulerParams &P, const double (&bnd)[NB], const double (&U)[K],
                                         const double (&Pij)[K], bool &success, double &t_r)
    {
      const double rho_min = bnd[0], rho_max = bnd[1], s_min = bnd[2];
      constexpr double t_min = 0., t_max = 1.;
      constexpr double eps = DBL_EPSILON;
      success = true;
      const double relax_small = 1. + P.vacuum_small * eps;
      const double relax = 1. + P.vacuum_large * eps;
      t_r = t_max;
      {
        const double rho_U = U[0];
        const double rho_P = Pij[0];
        const double test_min = filter_vacuum_density(P, fmax(0., rho_U - relax * rho_max));
        const double test_max = filter_vacuum_density(P, fmax(0., rho_min - relax * rho_U));
        if (!(test_min == 0. && test_max == 0.))
          success = false;
        /* the clip needs a division; with t_r = t_max = 1 neither comparison fires for most pairs and t_r stays
         * t_max exactly (the clamps below are the identity on it): the division is only issued where a lane needs
         * it (RYUJIN_LIMIT_GUARD_CLIP 0: always, as rounds 1 - 3) */
        if (!RYUJIN_LIMIT_GUARD_CLIP || rho_max < rho_U + t_r * rho_P || rho_U + t_r * rho_P < rho_min) {
          const double denominator = 1. / (fabs(rho_P) + eps * rho_max);
          t_r = rho_max < rho_U + t_r * rho_P ? (rho_max - rho_U) * denominator : t_r;
          t_r = rho_U + t_r * rho_P < rho_min ? (rho_U - rho_min) * denominator : t_r;
          t_r = fmin(t_r, t_max);
          t_r = fmax(t_r, t_min);
        }
      }
      double U_r[K];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_r[q] = U[q] + t_r * Pij[q];
      const double rho_r = U_r[0];
      const double rho_r_gamma = dev_pow(rho_r, P.gamma);
      const double rho_e_r = internal_energy(U_r);
      return relax_small * rho_r * rho_e_r - s_min * rho_r * rho_r_gamma;
    }

    static RYUJIN_DEV double limit_fast(const EulerParams &P, const double (&bnd)[NB],
                                        const double (&U)[K], const double (&Pij)[K], bool &success,
                                        bool &undecided)
    {
      constexpr double t_min = 0.;
      undecided = false;
      double t_r;
      const double psi_r = first_psi_r(P, bnd, U, Pij, success, t_r);
      if (P.lim_newton_max_iterations <= 0)
        return t_min;
      if (psi_r > 0.)
        return t_r; /* t_l = t_r, break */
      if (t_r == t_min)
        return t_min; /* t_l == t_r, break */
      undecided = true;
      return t_min;
    }

    /* limiter.template.h:15-327, production control flow (no EXPENSIVE_BOUNDS_CHECK).
     * Per-thread early exits: a converged lane is a fixed point of quadratic_newton_step,
     * see SURVEY.md Appendix E-3. */
    static RYUJIN_DEV double limit(const EulerParams &P, const double (&bnd)[NB],
                                   const double (&U)[K], const double (&Pij)[K], bool &success)
    {
      const double rho_min = bnd[0], rho_max = bnd[1], s_min = bnd[2];
      constexpr double t_min = 0., t_max = 1.;
      success = true;
      double t_r = t_max;

      constexpr double eps = DBL_EPSILON;
      const double relax_small = 1. + P.vacuum_small * eps;
      const double relax = 1. + P.vacuum_large * eps;

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
      }

      double t_l = t_min;
      const double gamma = P.gamma;
      const double gp1 = gamma + 1.;

      for (int n = 0; n < P.lim_newton_max_iterations; ++n) {
        double U_r[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_r[q] = U[q] + t_r * Pij[q];
        const double rho_r = U_r[0];
        const double rho_r_gamma = dev_pow(rho_r, gamma);
        const double rho_e_r = internal_energy(U_r);
        const double psi_r = relax_small * rho_r * rho_e_r - s_min * rho_r * rho_r_gamma;

        t_l = psi_r > 0. ? t_r : t_l;
        if (t_l == t_r)
          break;

        double U_l[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_l[q] = U[q] + t_l * Pij[q];
        const double rho_l = U_l[0];
        const double rho_l_gamma = dev_pow(rho_l, gamma);
        const double rho_e_l = internal_energy(U_l);
        const double psi_l = relax_small * rho_l * rho_e_l - s_min * rho_l * rho_l_gamma;

        const double lower_bound = (1. - relax) * s_min * rho_l * rho_l_gamma;
        if (n == 0 && !(fmin(0., psi_l - lower_bound) == 0.))
          success = false;

        if (fmax(0., t_r - t_l - P.lim_newton_tolerance) == 0.)
          break;

        /* internal_energy_derivative(U) * P: hyperbolic_system.h:797-819 */
        const double drho = Pij[0];
        double drho_e_l, drho_e_r;
        {
          const double rho_inverse = 1. / U_l[0];
          double u[DIM];
          double u2;
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            u[d] = U_l[1 + d] * rho_inverse;
          u2 = u[0] * u[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            u2 += u[d] * u[d];
          double s = (0.5 * u2) * Pij[0];
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            s += (-u[d]) * Pij[1 + d];
          s += 1. * Pij[1 + DIM];
          drho_e_l = s;
        }
        {
          const double rho_inverse = 1. / U_r[0];
          double u[DIM];
          double u2;
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            u[d] = U_r[1 + d] * rho_inverse;
          u2 = u[0] * u[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            u2 += u[d] * u[d];
          double s = (0.5 * u2) * Pij[0];
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            s += (-u[d]) * Pij[1 + d];
          s += 1. * Pij[1 + DIM];
          drho_e_r = s;
        }
        const double dpsi_l = rho_l * drho_e_l + (rho_e_l - gp1 * s_min * rho_l_gamma) * drho;
        const double dpsi_r = rho_r * drho_e_r + (rho_e_r - gp1 * s_min * rho_r_gamma) * drho;

        quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
      }
      return t_l;
    }

    /* View::is_admissible (hyperbolic_system.h:955-979): rho > 0, e > 0, s > 0 */
    static RYUJIN_DEV bool is_admissible(const EulerParams &P, const double (&U)[K])
    {
      return U[0] > 0. && internal_energy(U) > 0. && specific_entropy(P, U) > 0.;
    }

    /* limiter.template.h:15-327 in the EXPENSIVE_BOUNDS_CHECK control flow (the reference's checked builds, and
     * the build that wrote tests/euler/limiter.output): the high-order density check behind the clip (:110-134), no
     * "psi_r > 0" shortcut in front of psi_l (:183-217 against :244-252), the final check of the limited state
     * (:291-322). Same t_l as limit(); `success` has more ways to be false. Debug kernels only
     * (ryujin_hip_params::debug_expensive_bounds_check, RYUJIN_DEBUG_EULER_LIMIT_CHECKED_1D). */
    static RYUJIN_DEV double limit_checked(const EulerParams &P, const double (&bnd)[NB], const double (&U)[K],
                                           const double (&Pij)[K], bool &success)
    {
      const double rho_min = bnd[0], rho_max = bnd[1], s_min = bnd[2];
      constexpr double t_min = 0., t_max = 1.;
      success = true;
      double t_r = t_max;
      constexpr double eps = DBL_EPSILON;
      const double relax_small = 1. + P.vacuum_small * eps;
      const double relax = 1. + P.vacuum_large * eps;
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
      const double gamma = P.gamma;
      const double gp1 = gamma + 1.;
      for (int n = 0; n < P.lim_newton_max_iterations; ++n) {
        double U_r[K], U_l[K];
#pragma unroll
        for (int q = 0; q < K; ++q) {
          U_r[q] = U[q] + t_r * Pij[q];
          U_l[q] = U[q] + t_l * Pij[q];
        }
        const double rho_r = U_r[0];
        const double rho_r_gamma = dev_pow(rho_r, gamma);
        const double rho_e_r = internal_energy(U_r);
        const double psi_r = relax_small * rho_r * rho_e_r - s_min * rho_r * rho_r_gamma;
        const double rho_l = U_l[0];
        const double rho_l_gamma = dev_pow(rho_l, gamma);
        const double rho_e_l = internal_energy(U_l);
        const double psi_l = relax_small * rho_l * rho_e_l - s_min * rho_l * rho_l_gamma;
        const double lower_bound = (1. - relax) * s_min * rho_l * rho_l_gamma;
        if (n == 0 && !(fmin(0., psi_l - lower_bound) == 0.))
          success = false;
        t_l = psi_r > 0. ? t_r : t_l;
        if (fmax(0., t_r - t_l - P.lim_newton_tolerance) == 0.)
          break;
        const double drho = Pij[0];
        double drho_e[2];
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          const double(&V)[K] = side == 0 ? U_l : U_r;
          const double rho_inverse = 1. / V[0];
          double u[DIM];
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            u[d] = V[1 + d] * rho_inverse;
          double u2 = u[0] * u[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            u2 += u[d] * u[d];
          double sum = (0.5 * u2) * Pij[0];
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            sum += (-u[d]) * Pij[1 + d];
          sum += 1. * Pij[1 + DIM];
          drho_e[side] = sum;
        }
        const double dpsi_l = rho_l * drho_e[0] + (rho_e_l - gp1 * s_min * rho_l_gamma) * drho;
        const double dpsi_r = rho_r * drho_e[1] + (rho_e_r - gp1 * s_min * rho_r_gamma) * drho;
        quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
      }
      {
        double U_new[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_new[q] = U[q] + t_l * Pij[q];
        const double rho_new = U_new[0];
        const double rho_new_gamma = dev_pow(rho_new, gamma);
        const double rho_e_new = internal_energy(U_new);
        const double psi_new = relax_small * rho_new * rho_e_new - s_min * rho_new * rho_new_gamma;
        const double lower_bound = (1. - relax) * s_min * rho_new * rho_new_gamma;
        const bool e_valid = fmin(0., rho_e_new) == 0.;
        const bool psi_valid = fmin(0., psi_new - lower_bound) == 0.;
        if (!e_valid || !psi_valid)
          success = false;
      }
      return t_l;
    }

    /* ------------------------------------------------------------------ boundary conditions */

    /* hyperbolic_system.h:1040-1093 */
    template <int component>
    static RYUJIN_DEV void prescribe_riemann_characteristic(const EulerParams &P, const double (&U)[K],
                                                            const double (&U_bar)[K],
                                                            const double (&normal)[DIM],
                                                            double (&U_new)[K])
    {
      const double gamma = P.gamma;
      const double rho = U[0];
      const double a = sqrt(gamma * ((gamma - 1.) * internal_energy(U)) * (1. / rho));
      double mn = U[1] * normal[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        mn += U[1 + d] * normal[d];
      const double vn = mn / rho;

      const double rho_bar = U_bar[0];
      const double a_bar = sqrt(gamma * ((gamma - 1.) * internal_energy(U_bar)) * (1. / rho_bar));
      double mn_bar = U_bar[1] * normal[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        mn_bar += U_bar[1 + d] * normal[d];
      const double vn_bar = mn_bar / rho_bar;

      const double R_1 =
          component == 1 ? vn_bar - 2. * a_bar / (gamma - 1.) : vn - 2. * a / (gamma - 1.);
      const double R_2 =
          component == 2 ? vn_bar + 2. * a_bar / (gamma - 1.) : vn + 2. * a / (gamma - 1.);

      const double p = (gamma - 1.) * internal_energy(U);
      const double s = p / dev_pow(rho, gamma);

      double vperp[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        vperp[d] = U[1 + d] / rho - vn * normal[d];

      const double vn_new = 0.5 * (R_1 + R_2);
      const double tmp = ((gamma - 1.) / 4.) * (R_2 - R_1);
      double rho_new = 1. / (gamma * s) * (tmp * tmp);
      rho_new = dev_pow(rho_new, 1. / (gamma - 1.));
      const double p_new = s * dev_pow(rho_new, gamma);

      U_new[0] = rho_new;
      double vperp2 = vperp[0] * vperp[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        vperp2 += vperp[d] * vperp[d];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        U_new[1 + d] = rho_new * (vn_new * normal[d] + vperp[d]);
      U_new[1 + DIM] = p_new / (gamma - 1.) + 0.5 * rho_new * (vn_new * vn_new + vperp2);
    }

    /* hyperbolic_system.h:1099-1159 */
    static RYUJIN_DEV void apply_boundary_conditions(const EulerParams &P, const int id,
                                                     const double (&U)[K], const double (&normal)[DIM],
                                                     const double (&U_D)[K], double (&result)[K])
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
      } else if (id == RYUJIN_BC_DYNAMIC) {
        const double rho = U[0];
        const double a = sqrt(P.gamma * ((P.gamma - 1.) * internal_energy(U)) * (1. / rho));
        double mn = U[1] * normal[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          mn += U[1 + d] * normal[d];
        const double vn = mn / rho;
        if (vn < -a) {
#pragma unroll
          for (int q = 0; q < K; ++q)
            result[q] = U_D[q];
        }
        if (vn >= -a && vn <= 0.)
          prescribe_riemann_characteristic<2>(P, U_D, U, normal, result);
        if (vn > 0. && vn <= a)
          prescribe_riemann_characteristic<1>(P, U, U_D, normal, result);
      }
    }
  };
} // namespace ryujin_hip

// Device-side shallow-water "Description" (BASELINE.json configs[4]: second equation module through
// the same sweeps). Restates, operation order preserved:
//   HyperbolicSystemView  source/shallow_water/hyperbolic_system.h:676-1248
//   RiemannSolver         source/shallow_water/riemann_solver.template.h:25-251
//   Indicator             source/shallow_water/indicator.h:155-222
//   Limiter               source/shallow_water/limiter.h:247-363, limiter.template.h:16-449
// State U = (h, q), k = dim + 1; precomputed = (eta_m, h_sharp^{4/3}); initial_precomputed = Z.

#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

#include "euler_device.hpp"

namespace ryujin_hip
{
  struct ShallowWaterParams {
    double gravity, manning, reference_water_depth;
    double dry_state_relaxation_factor, dry_small, dry_large;
    double evc_factor;
    double lim_newton_tolerance, lim_relaxation_factor;
    int limit_on_kinetic_energy, limit_on_square_velocity;
  };

  template <int DIM>
  struct ShallowWater {
    static constexpr int DIMENSION = DIM;
    static constexpr int K = DIM + 1;
    static constexpr int NB = 5;
    using Params = ShallowWaterParams;

    /* hyperbolic_system.h:729-742 */
    static RYUJIN_DEV double inverse_water_depth_mollified(const Params &P, const double (&U)[K])
    {
      const double h_cutoff_mollified = P.reference_water_depth * P.dry_large * DBL_EPSILON;
      const double h = U[0];
      const double h_pos = positive_part(h);
      const double h_max = fmax(h, h_cutoff_mollified);
      const double denom = h * h + h_max * h_max;
      return 2. * h_pos / denom;
    }

    /* :747-768 */
    static RYUJIN_DEV double water_depth_sharp(const Params &P, const double (&U)[K])
    {
      const double h_cutoff_small = P.reference_water_depth * P.dry_small * DBL_EPSILON;
      return fmax(U[0], h_cutoff_small);
    }
    static RYUJIN_DEV double inverse_water_depth_sharp(const Params &P, const double (&U)[K])
    {
      return 1. / water_depth_sharp(P, U);
    }

    /* :773-784 */
    static RYUJIN_DEV double filter_dry_water_depth(const Params &P, const double h)
    {
      const double h_cutoff_large = P.reference_water_depth * P.dry_large * DBL_EPSILON;
      return fabs(h) < h_cutoff_large ? 0. : h;
    }

    /* :801-810 */
    static RYUJIN_DEV double kinetic_energy(const Params &P, const double (&U)[K])
    {
      const double h = U[0];
      const double ih = inverse_water_depth_sharp(P, U);
      double v2;
      {
        const double v = U[1] * ih;
        v2 = v * v;
      }
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        const double v = U[1 + d] * ih;
        v2 += v * v;
      }
      return 0.5 * h * v2;
    }

    /* :815-822 */
    static RYUJIN_DEV double pressure(const Params &P, const double (&U)[K])
    {
      const double h_sqd = U[0] * U[0];
      return 0.5 * P.gravity * h_sqd;
    }

    /* :849-877 */
    static RYUJIN_DEV void mathematical_entropy_derivative(const Params &P, const double (&U)[K],
                                                           double (&result)[K])
    {
      const double h = U[0];
      const double ih = inverse_water_depth_sharp(P, U);
      double v2;
      {
        const double v = U[1] * ih;
        v2 = v * v;
        result[1] = v;
      }
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        const double v = U[1 + d] * ih;
        v2 += v * v;
        result[1 + d] = v;
      }
      result[0] = P.gravity * h - 0.5 * v2;
    }

    /* g: :1042-1055, f = g + p I: :1022-1037 */
    static RYUJIN_DEV void g(const Params &P, const double (&U)[K], double (&r)[K][DIM])
    {
      const double h_inverse = inverse_water_depth_sharp(P, U);
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        r[0][d] = (U[1 + d] * h_inverse) * U[0];
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          r[1 + i][d] = (U[1 + d] * h_inverse) * U[1 + i];
    }
    static RYUJIN_DEV void f(const Params &P, const double (&U)[K], double (&r)[K][DIM])
    {
      g(P, U, r);
      const double p = pressure(P, U);
#pragma unroll
      for (int i = 0; i < DIM; ++i)
        r[1 + i][i] += p;
    }

    /* :1060-1070 */
    static RYUJIN_DEV void star_state(const Params &P, const double (&U)[K], const double Z_left,
                                      const double Z_right, double (&r)[K])
    {
      const double Z_max = fmax(Z_left, Z_right);
      const double H_star = fmax(0., U[0] + Z_left - Z_max);
      const double ihm = inverse_water_depth_mollified(P, U);
#pragma unroll
      for (int q = 0; q < K; ++q)
        r[q] = U[q] * H_star * ihm;
    }

    static RYUJIN_DEV void contract(const double (&fl)[K][DIM], const double (&c)[DIM],
                                    double (&out)[K])
    {
#pragma unroll
      for (int q = 0; q < K; ++q) {
        double s = fl[q][0] * c[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          s += fl[q][d] * c[d];
        out[q] = s;
      }
    }

    /* :1117-1144 */
    static RYUJIN_DEV void flux_divergence(const Params &P, const double (&U_i)[K],
                                           const double (&U_star_ij)[K],
                                           const double (&U_star_ji)[K], const double (&c)[DIM],
                                           double (&out)[K])
    {
      const double H_i = U_i[0];
      const double H_star_ij = U_star_ij[0];
      const double H_star_ji = U_star_ji[0];
      double g_i[K][DIM], g_j[K][DIM], result[K][DIM];
      g(P, U_star_ij, g_i);
      g(P, U_star_ji, g_j);
#pragma unroll
      for (int q = 0; q < K; ++q)
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          result[q][d] = -(g_i[q][d] + g_j[q][d]);
      const double factor =
          (0.5 * (H_star_ji * H_star_ji - H_star_ij * H_star_ij) + H_i * H_i) * P.gravity;
#pragma unroll
      for (int i = 0; i < DIM; ++i)
        result[1 + i][i] -= factor;
      contract(result, c, out);
    }

    /* :1149-1171 */
    static RYUJIN_DEV void high_order_flux_divergence(const Params &P, const double (&U_i)[K],
                                                      const double Z_i, const double (&U_j)[K],
                                                      const double Z_j, const double (&c)[DIM],
                                                      double (&out)[K])
    {
      double g_i[K][DIM], g_j[K][DIM], result[K][DIM];
      g(P, U_i, g_i);
      g(P, U_j, g_j);
#pragma unroll
      for (int q = 0; q < K; ++q)
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          result[q][d] = -(g_i[q][d] + g_j[q][d]);
      const double factor = P.gravity * U_i[0] * (U_j[0] + Z_j - Z_i);
#pragma unroll
      for (int i = 0; i < DIM; ++i)
        result[1 + i][i] -= factor;
      contract(result, c, out);
    }

    /* :1196-1218 */
    static RYUJIN_DEV void manning_friction(const Params &P, const double (&U)[K],
                                            const double h_star, const double tau, double (&r)[K])
    {
      /* n = 0 (the reference's default and BASELINE configs[4]): factor = 0 and the source is -0 * x = (-)0 for
       * every finite x -- return the zeros without the division and the square root (uniform branch on a kernel
       * argument; exact up to the sign of a zero that is only ever added to something) */
      if (P.manning == 0.) {
#pragma unroll
        for (int q = 0; q < K; ++q)
          r[q] = 0.;
        return;
      }
      const double h_inverse = inverse_water_depth_mollified(P, U);
      double v2;
      {
        const double v = U[1] * h_inverse;
        v2 = v * v;
      }
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        const double v = U[1 + d] * h_inverse;
        v2 += v * v;
      }
      const double v_norm = sqrt(v2);
      const double factor = 2. * P.gravity * P.manning * P.manning * v_norm;
      const double denominator = h_star + fmax(h_star, tau * factor);
      const double denominator_inverse = 1. / denominator;
      r[0] = 0.;
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        r[d + 1] = -factor * denominator_inverse * U[1 + d];
    }

    /* precomputed (eta_m, h_star): :676-716 */
    static RYUJIN_DEV double2 precompute(const Params &P, const double (&U)[K])
    {
      double2 out;
      out.x = pressure(P, U) + kinetic_energy(P, U);
      out.y = dev_pow(water_depth_sharp(P, U), 4. / 3.);
      return out;
    }

    /* `left` accumulates (eta_j + p_j) v_j . c_ij, not a difference: the row's own column counts (c_ii != 0 on the
     * boundary) */
    static constexpr bool kIndicatorDiagonalIsZero = false;
    /* precompute() and riemann_record() are functions of the row's state alone: the last sweep of a step can
     * leave them behind for the next prepare_state_vector() (FusedPrecompute) */
    static constexpr bool kFusablePrecompute = true;
    /* steps 6/7 may form a limited row's update as V_i - sum (1 - l_ij) lambda P_ij (kernels_limiter.hpp): another
     * rounding of the reference's sum. Not where l = 0 has to return the low-order update EXACTLY (a dry node) */
    static constexpr bool kLimitedUpdateFromV = false;

    /* ------------------------------------------------------------------ Indicator */
    struct Indicator {
      double eta_i, left;
      double d_eta_i[K], f_i[K][DIM], right[K];

      RYUJIN_DEV void reset(const Params &P, const double (&U_i)[K], const double2 prec_i)
      {
        eta_i = prec_i.x;
        mathematical_entropy_derivative(P, U_i, d_eta_i);
        f(P, U_i, f_i);
        left = 0.;
#pragma unroll
        for (int q = 0; q < K; ++q)
          right[q] = 0.;
      }
      RYUJIN_DEV void accumulate(const Params &P, const double (&U_j)[K], const double2 prec_j,
                                 const double (&c_ij)[DIM])
      {
        const double eta_j = prec_j.x;
        const double ih = inverse_water_depth_sharp(P, U_j);
        double f_j[K][DIM];
        f(P, U_j, f_j);
        const double pressure_j = pressure(P, U_j);
        double v_c = (U_j[1] * ih) * c_ij[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          v_c += (U_j[1 + d] * ih) * c_ij[d];
        left += (eta_j + pressure_j) * v_c;
#pragma unroll
        for (int q = 0; q < K; ++q) {
          double s = (f_j[q][0] - f_i[q][0]) * c_ij[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            s += (f_j[q][d] - f_i[q][d]) * c_ij[d];
          right[q] += s;
        }
      }
      RYUJIN_DEV double alpha(const Params &P, const double hd_i) const
      {
        double my_sum = 0.;
#pragma unroll
        for (int q = 0; q < K; ++q)
          my_sum += d_eta_i[q] * right[q];
        const double numerator = fabs(left - my_sum);
        const double denominator = fabs(left) + fabs(my_sum);
        const double regularization = 100. * DBL_MIN;
        const double quotient =
            fabs(numerator) / (denominator + fmax(hd_i * fabs(eta_i), regularization));
        return fmin(1., P.evc_factor * quotient);
      }
    };

    /* ------------------------------------------------------------------ Riemann solver */
    struct RiemannData {
      double h, u, a;
    };

    static RYUJIN_DEV double rs_f(const Params &P, const RiemannData &rd, const double h)
    {
      const double left_value = 2. * (sqrt(P.gravity * h) - rd.a);
      const double radicand = 0.5 * P.gravity * (h + rd.h) / (h * rd.h);
      const double right_value = (h - rd.h) * sqrt(radicand);
      return h <= rd.h ? left_value : right_value;
    }
    static RYUJIN_DEV double rs_phi(const Params &P, const RiemannData &rd_i, const RiemannData &rd_j,
                                    const double h)
    {
      return rs_f(P, rd_i, h) + rs_f(P, rd_j, h) + rd_j.u - rd_i.u;
    }

    /* riemann_solver.template.h:111-204 (the first mask result is overwritten, as written) */
    static RYUJIN_DEV double compute_h_star(const Params &P, const RiemannData &rd_i,
                                            const RiemannData &rd_j)
    {
      const double gravity_inverse = 1. / P.gravity;
      const double h_min = fmin(rd_i.h, rd_j.h);
      const double h_max = fmax(rd_i.h, rd_j.h);
      const double a_min = sqrt(P.gravity * h_min);
      const double a_max = sqrt(P.gravity * h_max);
      const double sqrt_two = sqrt(2.);
      const double x0 = 9. - 4. * sqrt_two;
      const double phi_value_max = rs_phi(P, rd_i, rd_j, x0 * h_max);

      double tmp = 1. + sqrt_two * (rd_i.u - rd_j.u) / (a_min + a_max);
      const double h_star_middle = sqrt(h_min * h_max) * tmp;

      const double left_radicand = 3. * h_min + 2. * sqrt_two * sqrt(h_min * h_max);
      const double right_radicand = sqrt_two * sqrt(gravity_inverse * h_min) * (rd_i.u - rd_j.u);
      tmp = sqrt(positive_part(left_radicand + right_radicand));
      tmp -= sqrt_two * sqrt(h_min);
      const double h_star_right = tmp * tmp;

      return phi_value_max < 0. ? h_star_middle : h_star_right;
    }

    static RYUJIN_DEV double lambda_max(const Params &P, const RiemannData &rd_i,
                                        const RiemannData &rd_j)
    {
      const double h_star = compute_h_star(P, rd_i, rd_j);
      double lambda1, lambda3;
      {
        const double factor = positive_part((h_star - rd_i.h) / rd_i.h);
        const double half_factor = 0.5 * factor;
        lambda1 = rd_i.u - rd_i.a * sqrt((1. + half_factor) * (1. + factor));
      }
      {
        const double factor = positive_part((h_star - rd_j.h) / rd_j.h);
        const double half_factor = 0.5 * factor;
        lambda3 = rd_j.u + rd_j.a * sqrt((1. + half_factor) * (1. + factor));
      }
      return fmax(negative_part(lambda1), positive_part(lambda3));
    }

    static RYUJIN_DEV RiemannData riemann_data_from_state(const Params &P, const double (&U)[K],
                                                          const double (&n)[DIM])
    {
      const double h = water_depth_sharp(P, U);
      double pv = n[0] * (U[1] / h);
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        pv += n[d] * (U[1 + d] / h);
      return {h, pv, sqrt(h * P.gravity)};
    }

    static RYUJIN_DEV double dij_from_states(const Params &P, const double (&U_i)[K],
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
      return norm * lambda_max(P, riemann_data_from_state(P, U_i, n), riemann_data_from_state(P, U_j, n));
    }

    /* ------------------------------------------------------------------ Riemann solver, node records
     * As for Euler (euler_device.hpp): what riemann_data_from_state derives from one state apart from the
     * normal velocity -- h (sharp), the velocity q / h, a = sqrt(g h) -- and sqrt(h) are computed once per node;
     * per pair this removes the 2 dim divisions and 2 square roots of the projections and, in compute_h_star,
     * sqrt(g h_min), sqrt(g h_max), sqrt(h_min h_max), sqrt(h_min / g) and sqrt(h_min), which become products of
     * stored roots. phi(x_0 h_max) is evaluated on its shock branch only: x_0 = (2 sqrt 2 - 1)^2 > 1, so
     * x_0 h_max > h_Z for both states and rs_f never takes the rarefaction branch there.
     * 5 divisions + 5 square roots per pair instead of 9 + 12; results within a few ulp (1e-12 contract on d_ij).
     * record = (h, a, v[DIM]) padded to an even number of doubles (32 bytes); sqrt(h) = a / sqrt(g) */
    static constexpr int RS = (2 + DIM + 1) / 2 * 2;

    static RYUJIN_DEV void riemann_record(const Params &P, const double (&U)[K], double (&rec)[RS])
    {
      const double h = water_depth_sharp(P, U);
      rec[0] = h;
      rec[1] = sqrt(h * P.gravity);
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        rec[2 + d] = U[1 + d] / h;
#pragma unroll
      for (int d = 2 + DIM; d < RS; ++d)
        rec[d] = 0.;
    }

    /* (the record holds the Riemann data only: step 2 reads the state and the precomputed values next to it) */
    static constexpr bool kRecordHoldsState = false;
    static RYUJIN_DEV void node_record(const Params &P, const double (&U)[K], const double2, double (&rec)[RS])
    {
      riemann_record(P, U, rec);
    }

    template <bool GENERAL = false>
    static RYUJIN_DEV double dij_from_records(const Params &P, const double (&ri)[RS], const double (&rj)[RS],
                                              const double (&c)[DIM])
    {
      double norm2 = c[0] * c[0];
      double vc_i = ri[2] * c[0], vc_j = rj[2] * c[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        norm2 += c[d] * c[d];
        vc_i += ri[2 + d] * c[d];
        vc_j += rj[2 + d] * c[d];
      }
      const double norm = sqrt(norm2);
      const double inverse_norm = 1. / norm;
      const double u_i = vc_i * inverse_norm, u_j = vc_j * inverse_norm;
      const double h_i = ri[0], a_i = ri[1], h_j = rj[0], a_j = rj[1];

      /* compute_h_star, riemann_solver.template.h:111-204 */
      const bool i_is_min = h_i <= h_j;
      const double h_min = i_is_min ? h_i : h_j, h_max = i_is_min ? h_j : h_i;
      const double a_min = i_is_min ? a_i : a_j, a_max = i_is_min ? a_j : a_i;
      const double sqrt_two = 1.4142135623730951;
      const double inverse_sqrt_g = sqrt(1. / P.gravity); /* wave-uniform */
      const double sq_min = a_min * inverse_sqrt_g, sq_max = a_max * inverse_sqrt_g; /* sqrt(h) */
      const double x0 = 9. - 4. * sqrt_two;
      double phi_value_max;
      {
        const double h = x0 * h_max; /* > h_i, h_j: shock branch of rs_f for both states */
        const double f_i = (h - h_i) * sqrt(0.5 * P.gravity * (h + h_i) / (h * h_i));
        const double f_j = (h - h_j) * sqrt(0.5 * P.gravity * (h + h_j) / (h * h_j));
        phi_value_max = f_i + f_j + u_j - u_i;
      }
      const double sq_min_max = sq_min * sq_max; /* sqrt(h_min h_max) */
      const double h_star_middle = sq_min_max * (1. + sqrt_two * (u_i - u_j) / (a_min + a_max));
      const double left_radicand = 3. * h_min + 2. * sqrt_two * sq_min_max;
      const double right_radicand = sqrt_two * (sq_min * inverse_sqrt_g) * (u_i - u_j);
      double tmp = sqrt(positive_part(left_radicand + right_radicand));
      tmp -= sqrt_two * sq_min;
      const double h_star_right = tmp * tmp;
      const double h_star = phi_value_max < 0. ? h_star_middle : h_star_right;

      /* lambda_max, :206-251 */
      double lambda1, lambda3;
      {
        const double factor = positive_part((h_star - h_i) / h_i);
        lambda1 = u_i - a_i * sqrt((1. + 0.5 * factor) * (1. + factor));
      }
      {
        const double factor = positive_part((h_star - h_j) / h_j);
        lambda3 = u_j + a_j * sqrt((1. + 0.5 * factor) * (1. + factor));
      }
      return norm * fmax(negative_part(lambda1), positive_part(lambda3));
    }

    /* ------------------------------------------------------------------ Limiter::limit */

    static RYUJIN_DEV double q_dot(const double (&a)[K], const double (&b)[K])
    {
      double s = a[1] * b[1];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        s += a[1 + d] * b[1 + d];
      return s;
    }

    /* limiter.template.h:16-449. bounds = (h_min, h_max, h_small, kin_max, v2_max). The function is cheap (no
     * transcendental beyond two sqrt in the single quadratic Newton step), so the fast/slow split only separates the
     * Newton step. CHECKED: the reference's EXPENSIVE_BOUNDS_CHECK control flow -- the water depth behind the clip
     * (:106-126), the kinetic-energy and square-velocity constraints of the limited state (:242-262, :396-420) --,
     * the same t_l, `success` has more ways to be false (ryujin_hip_params::debug_expensive_bounds_check). */
    template <bool CHECKED>
    static RYUJIN_DEV double limit_impl(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                        const double (&Pij)[K], bool &success)
    {
      constexpr double t_min = 0., t_max = 1.;
      constexpr double eps = DBL_EPSILON;
      const double h_min = bnd[0], h_max = bnd[1], h_small = bnd[2], kin_max = bnd[3],
                   v2_max = bnd[4];
      const double relax_small = 1. + P.dry_small * eps;
      const double relax = 1. + P.dry_large * eps;
      success = true;
      double t_l = t_min, t_r = t_max;
      {
        const double h_U = U[0], h_P = Pij[0];
        const double test_min = filter_dry_water_depth(P, fmax(0., h_U - relax * h_max));
        const double test_max = filter_dry_water_depth(P, fmax(0., h_min - relax * h_U));
        if (!(test_min == 0. && test_max == 0.))
          success = false;
        /* (the division only where a lane clips, as Euler<DIM>::first_psi_r: t_r = t_max stays untouched otherwise) */
        const double h_min_tilde = fmax(h_small, h_min);
        if (h_max < h_U + t_r * h_P || h_U + t_r * h_P < h_min_tilde) {
          const double denominator = 1. / (fabs(h_P) + eps * h_max + DBL_MIN);
          t_r = h_max < h_U + t_r * h_P ? (h_max - h_U) * denominator : t_r;
          t_r = h_U + t_r * h_P < h_min_tilde ? (h_U - h_min_tilde) * denominator : t_r;
          t_r = fmin(t_r, t_max);
          t_r = fmax(t_r, t_min);
        }
        if constexpr (CHECKED) {
          const double h_new = U[0] + t_r * Pij[0];
          const double test_new_min = filter_dry_water_depth(P, fmax(0., h_new - relax * h_max));
          const double test_new_max = filter_dry_water_depth(P, fmax(0., h_min - relax * h_new));
          if (!(test_new_min == 0. && test_new_max == 0.))
            success = false;
        }
      }
      if (!P.limit_on_square_velocity && !P.limit_on_kinetic_energy)
        return t_l;

      if (P.limit_on_kinetic_energy) {
        double U_r[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_r[q] = U[q] + t_r * Pij[q];
        const double psi_r = relax_small * U_r[0] * kin_max - 0.5 * q_dot(U_r, U_r);
        t_l = psi_r > 0. ? t_r : t_l;
        if (!P.limit_on_square_velocity && t_l == t_r)
          return t_l;
        double U_l[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_l[q] = U[q] + t_l * Pij[q];
        const double h_l = U_l[0];
        const double psi_l = relax_small * h_l * kin_max - 0.5 * q_dot(U_l, U_l);
        const double filtered_h_l = filter_dry_water_depth(P, h_l);
        const double lower_bound = (1. - relax) * filtered_h_l * kin_max - eps;
        if (!(fmin(0., psi_l - lower_bound) == 0.))
          success = false;
        if (!(fmax(0., t_r - t_l - P.lim_newton_tolerance) == 0.)) {
          const double h_P = Pij[0];
          const double dpsi_l = h_P * kin_max - q_dot(U, Pij) - q_dot(Pij, Pij) * t_l;
          const double dpsi_r = h_P * kin_max - q_dot(U, Pij) - q_dot(Pij, Pij) * t_r;
          quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
        }
        if constexpr (CHECKED) {
          double U_new[K];
#pragma unroll
          for (int q = 0; q < K; ++q)
            U_new[q] = U[q] + t_l * Pij[q];
          const double psi_new = relax_small * U_new[0] * kin_max - 0.5 * q_dot(U_new, U_new);
          const double lb = (1. - relax) * U_new[0] * kin_max - eps;
          if (!(fmin(0., psi_new - lb) == 0.))
            success = false;
        }
        if (P.limit_on_square_velocity) {
          t_r = t_l;
          t_l = t_min;
        }
      }

      if (P.limit_on_square_velocity) {
        double U_r[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_r[q] = U[q] + t_r * Pij[q];
        const double h_r = U_r[0];
        const double psi_r = relax_small * h_r * h_r * v2_max - q_dot(U_r, U_r);
        t_l = psi_r > 0. ? t_r : t_l;
        if (t_l == t_r)
          return t_l;
        double U_l[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_l[q] = U[q] + t_l * Pij[q];
        const double h_l = U_l[0];
        const double psi_l = relax_small * h_l * h_l * v2_max - q_dot(U_l, U_l);
        const double filtered_h_l = filter_dry_water_depth(P, h_l);
        const double lower_bound = (1. - relax) * filtered_h_l * filtered_h_l * v2_max - 100. * eps;
        if (!(fmin(0., psi_l - lower_bound) == 0.))
          success = false;
        if (!(fmax(0., t_r - t_l - P.lim_newton_tolerance) == 0.)) {
          const double h_U = U[0], h_P = Pij[0];
          const double dpsi_l =
              (h_U + t_l * h_P) * h_P * v2_max - 2. * (q_dot(U, Pij) - q_dot(Pij, Pij) * t_l);
          const double dpsi_r =
              (h_U + t_r * h_P) * h_P * v2_max - 2. * (q_dot(U, Pij) - q_dot(Pij, Pij) * t_r);
          quadratic_newton_step(t_l, t_r, psi_l, psi_r, dpsi_l, dpsi_r, -1.);
        }
        if constexpr (CHECKED) {
          double U_new[K];
#pragma unroll
          for (int q = 0; q < K; ++q)
            U_new[q] = U[q] + t_l * Pij[q];
          const double h_new = U_new[0];
          const double psi_new = relax_small * h_new * h_new * v2_max - q_dot(U_new, U_new);
          const double lb = (1. - relax) * h_new * h_new * v2_max - 100. * eps;
          if (!(fmin(0., psi_new - lb) == 0.))
            success = false;
        }
      }
      return t_l;
    }

    static RYUJIN_DEV double limit(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                   const double (&Pij)[K], bool &success)
    {
      return limit_impl<false>(P, bnd, U, Pij, success);
    }

    static RYUJIN_DEV double limit_checked(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                           const double (&Pij)[K], bool &success)
    {
      return limit_impl<true>(P, bnd, U, Pij, success);
    }

    /* View::is_admissible (shallow_water/hyperbolic_system.h:882-902): the filtered water depth is not negative */
    static RYUJIN_DEV bool is_admissible(const Params &P, const double (&U)[K])
    {
      return filter_dry_water_depth(P, U[0]) >= 0.;
    }

    /* the whole limiter is short: every pair is decided here */
    static RYUJIN_DEV double limit_fast(const Params &P, const double (&bnd)[NB],
                                        const double (&U)[K], const double (&Pij)[K], bool &success,
                                        bool &undecided)
    {
      undecided = false;
      return limit(P, bnd, U, Pij, success);
    }

    /* ------------------------------------------------------------------ boundary conditions */

    template <int component>
    static RYUJIN_DEV void prescribe_riemann_characteristic(const Params &P, const double (&U)[K],
                                                            const double (&U_bar)[K],
                                                            const double (&normal)[DIM],
                                                            double (&U_new)[K])
    {
      const double a = sqrt(P.gravity * U[0]);
      const double ih = inverse_water_depth_sharp(P, U);
      double mn = U[1] * normal[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        mn += U[1 + d] * normal[d];
      const double vn = mn * ih;

      const double a_bar = sqrt(P.gravity * U_bar[0]);
      double mn_bar = U_bar[1] * normal[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        mn_bar += U_bar[1 + d] * normal[d];
      const double vn_bar = mn_bar * inverse_water_depth_sharp(P, U_bar);

      const double R_1 = component == 1 ? vn_bar - 2. * a_bar : vn - 2. * a;
      const double R_2 = component == 2 ? vn_bar + 2. * a_bar : vn + 2. * a;

      const double vn_new = 0.5 * (R_1 + R_2);
      const double tmp = (R_2 - R_1) / 4.;
      const double h_new = tmp * tmp / P.gravity;
      U_new[0] = h_new;
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        const double vperp = U[1 + d] * ih - vn * normal[d];
        U_new[1 + d] = h_new * (vn_new * normal[d] + vperp);
      }
    }

    /* hyperbolic_system.h:954-1017 */
    static RYUJIN_DEV void apply_boundary_conditions(const Params &P, const int id,
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
      } else if (id == RYUJIN_BC_DIRICHLET_MOMENTUM) {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          result[1 + d] = U_D[1 + d];
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
        const double h_inverse = inverse_water_depth_sharp(P, U);
        const double a = sqrt(P.gravity * U[0]);
        double mn = U[1] * normal[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          mn += U[1 + d] * normal[d];
        const double vn = mn * h_inverse;
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

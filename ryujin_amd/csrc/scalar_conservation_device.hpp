// Device-side scalar conservation Description and the sweeps that differ for it.
//
// Restates (operation order preserved):
//   FluxLibrary           source/scalar_conservation/flux_{burgers,kpp,function}.h ("function" restricted
//                         to polynomials, central-difference gradient as dealii::FunctionParser)
//   HyperbolicSystemView  source/scalar_conservation/hyperbolic_system.h:264-480
//   RiemannSolver         source/scalar_conservation/riemann_solver.template.h:21-175 (random entropies = 0)
//   Indicator             source/scalar_conservation/indicator.h:160-205
//   Limiter               source/scalar_conservation/limiter.h:190-290, limiter.template.h:15-110
// One state component (stored padded to 2 doubles), 2*dim precomputed values (f, df), 2 limiter bounds.

#pragma once

#include "kernels_euler.hpp"
#include "ryujin_hip.h"

namespace ryujin_hip
{
  struct ScalarParams {
    int flux, use_greedy_wavespeed, use_averaged_entropy;
    double poly[3][4];
    double delta; /* derivative_approximation_delta */
    double evc_factor, lim_relaxation_factor;
  };

  template <int DIM>
  struct ScalarConservation {
    static constexpr int DIMENSION = DIM;
    static constexpr int K = 1;
    static constexpr int NB = 2;
    static constexpr bool kFusablePrecompute = false; /* (FusedPrecompute: Euler and shallow water only) */
    /* steps 6/7 may form a limited row's update as V_i - sum (1 - l_ij) lambda P_ij (kernels_limiter.hpp): another
     * rounding of the reference's sum. Not where l = 0 has to return the low-order update EXACTLY (a dry node) */
    static constexpr bool kLimitedUpdateFromV = false;
    static constexpr int NPREC = 2 * DIM;
    using Params = ScalarParams;

    static RYUJIN_DEV double polynomial(const Params &P, const double u, const int d)
    {
      return P.poly[d][0] + u * (P.poly[d][1] + u * (P.poly[d][2] + u * P.poly[d][3]));
    }

    static RYUJIN_DEV double flux_value(const Params &P, const double u, const int d)
    {
      switch (P.flux) {
      case RYUJIN_FLUX_BURGERS:
        return 0.5 * u * u;
      case RYUJIN_FLUX_KPP:
        return d == 0 ? sin(u) : cos(u);
      default:
        return polynomial(P, u, d);
      }
    }

    static RYUJIN_DEV double flux_gradient(const Params &P, const double u, const int d)
    {
      switch (P.flux) {
      case RYUJIN_FLUX_BURGERS:
        return u;
      case RYUJIN_FLUX_KPP:
        return d == 0 ? cos(u) : -sin(u);
      default:
        return (polynomial(P, u + P.delta, d) - polynomial(P, u - P.delta, d)) / (2 * P.delta);
      }
    }

    /* precomputed (f[DIM], df[DIM]) */
    static RYUJIN_DEV void load_prec(const double *__restrict__ prec, const uint32_t i, double (&p)[NPREC])
    {
#pragma unroll
      for (int q = 0; q < NPREC; ++q)
        p[q] = prec[(size_t)i * NPREC + q];
    }

    static RYUJIN_DEV double dot(const double (&a)[DIM], const double (&b)[DIM])
    {
      double s = a[0] * b[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        s += a[d] * b[d];
      return s;
    }

    static RYUJIN_DEV double kruzkov_entropy_derivative(const double k, const double u)
    {
      return u >= k ? 1. : -1.;
    }

    /* RiemannSolver::compute */
    static RYUJIN_DEV double lambda_max(const Params &P, const double u_i, const double u_j,
                                        const double (&prec_i)[NPREC], const double (&prec_j)[NPREC],
                                        const double (&n)[DIM])
    {
      double f_i = prec_i[0] * n[0], f_j = prec_j[0] * n[0];
      double df_i = prec_i[DIM] * n[0], df_j = prec_j[DIM] * n[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        f_i += prec_i[d] * n[d];
        f_j += prec_j[d] * n[d];
        df_i += prec_i[DIM + d] * n[d];
        df_j += prec_j[DIM + d] * n[d];
      }
      const double h2 = 2. * P.delta;
      double lambda = fabs(f_i - f_j) / fmax(fabs(u_i - u_j), h2);
      if (P.use_greedy_wavespeed) {
        lambda = fabs(u_i - u_j) >= h2 ? lambda : fabs(0.5 * (df_i + df_j));
      } else {
        lambda = fmax(lambda, fabs(df_i));
        lambda = fmax(lambda, fabs(df_j));
      }
      if (P.use_averaged_entropy) {
        const double k = 0.5 * (u_i + u_j);
        double f_k = flux_value(P, k, 0) * n[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          f_k += flux_value(P, k, d) * n[d];
        const double eta_i = fabs(k - u_i);
        const double q_i = kruzkov_entropy_derivative(k, u_i) * (f_i - f_k);
        const double eta_j = fabs(k - u_j);
        const double q_j = kruzkov_entropy_derivative(k, u_j) * (f_j - f_k);
        const double a = u_i + u_j - 2. * k;
        const double b = f_j - f_i;
        const double c = eta_i + eta_j;
        const double d = q_j - q_i;
        const double lambda_left = fabs(d + b) / (fabs(c + a) + h2);
        const double lambda_right = fabs(d - b) / (fabs(c - a) + h2);
        lambda = fmax(lambda, lambda_left);
        lambda = fmax(lambda, lambda_right);
      }
      return lambda;
    }

    static RYUJIN_DEV double dij_from_states(const Params &P, const double u_i, const double (&prec_i)[NPREC],
                                             const double u_j, const double (&prec_j)[NPREC],
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
      return norm * lambda_max(P, u_i, u_j, prec_i, prec_j, n);
    }

    /* Limiter::limit: a clip, always decided by the "fast" part */
    static RYUJIN_DEV double limit(const Params &, const double (&bnd)[NB], const double (&U)[K],
                                   const double (&Pij)[K], bool &success)
    {
      constexpr double t_min = 0., t_max = 1.;
      constexpr double eps = DBL_EPSILON;
      const double relax = 1. + 10000. * eps;
      success = true;
      double t_r = t_max;
      const double u_U = U[0], u_P = Pij[0];
      const double u_min = bnd[0], u_max = bnd[1];
      const double test_max = fmax(0., fmin(u_U - relax * u_max, relax * u_U - u_max));
      const double test_min = fmax(0., fmin(u_min - relax * u_U, relax * u_min - u_U));
      if (!(test_max == 0. && test_min == 0.))
        success = false;
      const double regularization = 100. * DBL_MIN;
      const double denominator = 1. / fmax(regularization, fabs(u_P) + eps * u_max);
      t_r = u_max < u_U + t_r * u_P ? (u_max - u_U) * denominator : t_r;
      t_r = u_U + t_r * u_P < u_min ? (u_U - u_min) * denominator : t_r;
      t_r = fmin(t_r, t_max);
      t_r = fmax(t_r, t_min);
      return t_r;
    }
    static RYUJIN_DEV double limit_fast(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                        const double (&Pij)[K], bool &success, bool &undecided)
    {
      undecided = false;
      return limit(P, bnd, U, Pij, success);
    }

    /* the EXPENSIVE_BOUNDS_CHECK control flow (scalar_conservation/limiter.template.h): the clipped value itself is
     * tested against the relaxed bounds as well; is_admissible() is trivially true for a scalar */
    static RYUJIN_DEV double limit_checked(const Params &P, const double (&bnd)[NB], const double (&U)[K],
                                           const double (&Pij)[K], bool &success)
    {
      const double t_r = limit(P, bnd, U, Pij, success);
      const double relax = 1. + 10000. * DBL_EPSILON;
      const double u_min = bnd[0], u_max = bnd[1];
      const double u_new = U[0] + t_r * Pij[0];
      const double test_new_max = fmax(0., fmin(u_new - relax * u_max, relax * u_new - u_max));
      const double test_new_min = fmax(0., fmin(u_min - relax * u_new, relax * u_min - u_new));
      if (!(test_new_max == 0. && test_new_min == 0.))
        success = false;
      return t_r;
    }

    static RYUJIN_DEV bool is_admissible(const Params &, const double (&)[K]) { return true; }

    /* apply_boundary_conditions (:381-420): Dirichlet; slip / no_slip / dynamic are rejected by create() */
    static RYUJIN_DEV void apply_boundary_conditions(const Params &, const int id, const double (&U)[K],
                                                     const double (&)[DIM], const double (&U_D)[K],
                                                     double (&result)[K])
    {
      result[0] = id == RYUJIN_BC_DIRICHLET ? U_D[0] : U[0];
    }
  };


  /* ------------------------------------------------------------------ step 1: precomputation_loop */
  template <int DIM, bool WITH_BC>
  __global__ void __launch_bounds__(kBlock)
  k_precompute_sc(const ScalarParams P, const DeviceMesh M, const BcFold B, double *U,
                  double *__restrict__ prec)
  {
    using E = ScalarConservation<DIM>;
    const uint32_t i = M.slice_begin * 64 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.n_owned || i >= M.slice_end * 64)
      return;
    if constexpr (WITH_BC)
      apply_bc_row<E>(P, B, i, U);
    if (M.row_len[i] == 1)
      return;
    const double u = U[(size_t)i * 2];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      prec[(size_t)i * E::NPREC + d] = E::flux_value(P, u, d);
      prec[(size_t)i * E::NPREC + DIM + d] = E::flux_gradient(P, u, d);
    }
  }

  /* ------------------------------------------------------------------ step 2: alpha_i and upper d_ij */
  template <int DIM>
  __global__ void __launch_bounds__(kBlock)
  k_dij_alpha_sc(const ScalarParams P, const DeviceMesh M, const double *__restrict__ U,
                 const double *__restrict__ prec, double *__restrict__ dij, double *__restrict__ alpha)
  {
    using E = ScalarConservation<DIM>;
    constexpr int NP = E::NPREC;
    step_begin(M);
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double u_i = U[(size_t)i * 2];
    double prec_i[NP];
    E::load_prec(prec, i, prec_i);
    double f_i[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d)
      f_i[d] = prec_i[d];
    /* Indicator::reset */
    double u_abs_max = fabs(u_i), left = 0., right = 0.;
    for (uint32_t c = 0; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const uint32_t j = ld_stream(M.cols + pos);
      double c_ij[DIM], prec_j[NP];
      load_entry<DIM>(M.cij, colbase, r.lane, c_ij);
      const double u_j = U[(size_t)j * 2];
      E::load_prec(prec, j, prec_j);
      if (!(row_active && c < r.len))
        continue;
      /* Indicator::accumulate */
      u_abs_max = fmax(u_abs_max, fabs(u_j));
      const double d_eta_j = E::kruzkov_entropy_derivative(u_i, u_j);
      double f_j[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        f_j[d] = prec_j[d];
      left += d_eta_j * E::dot(f_j, c_ij);
      right += d_eta_j * E::dot(f_i, c_ij);
      if (c > 0 && j > i)
        dij[pos] = E::dij_from_states(P, u_i, prec_i, u_j, prec_j, c_ij);
    }
    if (row_active) {
      const double hd_i = M.mi[i] * M.measure_of_omega_inverse;
      const double numerator = left - right;
      const double denominator = fabs(left) + fabs(right);
      const double regularization = 100. * DBL_MIN;
      const double quotient =
          fabs(numerator) / (denominator + fmax(hd_i * fabs(u_abs_max), regularization));
      alpha[i] = fmin(1., P.evc_factor * quotient);
    }
  }

  /* step 3, boundary pairs */
  template <int DIM>
  __global__ void __launch_bounds__(kBlock)
  k_dij_boundary_sc(const ScalarParams P, const uint32_t n_pairs, const uint32_t *__restrict__ p_i,
                    const uint32_t *__restrict__ p_j, const uint32_t *__restrict__ p_pos,
                    const double *__restrict__ cji, const double *__restrict__ U,
                    const double *__restrict__ prec, double *__restrict__ dij)
  {
    using E = ScalarConservation<DIM>;
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pairs)
      return;
    const uint32_t i = p_i[q], j = p_j[q];
    if (j < i)
      return;
    double prec_i[E::NPREC], prec_j[E::NPREC], c_ji[DIM];
    E::load_prec(prec, i, prec_i);
    E::load_prec(prec, j, prec_j);
#pragma unroll
    for (int d = 0; d < DIM; ++d)
      c_ji[d] = cji[(size_t)q * DIM + d];
    const double d_ji = E::dij_from_states(P, U[(size_t)j * 2], prec_j, U[(size_t)i * 2], prec_i, c_ji);
    const uint32_t pos = p_pos[q];
    dij[pos] = fmax(dij[pos], d_ji);
  }

  /* ------------------------------------------------------------------ step 4 */
  template <int DIM, bool HAS_STAGES, bool DG = false>
  __global__ void __launch_bounds__(kBlock)
  k_low_order_sc(const ScalarParams P, const DeviceMesh M, DeviceScalars *scalars,
                 const double weight, const StageArgs<DIM> S, const double *__restrict__ U,
                 const double *__restrict__ prec, const double *__restrict__ alpha,
                 const double *__restrict__ dij, double *__restrict__ new_U, double *__restrict__ r_out,
                 double *__restrict__ bounds, double *__restrict__ pij)
  {
    using E = ScalarConservation<DIM>;
    constexpr int NP = E::NPREC;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double tau = finalize_tau(scalars);
    const double u_i = U[(size_t)i * 2];
    double u_i_new = u_i, F_iH = 0.;
    const double alpha_i = alpha[i];
    const double m_i = M.mi[i];
    const double m_i_inv = M.mi_inv[i];
    double f_i[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d)
      f_i[d] = prec[(size_t)i * NP + d];

    /* Limiter::reset */
    double u_min = DBL_MAX, u_max = -DBL_MAX;
    double u_relaxation_numerator = 0., u_relaxation_denominator = 0.;

    for (uint32_t c = 0; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const uint32_t j = ld_stream(M.cols + pos);
      double c_ij[DIM], f_j[DIM];
      load_entry<DIM>(M.cij, colbase, r.lane, c_ij);
      const double d_ij = dij[pos];
      const double u_j = U[(size_t)j * 2];
      const double alpha_j = alpha[j];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        f_j[d] = prec[(size_t)j * NP + d];
      if (!(row_active && c < r.len))
        continue;

      double factor = (alpha_i + alpha_j) * .5;
      if constexpr (DG) /* hyperbolic_module.template.h:733-737 */
        factor = fmax(factor, M.incidence[colbase * 64 + r.lane]);
      const double d_ijH = d_ij * factor;
      const double denom = fmax(d_ij, 100. * DBL_MIN);
      double scaled_c_ij[DIM];
      const double inverse_denom = 1. / denom; /* dealii::Tensor / scalar multiplies by the inverse */
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        scaled_c_ij[d] = c_ij[d] * inverse_denom;

      /* flux_divergence = -contract(add(flux_i, flux_j), c_ij) */
      double s = (f_i[0] + f_j[0]) * c_ij[0];
#pragma unroll
      for (int d = 1; d < DIM; ++d)
        s += (f_i[d] + f_j[d]) * c_ij[d];
      const double flux_ij = -s;

      u_i_new += tau * m_i_inv * flux_ij;
      double P_ij = -flux_ij;
      const double dU = u_j - u_i;
      u_i_new += tau * m_i_inv * d_ij * dU;
      F_iH += d_ijH * dU;
      P_ij += (d_ijH - d_ij) * dU;

      /* Limiter::accumulate */
      {
        double contracted = (f_j[0] + (-f_i[0])) * scaled_c_ij[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          contracted += (f_j[d] + (-f_i[d])) * scaled_c_ij[d];
        const double u_ij_bar = 0.5 * (u_i + u_j) - 0.5 * contracted + 0.;
        u_min = fmin(u_min, u_ij_bar);
        u_max = fmax(u_max, u_ij_bar);
        u_relaxation_numerator += 1. * (u_i + u_j);
        u_relaxation_denominator += 1.;
      }

      F_iH += weight * flux_ij;
      P_ij += weight * flux_ij;

      if constexpr (HAS_STAGES) {
        for (int st = 0; st < S.stages; ++st) {
          double ss = (S.prec[st][(size_t)i * NP] + S.prec[st][(size_t)j * NP]) * c_ij[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            ss += (S.prec[st][(size_t)i * NP + d] + S.prec[st][(size_t)j * NP + d]) * c_ij[d];
          const double flux_s = -ss;
          F_iH += S.w[st] * flux_s;
          P_ij += S.w[st] * flux_s;
        }
      }
      pij[pos] = P_ij;
    }

    if (!row_active)
      return;
    {
      double2 v;
      v.x = u_i_new;
      v.y = 0.;
      reinterpret_cast<double2 *>(new_U)[i] = v;
      v.x = F_iH;
      reinterpret_cast<double2 *>(r_out)[i] = v;
    }

    /* Limiter::bounds */
    const double hd_i = m_i * M.measure_of_omega_inverse;
    double r_i = sqrt(hd_i);
    if constexpr (DIM == 2) {
      const double t = sqrt(r_i);
      r_i = t * t * t;
    } else if constexpr (DIM == 1) {
      r_i = r_i * r_i * r_i;
    }
    r_i *= P.lim_relaxation_factor;
    const double u_relaxation =
        fabs(u_relaxation_numerator) / (fabs(u_relaxation_denominator) + DBL_EPSILON);
    const double u_min_r =
        fmax(fmin((1. - r_i) * u_min, (1. + r_i) * u_min), u_min - 2. * u_relaxation);
    const double u_max_r =
        fmin(fmax((1. + r_i) * u_max, (1. - r_i) * u_max), u_max + 2. * u_relaxation);
    const size_t stride = M.bounds_stride;
    bounds[i] = u_min_r;
    bounds[stride + i] = u_max_r;
  }
} // namespace ryujin_hip

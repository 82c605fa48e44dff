// Sweeps of HyperbolicModule::prepare_state_vector / ::step that differ for the EulerAEOS Description
// (source/euler_aeos/): four precomputed values with two precomputation cycles, Riemann data and fluxes
// from the precomputed EOS pressure, limiter bounds (rho_min, rho_max, s_min, gamma_min) with the surrogate
// entropy evaluated for the row's gamma_min. Steps 3, 5, 6, 7 and the boundary kernel are the generic ones
// (kernels_euler.hpp, kernels_limiter.hpp) instantiated with EulerAeos<DIM>.
// Same thread mapping as everywhere: one row per lane, one SELL-64 slice per wave.

#pragma once

#include "euler_aeos_device.hpp"
#include "kernels_euler.hpp"

namespace ryujin_hip
{
  /* precomputation_loop cycle 0 (hyperbolic_system.h:919-938): p_i from the equation of state,
   * surrogate gamma_i */
  template <int DIM, bool WITH_BC>
  __global__ void __launch_bounds__(kBlock)
  k_precompute_aeos0(const EulerAeosParams P, const DeviceMesh M, const BcFold B, double *U,
                     double *__restrict__ prec, double *__restrict__ rec, double *__restrict__ gamma)
  {
    using E = EulerAeos<DIM>;
    constexpr int K = E::K, RS = E::RS;
    const uint32_t i = M.slice_begin * 64 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.n_owned || i >= M.slice_end * 64)
      return;
    if constexpr (WITH_BC)
      apply_bc_row<E>(P, B, i, U);
    if (M.row_len[i] == 1)
      return;
    double U_i[K], rr[RS];
    load_state<K>(U, i, U_i);
    const auto prec_i = E::precompute_cycle0(P, U_i);
    E::store_prec(prec, i, prec_i);
    /* the node's Riemann record for step 2 (EulerAeos::riemann_record) */
    E::riemann_record(P, U_i, prec_i.p, rr);
#pragma unroll
    for (int g = 0; g < RS; ++g)
      rec[(size_t)i * RS + g] = rr[g];
    gamma[i] = rr[2]; /* once more as a dense vector: cycle 1 gathers it over the stencil */
  }

  /* Riemann records of the ghost rows, from the exchanged ghost states and pressures (functions of (U_j, p_j)
   * alone): behind the exchange of the precomputed values of cycle 0, on the stream of the exchange */
  template <int DIM>
  __global__ void __launch_bounds__(kBlock)
  k_ghost_records_aeos(const EulerAeosParams P, const uint32_t first, const uint32_t last,
                       const double *__restrict__ U, const double *__restrict__ prec, double *__restrict__ rec,
                       double *__restrict__ gamma)
  {
    using E = EulerAeos<DIM>;
    constexpr int K = E::K, RS = E::RS;
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last)
      return;
    double U_i[K], rr[RS];
    load_state<K>(U, i, U_i);
    E::riemann_record(P, U_i, prec[(size_t)i * 4], rr);
#pragma unroll
    for (int g = 0; g < RS; ++g)
      rec[(size_t)i * RS + g] = rr[g];
    gamma[i] = rr[2];
  }

  /* cycle 1 (:942-975): gamma_min over the stencil, then s_i and eta_i for that gamma_min. The reference
   * recomputes the neighbours' gamma_j from (U_j, p_j) (slot 1 of a neighbour may already hold its minimum);
   * here gamma_j = surrogate_gamma(U_j, p_j) was left in a dense vector by cycle 0 (and by k_ghost_records_aeos for
   * the ghost rows) -- the same function on the same arguments: one coalesced 8-byte gather per neighbour instead
   * of two 32-byte ones and a division. A row writes slots 1-3 of its own entry only: the sweep runs in place
   * (prec_out == prec_in). */
  template <int DIM>
  __global__ void __launch_bounds__(kBlock)
  k_precompute_aeos1(const EulerAeosParams P, const DeviceMesh M, const double *__restrict__ U,
                     const double *__restrict__ gamma, const double *prec_in, double *prec_out)
  {
    using E = EulerAeos<DIM>;
    constexpr int K = E::K;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    double U_i[K];
    load_state<K>(U, i, U_i);
    typename E::Prec prec_i = E::load_prec(prec_in, i);
    double gamma_min_i = prec_i.gamma_min;
    /* four columns at a time: the index loads, then the dependent gathers, are in flight together (one column
     * after the other the sweep was bound by 2 x 8 memory latencies per wave) */
    for (uint32_t c = 1; c < r.width; c += 4) {
      uint32_t j[4];
      double gamma_j[4];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        j[u] = c + u < r.width ? ld_stream(M.cols + (((uint64_t)r.base + c + u) * 64 + r.lane)) : i;
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        gamma_j[u] = gamma[j[u]];
#pragma unroll
      for (uint32_t u = 0; u < 4; ++u)
        if (row_active && c + u < r.len)
          gamma_min_i = fmin(gamma_min_i, gamma_j[u]);
    }
    if (!row_active)
      return;
    prec_i.gamma_min = gamma_min_i;
    prec_i.s = E::surrogate_specific_entropy(P, U_i, gamma_min_i);
    prec_i.eta = E::surrogate_harten_entropy(P, U_i, gamma_min_i);
    E::store_prec(prec_out, i, prec_i);
  }

  /* step 2a: indicator (hyperbolic_module.template.h:341-424 with euler_aeos/indicator.h) */
  template <int DIM>
  __global__ void __launch_bounds__(kBlock)
  k_alpha_aeos(const EulerAeosParams P, const DeviceMesh M, const double *__restrict__ U,
               const double *__restrict__ prec, double *__restrict__ alpha)
  {
    using E = EulerAeos<DIM>;
    constexpr int K = E::K;
    step_begin(M);
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;

    double U_i[K];
    load_state<K>(U, i, U_i);
    typename E::Indicator indicator;
    indicator.reset(P, U_i, E::load_prec(prec, i));

    uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
    double c_n[DIM], U_n[K];
    load_entry<DIM>(cij, r.base, r.lane, c_n);
    load_state<K>(U, j_n, U_n);
    for (uint32_t c = 0; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      double c_ij[DIM], U_j[K];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        c_ij[d] = c_n[d];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_j[q] = U_n[q];
      if (c + 1 < r.width) {
        j_n = ld_stream(cols + ((colbase + 1) * 64 + r.lane));
        load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
        load_state<K>(U, j_n, U_n);
      }
      /* column 0 is the row itself: eta_j = eta_i and f_j = f_i bit for bit, its terms are exact zeros */
      if (row_active && c < r.len && c > 0)
        indicator.accumulate(P, U_j, c_ij);
    }
    if (row_active)
      alpha[i] = indicator.alpha(P, M.mi[i] * M.measure_of_omega_inverse);
  }

  /* step 2b: upper-triangular d_ij from the per-node Riemann records (EulerAeos::dij_from_records) */
  template <int DIM>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_DIJ)
  k_dij_aeos(const EulerAeosParams P, const DeviceMesh M, const uint32_t *__restrict__ lower_mask,
             const double *__restrict__ rec, double *__restrict__ dij)
  {
    using E = EulerAeos<DIM>;
    constexpr int RS = E::RS;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const uint32_t upper =
        row_active ? (~lower_mask[r.row] & (r.len >= 32 ? 0xFFFFFFFFu : ((1u << r.len) - 1u)) & ~1u) : 0u;

    double rec_i[RS];
    load_state<RS>(rec, i, rec_i);
    for (uint32_t c = 1; c < r.width; ++c) {
      const bool mine = (upper >> c) & 1u;
      if (!__any(mine))
        continue;
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const uint32_t j = ld_stream(M.cols + pos);
      double c_ij[DIM], rec_j[RS];
      load_entry<DIM>(M.cij, colbase, r.lane, c_ij);
      load_state<RS>(rec, j, rec_j);
      if (mine)
        dij[pos] = E::dij_from_records(P, rec_i, rec_j, c_ij);
    }
  }

  /* step 3, boundary pairs (:462-490) */
  template <int DIM>
  __global__ void __launch_bounds__(kBlock)
  k_dij_boundary_aeos(const EulerAeosParams P, const uint32_t n_pairs, const uint32_t *__restrict__ p_i,
                      const uint32_t *__restrict__ p_j, const uint32_t *__restrict__ p_pos,
                      const double *__restrict__ cji, const double *__restrict__ rec, double *__restrict__ dij)
  {
    using E = EulerAeos<DIM>;
    constexpr int RS = E::RS;
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pairs)
      return;
    const uint32_t i = p_i[q], j = p_j[q];
    if (j < i)
      return;
    double rec_i[RS], rec_j[RS], c_ji[DIM];
    load_state<RS>(rec, i, rec_i);
    load_state<RS>(rec, j, rec_j);
#pragma unroll
    for (int d = 0; d < DIM; ++d)
      c_ji[d] = cji[(size_t)q * DIM + d];
    const double d_ji = E::dij_from_records(P, rec_j, rec_i, c_ji);
    const uint32_t pos = p_pos[q];
    dij[pos] = fmax(dij[pos], d_ji);
  }

  /* step 4 (:597-884) with Limiter::{reset,accumulate,bounds} of euler_aeos/limiter.h:258-410 */
  /* STORE_P = false (stages == 0): P_ij is formed in step 5 (kernels_limiter_stage0.hpp), nothing is stored here */
  /* DG: discontinuous ansatz, the incidence matrix enters the high-order viscosity (hyperbolic_module.template.h:733-737) */
  template <int DIM, bool HAS_STAGES, bool STORE_P = true, bool DG = false>
  __global__ void __launch_bounds__(kBlock, (DIM == 3 && RYUJIN_OCC_LOW_3D_STAGES) ? 1 : RYUJIN_OCC_LOW_AEOS)
  k_low_order_aeos(const EulerAeosParams P, const DeviceMesh M, DeviceScalars *scalars,
                   const double weight, const StageArgs<DIM> S, const double *__restrict__ U,
                   const double *__restrict__ prec, const double *__restrict__ alpha,
                   const double *__restrict__ dij, double *__restrict__ new_U, double *__restrict__ r_out,
                   double *__restrict__ bounds, double *__restrict__ pij)
  {
    using E = EulerAeos<DIM>;
    constexpr int K = E::K;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double tau = finalize_tau(scalars);

    double U_i[K], U_i_new[K], F_iH[K];
    load_state<K>(U, i, U_i);
#pragma unroll
    for (int q = 0; q < K; ++q) {
      U_i_new[q] = U_i[q];
      F_iH[q] = 0.;
    }
    const double alpha_i = alpha[i];
    const double m_i = M.mi[i];
    const double m_i_inv = M.mi_inv[i];
    const typename E::Prec prec_i = E::load_prec(prec, i);
    double f_i[K][DIM];
    E::flux(U_i, prec_i.p, f_i); /* flux_contribution = f(U_i, p_i): hyperbolic_system.h:1405-1416 */

    /* Limiter::reset (limiter.h:258-284) */
    double rho_min = DBL_MAX, rho_max = 0., s_min = DBL_MAX;
    const double gamma_min = prec_i.gamma_min;
    double rho_relaxation_numerator = 0., rho_relaxation_denominator = 0., s_interp_max = 0.;

    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;
    uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
    double c_n[DIM], U_n[K];
    load_entry<DIM>(cij, r.base, r.lane, c_n);
    double d_n = dij[(uint64_t)r.base * 64 + r.lane];
    load_state<K>(U, j_n, U_n);
    double alpha_n = alpha[j_n];
    double p_n = prec[(size_t)j_n * 4 + 0];
    double s_n = prec[(size_t)j_n * 4 + 2];

    for (uint32_t c = 0; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const bool active = row_active && c < r.len;
      const uint32_t j = j_n;
      double c_ij[DIM], U_j[K];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        c_ij[d] = c_n[d];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_j[q] = U_n[q];
      const double d_ij = d_n, alpha_j = alpha_n, p_j = p_n, s_j = s_n;
      if (c + 1 < r.width) {
        j_n = ld_stream(cols + ((colbase + 1) * 64 + r.lane));
        load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
        d_n = dij[(colbase + 1) * 64 + r.lane];
        load_state<K>(U, j_n, U_n);
        alpha_n = alpha[j_n];
        p_n = prec[(size_t)j_n * 4 + 0];
        s_n = prec[(size_t)j_n * 4 + 2];
      }
      if (!active)
        continue;

      double factor = (alpha_i + alpha_j) * .5;
      if constexpr (DG)
        factor = fmax(factor, M.incidence[colbase * 64 + r.lane]);
      const double d_ijH = d_ij * factor;

      const double regularization = 100. * DBL_MIN;
      const double denom = fmax(d_ij, regularization);
      double scaled_c_ij[DIM];
      const double inverse_denom = 1. / denom; /* dealii::Tensor / scalar multiplies by the inverse */
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        scaled_c_ij[d] = c_ij[d] * inverse_denom;

      double f_j[K][DIM];
      E::flux(U_j, p_j, f_j);
      double flux_ij[K];
      E::flux_divergence(f_i, f_j, c_ij, flux_ij);

      double P_ij[K];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        U_i_new[q] += tau * m_i_inv * flux_ij[q];
        P_ij[q] = -flux_ij[q];
      }
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double dU = U_j[q] - U_i[q];
        U_i_new[q] += tau * m_i_inv * d_ij * dU;
        F_iH[q] += d_ijH * dU;
        P_ij[q] += (d_ijH - d_ij) * dU;
      }

      /* Limiter::accumulate (limiter.h:287-353) */
      {
        const double rho_i = U_i[0], rho_j = U_j[0];
        double U_ij_bar[K], U_avg[K];
#pragma unroll
        for (int q = 0; q < K; ++q) {
          double contracted = (f_j[q][0] + (-f_i[q][0])) * scaled_c_ij[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            contracted += (f_j[q][d] + (-f_i[q][d])) * scaled_c_ij[d];
          U_ij_bar[q] = 0.5 * (U_i[q] + U_j[q]) - 0.5 * contracted + 0.;
          U_avg[q] = (U_i[q] + U_j[q]) * .5;
        }
        const double rho_ij_bar = U_ij_bar[0];
        rho_min = fmin(rho_min, rho_ij_bar);
        rho_max = fmax(rho_max, rho_ij_bar);
        rho_relaxation_numerator += 1. * (rho_i + rho_j);
        rho_relaxation_denominator += 1.;
        /* column 0 (j = i): U_ij_bar = U_avg = U_j = U_i exactly, and s(U_i, gamma_min) is the precomputed s_i
         * (cycle 1 of the precomputation evaluates the same function on the same arguments): three powers less
         * per row, the same bits. c is wave-uniform. */
        const double s_ij_bar = c == 0 ? prec_i.s : E::surrogate_specific_entropy(P, U_ij_bar, gamma_min);
        if (P.strict) {
          const double s_j_strict = c == 0 ? prec_i.s : E::surrogate_specific_entropy(P, U_j, gamma_min);
          const double s_interp = c == 0 ? prec_i.s : E::surrogate_specific_entropy(P, U_avg, gamma_min);
          s_min = fmin(s_min, s_j_strict);
          s_min = fmin(s_min, s_ij_bar);
          s_interp_max = fmax(s_interp_max, s_interp);
        } else {
          s_min = fmin(s_min, s_j);
          s_min = fmin(s_min, s_ij_bar);
          s_interp_max = fmax(s_interp_max, s_ij_bar);
        }
      }

#pragma unroll
      for (int q = 0; q < K; ++q) {
        F_iH[q] += weight * flux_ij[q];
        P_ij[q] += weight * flux_ij[q];
      }

      if constexpr (HAS_STAGES) {
        for (int s = 0; s < S.stages; ++s) {
          double U_iHs[K], U_jHs[K];
          load_state<K>(S.U[s], i, U_iHs);
          load_state<K>(S.U[s], j, U_jHs);
          double f_iHs[K][DIM], f_jHs[K][DIM];
          E::flux(U_iHs, S.prec[s][(size_t)i * 4], f_iHs);
          E::flux(U_jHs, S.prec[s][(size_t)j * 4], f_jHs);
          double flux_s[K];
          E::flux_divergence(f_iHs, f_jHs, c_ij, flux_s);
          const double w = S.w[s];
#pragma unroll
          for (int q = 0; q < K; ++q) {
            F_iH[q] += w * flux_s[q];
            P_ij[q] += w * flux_s[q];
          }
        }
      }

      if constexpr (STORE_P)
        store_entry<K>(pij, colbase, r.lane, P_ij);
    }

    if (!row_active)
      return;

    store_state<K>(new_U, i, U_i_new);
    store_state<K>(r_out, i, F_iH);

    /* Limiter::bounds (limiter.h:356-410) */
    const double hd_i = m_i * M.measure_of_omega_inverse;
    double r_i = sqrt(hd_i);
    if constexpr (DIM == 2) {
      const double t = sqrt(r_i);
      r_i = t * t * t;
    } else if constexpr (DIM == 1) {
      r_i = r_i * r_i * r_i;
    }
    r_i *= P.lim_relaxation_factor;
    const double rho_relaxation =
        fabs(rho_relaxation_numerator) / (fabs(rho_relaxation_denominator) + DBL_EPSILON);
    const double relaxation = (2. * P.lim_relaxation_factor) * rho_relaxation;
    const double rho_min_r = fmax((1. - r_i) * rho_min, rho_min - relaxation);
    double rho_max_r = fmin((1. + r_i) * rho_max, rho_max + relaxation);
    const double entropy_relaxation = P.lim_relaxation_factor * (s_interp_max - s_min);
    const double s_min_r = fmax((1. - r_i) * s_min, s_min - entropy_relaxation);
    const double numerator = (gamma_min + 1.) * rho_max_r;
    const double denominator = gamma_min - 1. + 2. * P.b * rho_max_r;
    const double upper_bound = numerator / denominator;
    rho_max_r = fmin(upper_bound, rho_max_r);

    const size_t stride = M.bounds_stride;
    bounds[i] = rho_min_r;
    bounds[stride + i] = rho_max_r;
    bounds[2 * stride + i] = s_min_r;
    bounds[3 * stride + i] = gamma_min;
  }
} // namespace ryujin_hip

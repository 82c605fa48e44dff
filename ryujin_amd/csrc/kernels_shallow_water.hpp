// Step 4 of HyperbolicModule::step for the shallow-water equations: the module-level special
// cases of source/hyperbolic_module.template.h (:270-271 shallow_water flag, :660-720 sources and
// affine-shift pre-loop, :773-787 equilibrated states, :797-846 source terms / high-order flux).
// Steps 1-3 and 5-7 are the generic kernels instantiated with the ShallowWater<DIM> policy.

#pragma once

#include "kernels_euler.hpp"
#include "shallow_water_device.hpp"


#ifndef RYUJIN_SW_SINGLE_WALK
#define RYUJIN_SW_SINGLE_WALK 1 /* A/B on MI355X: see DESIGN.md section 3 */
#endif
#ifndef RYUJIN_OCC_LOW_SW
#define RYUJIN_OCC_LOW_SW 2 /* waves per SIMD asked of the register allocator for the single-walk kernel */
#endif

namespace ryujin_hip
{
  /* DG: discontinuous ansatz, the incidence matrix enters the high-order viscosity (:733-737) */
  template <int DIM, bool HAS_STAGES, bool DG = false>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_LOW_SW)
  k_low_order_sw(const ShallowWaterParams P, const DeviceMesh M,
                 DeviceScalars *scalars, const double weight,
                 const StageArgs<DIM> S, const double *__restrict__ U,
                 const double *__restrict__ prec, const double *__restrict__ Z,
                 const double *__restrict__ alpha, const double *__restrict__ dij,
                 double *__restrict__ new_U, double *__restrict__ r_out,
                 double *__restrict__ bounds, double *__restrict__ pij)
  {
    using E = ShallowWater<DIM>;
    constexpr int K = E::K;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double tau = finalize_tau(scalars);
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;
    const double *__restrict__ mij = M.mij;

    double U_i[K], U_i_new[K], F_iH[K], S_iH[K], S_i[K];
    load_state<K>(U, i, U_i);
    const double alpha_i = alpha[i];
    const double m_i = M.mi[i];
    const double m_i_inv = M.mi_inv[i];
    const double Z_i = Z[i];

#pragma unroll
    for (int q = 0; q < K; ++q)
      S_iH[q] = 0.;
    if constexpr (HAS_STAGES) {
      for (int s = 0; s < S.stages; ++s) {
        double U_iHs[K], Ss[K];
        load_state<K>(S.U[s], i, U_iHs);
        E::manning_friction(P, U_iHs, S.prec[s][(size_t)i * 2 + 1], tau, Ss);
#pragma unroll
        for (int q = 0; q < K; ++q)
          S_iH[q] += S.w[s] * Ss[q];
      }
    }
    E::manning_friction(P, U_i, prec[(size_t)i * 2 + 1], tau, S_i);
#pragma unroll
    for (int q = 0; q < K; ++q) {
      S_iH[q] += weight * S_i[q];
      U_i_new[q] = U_i[q];
      U_i_new[q] += tau * S_i[q];
      F_iH[q] = 0.;
      F_iH[q] += m_i * S_iH[q];
    }

    /* affine shift pre-loop (:700-720, shallow_water/hyperbolic_system.h:1176-1191) */
    double affine_shift[K];
#pragma unroll
    for (int q = 0; q < K; ++q)
      affine_shift[q] = 0.;
    {
      const double h_inverse = E::inverse_water_depth_sharp(P, U_i);
      /* software pipeline: the next column's index, c_ij, d_ij and the gathered Z_j are in flight while
       * this column is evaluated (as k_low_order) */
      uint32_t j_n = cols[(uint64_t)r.base * 64 + r.lane];
      double c_n[DIM];
      load_entry<DIM>(cij, r.base, r.lane, c_n);
      double d_n = dij[(uint64_t)r.base * 64 + r.lane];
      double Z_n = Z[j_n];
      for (uint32_t c = 0; c < r.width; ++c) {
        const uint64_t colbase = (uint64_t)r.base + c;
        double c_ij[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          c_ij[d] = c_n[d];
        const double d_ij = d_n, Z_j = Z_n;
        if (c + 1 < r.width) {
          j_n = cols[(colbase + 1) * 64 + r.lane];
          load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
          d_n = dij[(colbase + 1) * 64 + r.lane];
          Z_n = Z[j_n];
        }
        if (!(row_active && c < r.len))
          continue;
        double U_star_ij[K];
        E::star_state(P, U_i, Z_i, Z_j, U_star_ij);
        double m_c = U_i[1] * c_ij[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          m_c += U_i[1 + d] * c_ij[d];
        const double factor = 2. * (d_ij + h_inverse * m_c);
#pragma unroll
        for (int q = 0; q < K; ++q)
          affine_shift[q] += -factor * (U_star_ij[q] - U_i[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
      affine_shift[q] *= tau * m_i_inv;
      affine_shift[q] += tau * S_i[q];
    }

    /* Limiter::reset (shallow_water/limiter.h:247-268) */
    double h_min = DBL_MAX, h_max = 0., kin_max = 0., v2_max = 0.;
    double h_relaxation_numerator = 0., kin_relaxation_numerator = 0., v2_relaxation_numerator = 0.,
           relaxation_denominator = 0.;
    const double kin_i = E::kinetic_energy(P, U_i);
    double v2_i;
    {
      const double ihm_i = E::inverse_water_depth_mollified(P, U_i);
      const double v = U_i[1] * ihm_i;
      v2_i = v * v;
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        const double vd = U_i[1 + d] * ihm_i;
        v2_i += vd * vd;
      }
    }

    uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
    uint32_t j_nn = r.width > 1 ? ld_stream(cols + (((uint64_t)r.base + 1) * 64 + r.lane)) : i;
    double c_n[DIM], U_n[K];
    load_entry<DIM>(cij, r.base, r.lane, c_n);
    double d_n = dij[(uint64_t)r.base * 64 + r.lane];
    double m_n = ld_stream(mij + ((uint64_t)r.base * 64 + r.lane));
    load_state<K>(U, j_n, U_n);
    double alpha_n = alpha[j_n];
    double Z_n = Z[j_n];
    const bool friction = P.manning != 0.; /* h_star_j only enters the friction term */
    double h_star_n = friction ? prec[(size_t)j_n * 2 + 1] : 1.;
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
      const double d_ij = d_n, m_ij = m_n, alpha_j = alpha_n, Z_j = Z_n, h_star_j = h_star_n;
      if (c + 1 < r.width) {
        j_n = j_nn;
        load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
        d_n = dij[(colbase + 1) * 64 + r.lane];
        m_n = ld_stream(mij + ((colbase + 1) * 64 + r.lane));
        load_state<K>(U, j_n, U_n);
        alpha_n = alpha[j_n];
        Z_n = Z[j_n];
        h_star_n = friction ? prec[(size_t)j_n * 2 + 1] : 1.;
        j_nn = (c + 2 < r.width) ? ld_stream(cols + ((colbase + 2) * 64 + r.lane)) : i;
      }
      if (!active)
        continue;

      double factor = (alpha_i + alpha_j) * .5;
      if constexpr (DG)
        factor = fmax(factor, M.incidence[colbase * 64 + r.lane]);
      const double d_ijH = d_ij * factor;
      const double denom = fmax(d_ij, 100. * DBL_MIN);
      double scaled_c_ij[DIM];
      const double inverse_denom = 1. / denom; /* dealii::Tensor / scalar multiplies by the inverse */
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        scaled_c_ij[d] = c_ij[d] * inverse_denom;

      double U_star_ij[K], U_star_ji[K];
      E::star_state(P, U_i, Z_i, Z_j, U_star_ij);
      E::star_state(P, U_j, Z_j, Z_i, U_star_ji);

      double flux_ij[K];
      E::flux_divergence(P, U_i, U_star_ij, U_star_ji, c_ij, flux_ij);

      double P_ij[K];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        U_i_new[q] += tau * m_i_inv * flux_ij[q];
        P_ij[q] = -flux_ij[q];
      }
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double dU = U_star_ji[q] - U_star_ij[q];
        U_i_new[q] += tau * m_i_inv * d_ij * dU;
        F_iH[q] += d_ijH * dU;
        P_ij[q] += (d_ijH - d_ij) * dU;
      }

      /* Limiter::accumulate (shallow_water/limiter.h:271-330) */
      {
        double f_star_ij[K][DIM], f_star_ji[K][DIM], U_ij_bar[K];
        E::f(P, U_star_ij, f_star_ij);
        E::f(P, U_star_ji, f_star_ji);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          double s = (f_star_ij[q][0] + (-f_star_ji[q][0])) * scaled_c_ij[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            s += (f_star_ij[q][d] + (-f_star_ji[q][d])) * scaled_c_ij[d];
          U_ij_bar[q] = 0.5 * (U_star_ij[q] + U_star_ji[q] + s) + affine_shift[q];
        }
        const double h_bar_ij = U_ij_bar[0];
        h_min = fmin(h_min, h_bar_ij);
        h_max = fmax(h_max, h_bar_ij);
        kin_max = fmax(kin_max, E::kinetic_energy(P, U_ij_bar));
        {
          const double ihm = E::inverse_water_depth_mollified(P, U_ij_bar);
          const double v = U_ij_bar[1] * ihm;
          double v2 = v * v;
#pragma unroll
          for (int d = 1; d < DIM; ++d) {
            const double vd = U_ij_bar[1 + d] * ihm;
            v2 += vd * vd;
          }
          v2_max = fmax(v2_max, v2);
        }
        relaxation_denominator += 1.;
        h_relaxation_numerator += 1. * (U_i[0] + U_j[0]);
        kin_relaxation_numerator += 1. * (kin_i + E::kinetic_energy(P, U_j));
        double v2_j;
        {
          const double ihm_j = E::inverse_water_depth_mollified(P, U_j);
          const double v = U_j[1] * ihm_j;
          v2_j = v * v;
#pragma unroll
          for (int d = 1; d < DIM; ++d) {
            const double vd = U_j[1 + d] * ihm_j;
            v2_j += vd * vd;
          }
        }
        v2_relaxation_numerator += 1. * (-v2_i + v2_j);
      }

#pragma unroll
      for (int q = 0; q < K; ++q) {
        F_iH[q] -= m_ij * S_iH[q];
        P_ij[q] -= m_ij * /*sic!*/ S_i[q];
      }
      {
        double hof[K];
        E::high_order_flux_divergence(P, U_i, Z_i, U_j, Z_j, c_ij, hof);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          F_iH[q] += weight * hof[q];
          P_ij[q] += weight * hof[q];
        }
      }
      {
        double S_j[K];
        E::manning_friction(P, U_j, h_star_j, tau, S_j);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          F_iH[q] += weight * m_ij * S_j[q];
          P_ij[q] += weight * m_ij * S_j[q];
        }
      }
      if constexpr (HAS_STAGES) {
        for (int s = 0; s < S.stages; ++s) {
          double U_iHs[K], U_jHs[K], hof_s[K], S_js[K];
          load_state<K>(S.U[s], i, U_iHs);
          load_state<K>(S.U[s], j, U_jHs);
          E::high_order_flux_divergence(P, U_iHs, Z_i, U_jHs, Z_j, c_ij, hof_s);
          E::manning_friction(P, U_jHs, S.prec[s][(size_t)j * 2 + 1], tau, S_js);
          const double w = S.w[s];
#pragma unroll
          for (int q = 0; q < K; ++q) {
            F_iH[q] += w * hof_s[q];
            P_ij[q] += w * hof_s[q];
          }
#pragma unroll
          for (int q = 0; q < K; ++q) {
            F_iH[q] += w * m_ij * S_js[q];
            P_ij[q] += w * m_ij * S_js[q];
          }
        }
      }
      store_entry<K>(pij, colbase, r.lane, P_ij);
    }

    if (!row_active)
      return;

    store_state<K>(new_U, i, U_i_new);
    store_state<K>(r_out, i, F_iH);

    /* Limiter::bounds (shallow_water/limiter.h:333-377) */
    const double hd_i = m_i * M.measure_of_omega_inverse;
    double r_i = sqrt(hd_i);
    if constexpr (DIM == 2) {
      const double t = sqrt(r_i);
      r_i = t * t * t;
    } else if constexpr (DIM == 1) {
      r_i = r_i * r_i * r_i;
    }
    r_i *= P.lim_relaxation_factor;
    const double h_relaxed = 2. * fabs(h_relaxation_numerator) / (relaxation_denominator + DBL_EPSILON);
    const double h_min_r = fmax((1. - r_i) * h_min, h_min - h_relaxed);
    const double h_max_r = fmin((1. + r_i) * h_max, h_max + h_relaxed);
    const double kin_relaxed =
        2. * fabs(kin_relaxation_numerator) / (relaxation_denominator + DBL_EPSILON);
    const double kin_max_r = fmin((1. + r_i) * kin_max, kin_max + kin_relaxed);
    const double v2_relaxed =
        2. * fabs(v2_relaxation_numerator) / (relaxation_denominator + DBL_EPSILON);
    const double v2_max_r = fmin((1. + r_i) * v2_max, v2_max + v2_relaxed);
    double r2 = hd_i;
    if constexpr (DIM == 2)
      r2 = sqrt(hd_i);
    r2 *= P.dry_state_relaxation_factor;
    const double h_small = P.reference_water_depth * r2;

    const size_t stride = M.bounds_stride;
    bounds[i] = h_min_r;
    bounds[stride + i] = h_max_r;
    bounds[2 * stride + i] = h_small;
    bounds[3 * stride + i] = kin_max_r;
    bounds[4 * stride + i] = v2_max_r;
  }
  /* The same sweep with ONE walk over the stencil (rows of at most MAXW columns). The reference walks the
   * stencil twice because the limiter bounds need U_ij_bar = 1/2 (U*_ij + U*_ji + (f*_ij - f*_ji) c_ij / d_ij)
   * + affine_shift_i, and the affine shift is itself a sum over the stencil (:700-720). Here the walk accumulates
   * the shift (same terms, same order: bitwise the same sum) and parks the shift-free part of U_ij_bar in LDS
   * -- [column][component][lane], 8 B per lane and access: conflict free -- ; a second, memory-free pass adds
   * the shift and forms the bounds. cols / c_ij / d_ij / Z_j are read once instead of twice (-252 B of 952 B per
   * row in 2-D) and the star states of the pre-loop are not computed twice. (Round 2 parked the STREAMS of the
   * first walk in LDS and kept both walks: slower. This removes the walk.) */
  template <int DIM, bool HAS_STAGES, int MAXW, bool FRICTION>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_LOW_SW)
  k_low_order_sw_single_walk(const ShallowWaterParams P, const DeviceMesh M,
                 DeviceScalars *scalars, const double weight,
                 const StageArgs<DIM> S, const double *__restrict__ U,
                 const double *__restrict__ prec, const double *__restrict__ Z,
                 const double *__restrict__ alpha, const double *__restrict__ dij,
                 double *__restrict__ new_U, double *__restrict__ r_out,
                 double *__restrict__ bounds, double *__restrict__ pij)
  {
    using E = ShallowWater<DIM>;
    constexpr int K = E::K;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double tau = finalize_tau(scalars);
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;
    const double *__restrict__ mij = M.mij;

    double U_i[K], U_i_new[K], F_iH[K], S_iH[K], S_i[K];
    load_state<K>(U, i, U_i);
    const double alpha_i = alpha[i];
    const double m_i = M.mi[i];
    const double m_i_inv = M.mi_inv[i];
    const double Z_i = Z[i];

    /* FRICTION = false (Manning coefficient 0, the reference's default and BASELINE configs[4]): every source
     * term of this sweep is (+-)0 times something finite -- S_i, S_iH, the m_ij weighted neighbour sources -- and
     * only ever added to something; the kernel then holds no registers for them, issues none of their
     * multiply-adds and does not read m_ij at all (8 B per stencil entry). Exact up to the sign of a zero. */
#pragma unroll
    for (int q = 0; q < K; ++q)
      S_iH[q] = S_i[q] = 0.;
    if constexpr (FRICTION) {
      if constexpr (HAS_STAGES) {
        for (int s = 0; s < S.stages; ++s) {
          double U_iHs[K], Ss[K];
          load_state<K>(S.U[s], i, U_iHs);
          E::manning_friction(P, U_iHs, S.prec[s][(size_t)i * 2 + 1], tau, Ss);
#pragma unroll
          for (int q = 0; q < K; ++q)
            S_iH[q] += S.w[s] * Ss[q];
        }
      }
      E::manning_friction(P, U_i, prec[(size_t)i * 2 + 1], tau, S_i);
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
      U_i_new[q] = U_i[q];
      F_iH[q] = 0.;
      if constexpr (FRICTION) {
        S_iH[q] += weight * S_i[q];
        U_i_new[q] += tau * S_i[q];
        F_iH[q] += m_i * S_iH[q];
      }
    }

    /* affine shift (:700-720, shallow_water/hyperbolic_system.h:1176-1191): accumulated by the walk below */
    /* (the diagonal column stays in registers: MAXW - 1 columns of K doubles per lane = 48 KiB per block in
     * 2-D, so that three blocks fit the 160 KiB of a CU) */
    __shared__ double bar_lds[kWavesPerBlock][MAXW - 1][K][64];
    double(*bar)[K][64] = bar_lds[threadIdx.x >> 6];
    double bar_diag[K];
#pragma unroll
    for (int q = 0; q < K; ++q)
      bar_diag[q] = 0.;
    double affine_shift[K];
#pragma unroll
    for (int q = 0; q < K; ++q)
      affine_shift[q] = 0.;
    const double h_inverse_i = E::inverse_water_depth_sharp(P, U_i);

    /* Limiter::reset (shallow_water/limiter.h:247-268) */
    double h_min = DBL_MAX, h_max = 0., kin_max = 0., v2_max = 0.;
    double h_relaxation_numerator = 0., kin_relaxation_numerator = 0., v2_relaxation_numerator = 0.,
           relaxation_denominator = 0.;
    const double kin_i = E::kinetic_energy(P, U_i);
    double v2_i;
    {
      const double ihm_i = E::inverse_water_depth_mollified(P, U_i);
      const double v = U_i[1] * ihm_i;
      v2_i = v * v;
#pragma unroll
      for (int d = 1; d < DIM; ++d) {
        const double vd = U_i[1 + d] * ihm_i;
        v2_i += vd * vd;
      }
    }

    uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
    uint32_t j_nn = r.width > 1 ? ld_stream(cols + (((uint64_t)r.base + 1) * 64 + r.lane)) : i;
    double c_n[DIM], U_n[K];
    load_entry<DIM>(cij, r.base, r.lane, c_n);
    double d_n = dij[(uint64_t)r.base * 64 + r.lane];
    double m_n = FRICTION ? ld_stream(mij + ((uint64_t)r.base * 64 + r.lane)) : 0.;
    load_state<K>(U, j_n, U_n);
    double alpha_n = alpha[j_n];
    double Z_n = Z[j_n];
    double h_star_n = FRICTION ? prec[(size_t)j_n * 2 + 1] : 1.; /* h_star_j only enters the friction term */
    /* the P_ij of the column before, stored behind this column's loads (kernels_euler.hpp, arrived()) */
    double P_pending[K];
    bool P_pending_on = false;
#pragma unroll
    for (int q = 0; q < K; ++q)
      P_pending[q] = 0.;
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
      double d_ij = d_n, m_ij = m_n, alpha_j = alpha_n, Z_j = Z_n, h_star_j = h_star_n;
      if constexpr (RYUJIN_PIN_WAITS) {
        arrived(c_ij);
        arrived(U_j);
        arrived(d_ij);
        arrived(alpha_j);
        arrived(Z_j);
        arrived(j_nn);
        if constexpr (FRICTION) {
          arrived(m_ij);
          arrived(h_star_j);
        }
      }
      if (c + 1 < r.width) {
        j_n = j_nn;
        load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
        d_n = dij[(colbase + 1) * 64 + r.lane];
        if constexpr (FRICTION)
          m_n = ld_stream(mij + ((colbase + 1) * 64 + r.lane));
        load_state<K>(U, j_n, U_n);
        alpha_n = alpha[j_n];
        Z_n = Z[j_n];
        if constexpr (FRICTION)
          h_star_n = prec[(size_t)j_n * 2 + 1];
        j_nn = (c + 2 < r.width) ? ld_stream(cols + ((colbase + 2) * 64 + r.lane)) : i;
      }
      if (P_pending_on)
        store_entry<K>(pij, colbase - 1, r.lane, P_pending);
      P_pending_on = false;
      if (!active)
        continue;

      const double factor = (alpha_i + alpha_j) * .5;
      const double d_ijH = d_ij * factor;
      const double denom = fmax(d_ij, 100. * DBL_MIN);
      double scaled_c_ij[DIM];
      const double inverse_denom = 1. / denom; /* dealii::Tensor / scalar multiplies by the inverse */
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        scaled_c_ij[d] = c_ij[d] * inverse_denom;

      double U_star_ij[K], U_star_ji[K];
      E::star_state(P, U_i, Z_i, Z_j, U_star_ij);
      E::star_state(P, U_j, Z_j, Z_i, U_star_ji);

      {
        double m_c = U_i[1] * c_ij[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          m_c += U_i[1 + d] * c_ij[d];
        const double shift_factor = 2. * (d_ij + h_inverse_i * m_c);
#pragma unroll
        for (int q = 0; q < K; ++q)
          affine_shift[q] += -shift_factor * (U_star_ij[q] - U_i[q]);
      }

      double flux_ij[K];
      E::flux_divergence(P, U_i, U_star_ij, U_star_ji, c_ij, flux_ij);

      double P_ij[K];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        U_i_new[q] += tau * m_i_inv * flux_ij[q];
        P_ij[q] = -flux_ij[q];
      }
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double dU = U_star_ji[q] - U_star_ij[q];
        U_i_new[q] += tau * m_i_inv * d_ij * dU;
        F_iH[q] += d_ijH * dU;
        P_ij[q] += (d_ijH - d_ij) * dU;
      }

      /* Limiter::accumulate (shallow_water/limiter.h:271-330) */
      {
        double f_star_ij[K][DIM], f_star_ji[K][DIM];
        E::f(P, U_star_ij, f_star_ij);
        E::f(P, U_star_ji, f_star_ji);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          double s = (f_star_ij[q][0] + (-f_star_ji[q][0])) * scaled_c_ij[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            s += (f_star_ij[q][d] + (-f_star_ji[q][d])) * scaled_c_ij[d];
          const double parked = 0.5 * (U_star_ij[q] + U_star_ji[q] + s); /* + affine_shift: second pass */
          if (c == 0)
            bar_diag[q] = parked;
          else
            bar[c - 1][q][r.lane] = parked;
        }
        relaxation_denominator += 1.;
        h_relaxation_numerator += 1. * (U_i[0] + U_j[0]);
        kin_relaxation_numerator += 1. * (kin_i + E::kinetic_energy(P, U_j));
        double v2_j;
        {
          const double ihm_j = E::inverse_water_depth_mollified(P, U_j);
          const double v = U_j[1] * ihm_j;
          v2_j = v * v;
#pragma unroll
          for (int d = 1; d < DIM; ++d) {
            const double vd = U_j[1 + d] * ihm_j;
            v2_j += vd * vd;
          }
        }
        v2_relaxation_numerator += 1. * (-v2_i + v2_j);
      }

      if constexpr (FRICTION) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
          F_iH[q] -= m_ij * S_iH[q];
          P_ij[q] -= m_ij * /*sic!*/ S_i[q];
        }
      }
      {
        double hof[K];
        E::high_order_flux_divergence(P, U_i, Z_i, U_j, Z_j, c_ij, hof);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          F_iH[q] += weight * hof[q];
          P_ij[q] += weight * hof[q];
        }
      }
      if constexpr (FRICTION) {
        double S_j[K];
        E::manning_friction(P, U_j, h_star_j, tau, S_j);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          F_iH[q] += weight * m_ij * S_j[q];
          P_ij[q] += weight * m_ij * S_j[q];
        }
      }
      if constexpr (HAS_STAGES) {
        for (int s = 0; s < S.stages; ++s) {
          double U_iHs[K], U_jHs[K], hof_s[K];
          load_state<K>(S.U[s], i, U_iHs);
          load_state<K>(S.U[s], j, U_jHs);
          E::high_order_flux_divergence(P, U_iHs, Z_i, U_jHs, Z_j, c_ij, hof_s);
          const double w = S.w[s];
#pragma unroll
          for (int q = 0; q < K; ++q) {
            F_iH[q] += w * hof_s[q];
            P_ij[q] += w * hof_s[q];
          }
          if constexpr (FRICTION) {
            double S_js[K];
            E::manning_friction(P, U_jHs, S.prec[s][(size_t)j * 2 + 1], tau, S_js);
#pragma unroll
            for (int q = 0; q < K; ++q) {
              F_iH[q] += w * m_ij * S_js[q];
              P_ij[q] += w * m_ij * S_js[q];
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < K; ++q)
        P_pending[q] = P_ij[q];
      P_pending_on = true;
    }
    if (P_pending_on)
      store_entry<K>(pij, (uint64_t)r.base + r.width - 1, r.lane, P_pending);

    if (!row_active)
      return;

    /* second pass: U_ij_bar = parked part + affine shift; Limiter::accumulate (shallow_water/limiter.h:271-330) */
#pragma unroll
    for (int q = 0; q < K; ++q) {
      affine_shift[q] *= tau * m_i_inv;
      if constexpr (FRICTION)
        affine_shift[q] += tau * S_i[q];
    }
    for (uint32_t c = 0; c < r.len; ++c) {
      double U_ij_bar[K];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_ij_bar[q] = (c == 0 ? bar_diag[q] : bar[c > 0 ? c - 1 : 0][q][r.lane]) + affine_shift[q];
      const double h_bar_ij = U_ij_bar[0];
      h_min = fmin(h_min, h_bar_ij);
      h_max = fmax(h_max, h_bar_ij);
      kin_max = fmax(kin_max, E::kinetic_energy(P, U_ij_bar));
      {
        const double ihm = E::inverse_water_depth_mollified(P, U_ij_bar);
        const double v = U_ij_bar[1] * ihm;
        double v2 = v * v;
#pragma unroll
        for (int d = 1; d < DIM; ++d) {
          const double vd = U_ij_bar[1 + d] * ihm;
          v2 += vd * vd;
        }
        v2_max = fmax(v2_max, v2);
      }
    }

    store_state<K>(new_U, i, U_i_new);
    store_state<K>(r_out, i, F_iH);

    /* Limiter::bounds (shallow_water/limiter.h:333-377) */
    const double hd_i = m_i * M.measure_of_omega_inverse;
    double r_i = sqrt(hd_i);
    if constexpr (DIM == 2) {
      const double t = sqrt(r_i);
      r_i = t * t * t;
    } else if constexpr (DIM == 1) {
      r_i = r_i * r_i * r_i;
    }
    r_i *= P.lim_relaxation_factor;
    const double h_relaxed = 2. * fabs(h_relaxation_numerator) / (relaxation_denominator + DBL_EPSILON);
    const double h_min_r = fmax((1. - r_i) * h_min, h_min - h_relaxed);
    const double h_max_r = fmin((1. + r_i) * h_max, h_max + h_relaxed);
    const double kin_relaxed =
        2. * fabs(kin_relaxation_numerator) / (relaxation_denominator + DBL_EPSILON);
    const double kin_max_r = fmin((1. + r_i) * kin_max, kin_max + kin_relaxed);
    const double v2_relaxed =
        2. * fabs(v2_relaxation_numerator) / (relaxation_denominator + DBL_EPSILON);
    const double v2_max_r = fmin((1. + r_i) * v2_max, v2_max + v2_relaxed);
    double r2 = hd_i;
    if constexpr (DIM == 2)
      r2 = sqrt(hd_i);
    r2 *= P.dry_state_relaxation_factor;
    const double h_small = P.reference_water_depth * r2;

    const size_t stride = M.bounds_stride;
    bounds[i] = h_min_r;
    bounds[stride + i] = h_max_r;
    bounds[2 * stride + i] = h_small;
    bounds[3 * stride + i] = kin_max_r;
    bounds[4 * stride + i] = v2_max_r;
  }
  /* Discontinuous ansatz: extend the limiter bounds over the stencil (hyperbolic_module.template.h:938-948) with
   * Limiter::combine_bounds AS WRITTEN in shallow_water/limiter.h:386-397:
   *   (min h_min, max h_max, min h_small, max(k_max_l, H_MAX_r) [sic], max v2_max)
   * -- the kinetic-energy bound of the running result is combined with the water-depth bound of the neighbour.
   * Reads the ORIGINAL bounds of the neighbours and writes a second buffer (as k_bounds_combine_euler). */
  __global__ void __launch_bounds__(kBlock)
  k_bounds_combine_sw(const DeviceMesh M, const double *__restrict__ in, double *__restrict__ out)
  {
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const size_t stride = M.bounds_stride;
    double b[5];
#pragma unroll
    for (int q = 0; q < 5; ++q)
      b[q] = in[(size_t)q * stride + i];
    for (uint32_t c = 1; c < r.width; ++c) {
      const uint32_t j = M.cols[((uint64_t)r.base + c) * 64 + r.lane];
      if (row_active && c < r.len) {
        const double h_max_r = in[stride + j];
        b[0] = fmin(b[0], in[j]);
        b[1] = fmax(b[1], h_max_r);
        b[2] = fmin(b[2], in[2 * stride + j]);
        b[3] = fmax(b[3], h_max_r); /* sic */
        b[4] = fmax(b[4], in[4 * stride + j]);
      }
    }
    if (row_active) {
#pragma unroll
      for (int q = 0; q < 5; ++q)
        out[(size_t)q * stride + i] = b[q];
    }
  }
} // namespace ryujin_hip

// Step 5 of HyperbolicModule::step (source/hyperbolic_module.template.h:892-1041) without stage vectors
// (stages == 0: every update of SSPRK22/33, the first stage of the ERK schemes), for the Descriptions whose
// P_ij has no source terms (Euler): P_ij IS FORMED HERE, ONCE, instead of being started in step 4 and
// finished in step 5.
//
// Without stage vectors the weight of the current flux is 1 - sum_s omega_s = 1, and the first part of P_ij,
//   P_ij = -flux_ij + (d_ij^H - d_ij)(U_j - U_i) + 1 * flux_ij                          (:795-846)
// is (d_ij^H - d_ij)(U_j - U_i) up to the rounding of the two flux terms that cancel. The reference stores that
// first part in step 4 (8 k S bytes per row) and reads, corrects and stores it again in step 5 -- in 3-D a
// third of all bytes an update moves. Here step 4 does not touch P_ij, and step 5 forms
//   P_ij = tau / m_i * (S - 1) * [ (d_ij^H - d_ij)(U_j - U_i) + b_ij F_j - b_ji F_i ],
//   b_ij = -m_ij / m_j,  b_ji = -m_ij / m_i                                              (:987-1001)
// in registers from what it streams (d_ij, m_ij) and gathers from the per-node vectors (U_j, F_j, alpha_j,
// 1/m_j: they stay in L2 / Infinity Cache between neighbouring rows), limits it and stores it for the two
// high-order passes. Neither c_ij nor a flux is evaluated (the 2-D kernel of rounds 1-2 recomputed the first part
// in the reference's operation order: c_ij stream + one flux evaluation per pair). Against the reference this
// P_ij differs by the cancellation residue of its flux terms, eps * |flux_ij|: ~1e-15 of the largest entry, three
// orders inside the 1e-12 contract on P_ij (tests/helpers_parity.py); its first part is antisymmetric bit for
// bit (d_ij = d_ji, alpha_i + alpha_j symmetric), which the reference's is only to round-off.
//
// The sweep also leaves, per row, V_i = U_i^low + sum_j lambda P_ij accumulated exactly as step 6 accumulates
// U_i^low + sum_j l_ij lambda P_ij when every l_ij is 1: in slices where nothing was limited step 6 takes V_i
// and never reads P_ij (kernels_limiter.hpp) -- bit-identical, and most of a developed flow.
//
// TILES: ONLY WHAT STEPS 6 AND 7 WILL READ IS STORED. With V_i they read P_ij only for limited pairs
// (U_i = V_i - sum_j (1 - l_ij) lambda P_ij). A developed Mach-3 flow holds a limited pair in 70 - 95 % of its 64-row
// slices but in a small share of its (slice, column) tiles; the 8 k S bytes per row of the full matrix -- a third of
// this sweep's traffic, half of step 6's -- are mostly never looked at. The sweep stores a tile (as the bracket
// Q_ij, qij_stage0) exactly when one of its own l_ij comes out below 1 (or undecided); a pair that is limited
// through the neighbour's l_ji alone is read from the neighbour's tile with the opposite sign (Q_ji = -Q_ij bit for
// bit). The rule is exact and local -- no flags, no prediction, no repair; kernels_limiter.hpp (TileSrc) has the
// readers' side. Round 3 chose between "store everything" and "store nothing, form P_ij again in steps 6/7" for the
// whole mesh from the fraction of limited slices; the first version of this round stored per slice with a
// prediction from the previous update (profiles/r04a_*, r04b_*: smooth in the limited fraction, but a developed
// flow has hardly a slice without a limited pair).

#pragma once

#include "kernels_limiter.hpp"

namespace ryujin_hip
{
#ifndef RYUJIN_OCC_LIJ0
#define RYUJIN_OCC_LIJ0 3 /* waves per SIMD asked of the register allocator */
#endif
#ifndef RYUJIN_OCC_LIJ0_3D
#define RYUJIN_OCC_LIJ0_3D 2 /* A/B on MI355X (4.2 M gridpoints): 2.15 ms at 3 waves (28 B/lane of scratch), 1.85 at 2 */
#endif

#ifndef RYUJIN_OCC_LIJ0_AEOS
#define RYUJIN_OCC_LIJ0_AEOS 2 /* EulerAEOS (a fourth bound, gamma_min, powers in psi): 52 B/lane of scratch at 3 waves */
#endif
  struct EulerAeosParams;
  template <typename E>
  constexpr int lij0_waves_per_simd()
  {
    if (std::is_same<typename E::Params, EulerAeosParams>::value)
      return RYUJIN_OCC_LIJ0_AEOS;
    return E::DIMENSION == 3 ? RYUJIN_OCC_LIJ0_3D : RYUJIN_OCC_LIJ0;
  }

  /* (PairData, RowData, load_pair, pij_stage0: kernels_limiter.hpp -- steps 6 and 7 use them as well) */

  /* NY > 1 (small meshes): NY waves (blockIdx.y) share a slice, wave y taking the columns 1 + y, 1 + y + NY, ...;
   * no V_i then (the row's sum is spread over several waves): the caller passes V_out = nullptr.
   * TILES: see the head of the file (NY == 1 only: it needs V_i); pij is the tile matrix Q_ij then. Otherwise all
   * of P_ij is stored. n_export_slices: the slices [0, n) store every tile (their ghost columns). */
  template <typename E, int NY = 1, bool TILES = false>
  __global__ void __launch_bounds__(kBlock, lij0_waves_per_simd<E>())
  k_lij_stage0(const typename E::Params P, const DeviceMesh M, DeviceScalars *__restrict__ scalars,
               const double *__restrict__ old_U, const double *__restrict__ alpha,
               const double *__restrict__ dij, const double *__restrict__ new_U,
               const double *__restrict__ r_in, const double *__restrict__ bounds, double *__restrict__ pij,
               double *__restrict__ lij, double *__restrict__ V_out, const uint32_t n_export_slices = 0)
  {
    static_assert(!TILES || NY == 1, "the readers of the tile storage need V_i");
    constexpr int K = E::K;
    constexpr int NB = E::NB;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const uint32_t *__restrict__ cols = M.cols;

    const size_t stride = M.bounds_stride;
    double bnd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      bnd[b] = bounds[(size_t)b * stride + i];
    double U_i_new[K], V_i[K];
    load_state<K>(new_U, i, U_i_new);
#pragma unroll
    for (int q = 0; q < K; ++q)
      V_i[q] = U_i_new[q];
    RowData<K> row;
    load_state<K>(old_U, i, row.U_i);
    load_state<K>(r_in, i, row.F_i);
    /* TILES: the bracket of the column at hand waits in LDS, not in registers, for the limiter's verdict: k doubles per
     * lane, conflict free ([component][lane]) */
    __shared__ double lds_stage[TILES ? kWavesPerBlock * K * 64 : 1];
    [[maybe_unused]] double *const q_mine = lds_stage + (TILES ? ((threadIdx.x >> 6) * K) * 64 + r.lane : 0);
    row.alpha_i = alpha[i];
    row.m_i_inv = M.mi_inv[i];
    row.factor = scalars->tau * row.m_i_inv * (double)(r.len - 1);
    const double lambda = 1. / (double)(r.len - 1);
    bool all_ok = true;
    unsigned long long undecided_mask = 0;
    const bool store_all = !TILES || r.slice < n_export_slices;

    const uint32_t c0 = 1 + (NY > 1 ? blockIdx.y : 0);
    /* software pipeline: the loads of the next column are in flight while column c is limited */
    uint32_t j_n = r.width > c0 ? ld_stream(cols + (((uint64_t)r.base + c0) * 64 + r.lane)) : i;
    uint32_t j_nn = r.width > c0 + NY ? ld_stream(cols + (((uint64_t)r.base + c0 + NY) * 64 + r.lane)) : i;
    PairData<K> next;
    if (r.width > c0)
      load_pair<K>(M, old_U, r_in, alpha, dij, ((uint64_t)r.base + c0) * 64 + r.lane, j_n, next);

    for (uint32_t c = c0; c < r.width; c += NY) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const bool active = row_active && c < r.len;
      double P_ij[K];
      qij_stage0<K>(row, next, P_ij);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        if constexpr (TILES)
          q_mine[q * 64] = P_ij[q]; /* the bracket waits in LDS (not in registers) for the limiter's verdict */
        P_ij[q] *= row.factor;
      }
      if (c + NY < r.width) {
        j_n = j_nn;
        load_pair<K>(M, old_U, r_in, alpha, dij, (colbase + NY) * 64 + r.lane, j_n, next);
        j_nn = (c + 2 * NY < r.width) ? ld_stream(cols + ((colbase + 2 * NY) * 64 + r.lane)) : i;
      }
      if constexpr (!TILES) {
        if (active)
          store_entry<K>(pij, colbase, r.lane, P_ij);
      }
      bool success = true, undecided = false;
      double l_ij = 1.;
      if (active) {
        if (NY == 1) {
          /* what the first high-order pass adds when nothing is limited: U += l lambda P with l = 1 (:1107-1131) */
#pragma unroll
          for (int q = 0; q < K; ++q)
            V_i[q] += lambda * P_ij[q];
        }
        l_ij = E::limit_fast(P, bnd, U_i_new, P_ij, success, undecided);
      }
      if constexpr (TILES) {
        /* wave-uniform: the tile is stored iff one of its own l_ij is (or may come out) below 1 -- the test a reader
         * repeats on the stored l_ij (an undecided pair that comes out at exactly 1 leaves a tile nobody reads) */
        if (store_all || __any(active && (undecided || !(l_ij == 1.)))) {
          if (active) {
            double Q_ij[K];
#pragma unroll
            for (int q = 0; q < K; ++q)
              Q_ij[q] = q_mine[q * 64];
            store_entry<K>(pij, colbase, r.lane, Q_ij);
          }
        }
      }
      if (!active)
        continue;
      if (undecided) {
        undecided_mask |= 1ull << c;
      } else {
        lij[pos] = l_ij;
        all_ok = all_ok && success;
      }
    }
    if (NY == 1 && V_out != nullptr && row_active)
      store_state<K>(V_out, i, V_i);

    /* the few pairs that need the Newton iteration */
    while (undecided_mask) {
      const uint32_t c = (uint32_t)__builtin_ctzll(undecided_mask);
      undecided_mask &= undecided_mask - 1;
      const uint64_t colbase = (uint64_t)r.base + c;
      double P_ij[K];
      load_entry<K>(pij, colbase, r.lane, P_ij); /* own tile: stored above (an undecided pair stores its tile) */
      if constexpr (TILES) {
#pragma unroll
        for (int q = 0; q < K; ++q)
          P_ij[q] *= row.factor;
      }
      bool success;
      const double l_ij = E::limit(P, bnd, U_i_new, P_ij, success);
      lij[colbase * 64 + r.lane] = l_ij;
      all_ok = all_ok && success;
    }
    flag_restart(scalars, all_ok, r.lane);
  }

  /* ryujin_hip_debug_fetch(P_ij) behind a step with tile storage: the full matrix P_ij as the sweeps see it --
   * the own tile where it is stored, the transpose's with the other sign for a pair limited through l_ji, and for the
   * entries no sweep reads (pairs that were not limited) pij_stage0() of the operands, which all outlive the step.
   * l_first: the l_ij of the first limiter pass. */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_pij_tiles_fetch(const DeviceMesh M, const Stage0Src S0, const double *__restrict__ q,
                    const double *__restrict__ l_first, const uint32_t n_export_slices, double *__restrict__ out)
  {
    constexpr int K = E::K;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    RowData<K> row;
    if (row_active)
      load_row_data<K>(M, S0, r.row, r.len, row);
    for (uint32_t c = 1; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint32_t pos = (uint32_t)(colbase * 64 + r.lane);
      const bool lane_on = row_active && c < r.len;
      const double l_a = lane_on ? l_first[pos] : 1.;
      const double l_b = lane_on ? l_first[M.idx_t[pos]] : 1.;
      const bool own = r.slice < n_export_slices || __any(lane_on && !(l_a == 1.));
      if (!lane_on)
        continue;
      double P_ij[K];
      if (own || !(l_b == 1.))
        load_tile<K>(M, q, colbase, r.lane, own, row.factor, P_ij);
      else
        pij_on_the_fly<K>(M, S0, row, colbase, r.lane, P_ij);
      store_entry<K>(out, colbase, r.lane, P_ij);
    }
  }
} // namespace ryujin_hip

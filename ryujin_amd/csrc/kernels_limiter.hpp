// Steps 5, 6, 7 of HyperbolicModule::step (source/hyperbolic_module.template.h:892-1182):
// mass-matrix correction of P_ij, convex limiter, high-order update.
//
// Wave-divergence strategy: Limiter::limit decides ~97 % of all (i,j) pairs with its first
// psi_r test; the remaining pairs need the quadratic Newton iteration (second pow, derivatives,
// two sqrt + three divisions per iteration). With one row per lane almost every 64-lane wave
// contains such a pair in a shock region, so each sweep runs limit_fast() over all columns
// (wave-uniform) and afterwards finishes the undecided (row, col) pairs -- collected in a per-lane
// bit mask -- with the full limit(); the trip count of that tail is the maximum number of undecided
// columns of any lane of the wave instead of "every column".

#pragma once

#include "kernels_euler.hpp"


namespace ryujin_hip
{
  /* Data-dependent shortcuts of steps 6 and 7 (exact: the skipped work contributes +0):
   *  - second limiter pass: a pair with l_ij = min(l_ij, l_ji) == 1 was not limited; the limiter would see
   *    (1 - l) P_ij = 0 and the stored value (1 - l) l'_ij is 0 whatever l'_ij is -- store 0, skip the
   *    limiter (a pow per pair) and, where P_ij is not register cached, its second read;
   *  - last high-order update: a pair with l == 0 adds l lambda P_ij = 0 to U_i -- skip the read of P_ij,
   *    8k of the 8k + 12 bytes a pair moves in this sweep.
   * Both tests are wave-uniform (__any over the 64 rows of the slice, per column), so nothing diverges.
   * Away from shocks the first pass returns l == 1 exactly (t_r stays t_max and psi_r > 0,
   * limiter.template.h:88-108,188-216) and the second pass therefore stores exact zeros: in a developed
   * Mach-3 flow the large majority of (slice, column) tiles take the shortcut.
   * Assumption: finite inputs. Where U_i or P_ij is NaN / inf (an update that has already left the admissible
   * set) the reference stores (1 - l) l' = 0 * NaN = NaN and adds 0 * inf = NaN, the shortcuts store / add 0: the
   * device result looks clean in that one entry where the reference propagates the NaN. Such a state raises the
   * restart flag in step 5 of the same update (the limiter reports failure on it), which is what the caller acts on. */
  /* l = min(l_ij, l_ji) (:1100-1106) that propagates a NaN of either operand (fmin would drop it: a NaN l_ij next to
   * l_ji = 1 must not pass for "not limited") */
  RYUJIN_DEV double lmin(const double a, const double b)
  {
    return (a == a && b == b) ? fmin(a, b) : a + b;
  }

  RYUJIN_DEV void flag_restart(DeviceScalars *scalars, const bool all_ok, const uint32_t lane)
  {
    if (__any(!all_ok)) {
      if (lane == 0)
        atomicOr(&scalars->restart_needed, 1);
    }
  }

  /* The undecided pairs of a wave, compacted. limit() -- two Newton iterations, a pow each -- is needed by a few per
   * cent of the pairs; left to the lane that owns the pair, a wave walks max_lane(#undecided) rounds with a handful
   * of lanes busy in each (the sweeps spend 10 - 20 % of their time there in a developed flow,
   * profiles/r04d_ab_tail_and_pow_cost_*). Instead the wave lists its undecided (lane, column) pairs in LDS -- a
   * ballot per column, no atomics -- and works the list off 64 at a time, every lane taking ANY pair: the row's
   * bounds and state wait in LDS, P_ij comes from the matrix. Same function on the same
   * operands, so the same l_ij; ceil(n / 64) rounds instead of max_lane(n_lane).
   *   queue      this wave's LDS list of (column << 6 | lane), capacity 63 * 64 entries: the mask covers the columns
   *              c0 + 1 ... c0 + 63 (rows wider than 64 entries come in blocks of 63 columns; column < 1024)
   *   load_P     (column, owning lane, out[K]): the P_ij the limiter is asked about
   *   emit       (column, owning lane, l): store the result
   * Returns false if some limit() reported failure. All 64 lanes must call (inactive rows with an empty mask). */
  template <typename E>
  struct TailScratch { /* per wave, in LDS */
    static constexpr int kRowDoubles = (E::NB + E::K) * 64;
    double rows[kRowDoubles];
  };

  template <typename E, typename LoadP, typename Emit>
  RYUJIN_DEV bool limit_undecided_pairs(const typename E::Params &P, const RowCtx &r, unsigned long long undecided_mask,
                                        const double (&bnd)[E::NB], const double (&U_i_new)[E::K],
                                        uint16_t *__restrict__ queue, double *__restrict__ rows, const LoadP &load_P,
                                        const Emit &emit, const uint32_t c0 = 0)
  {
    constexpr int K = E::K, NB = E::NB;
#ifndef RYUJIN_COMPACT_TAIL
#define RYUJIN_COMPACT_TAIL 1 /* 0 (A/B): every lane works its own pairs off, as rounds 1 - 3 did */
#endif
#if !RYUJIN_COMPACT_TAIL
    {
      bool ok = true;
      while (undecided_mask) {
        const uint32_t c = c0 + (uint32_t)__builtin_ctzll(undecided_mask);
        undecided_mask &= undecided_mask - 1;
        double P_ij[K];
        load_P(c, r.lane, P_ij);
        bool success;
        const double l_ij = E::limit(P, bnd, U_i_new, P_ij, success);
        emit(c, r.lane, l_ij);
        ok = ok && success;
      }
      return ok;
    }
#endif
    if (!__any(undecided_mask != 0ull))
      return true;
    uint32_t total = 0; /* wave-uniform */
    for (uint32_t k = 1; k < 64 && c0 + k < r.width; ++k) { /* bit k of the mask: column c0 + k */
      const bool mine = (undecided_mask >> k) & 1ull;
      const unsigned long long b = __ballot(mine);
      if (b == 0ull)
        continue;
      if (mine)
        queue[total + (uint32_t)__popcll(b & ((1ull << r.lane) - 1ull))] = (uint16_t)(((c0 + k) << 6) | r.lane);
      total += (uint32_t)__popcll(b);
    }
    /* the rows' bounds and states for whoever takes their pairs ([component][lane]: conflict free) */
#pragma unroll
    for (int q = 0; q < NB; ++q)
      rows[q * 64 + r.lane] = bnd[q];
#pragma unroll
    for (int q = 0; q < K; ++q)
      rows[(NB + q) * 64 + r.lane] = U_i_new[q];
    /* written and read by different lanes of the wave: the LDS rows above AND the P_ij this wave stored to global
     * memory earlier in the kernel (store_entry / streaming stores), which load_P() of ANOTHER lane reads back below --
     * the fence orders those global stores at workgroup scope as well; keep it if either side changes */
    __threadfence_block();
    bool all_ok = true;
    for (uint32_t q0 = 0; q0 < total; q0 += 64) {
      if (q0 + r.lane < total) {
        const uint32_t e = (uint32_t)queue[q0 + r.lane];
        const uint32_t c = e >> 6, owner = e & 63u;
        double b_o[NB], U_o[K];
#pragma unroll
        for (int q = 0; q < NB; ++q)
          b_o[q] = rows[q * 64 + owner];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_o[q] = rows[(NB + q) * 64 + owner];
        double P_ij[K];
        load_P(c, owner, P_ij);
        bool success;
        const double l_ij = E::limit(P, b_o, U_o, P_ij, success);
        emit(c, owner, l_ij);
        all_ok = all_ok && success;
      }
    }
    return all_ok;
  }

  /* ---- ryujin_hip_params::debug_expensive_bounds_check (Euler): the reference's EXPENSIVE_BOUNDS_CHECK build as
   * separate kernels between the sweeps; any violation raises the restart flag. */

  /* View::is_admissible() of every new state (hyperbolic_module.template.h:851-855 behind step 4, :1121-1126 behind
   * each high-order update) */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_check_admissible(const typename E::Params P, const DeviceMesh M, DeviceScalars *__restrict__ scalars,
                     const double *__restrict__ new_U)
  {
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    bool ok = true;
    if (r.len > 1) {
      double U[E::K];
      load_state<E::K>(new_U, r.row, U);
      ok = E::is_admissible(P, U);
    }
    flag_restart(scalars, ok, r.lane);
  }

  /* the `success` of the limiter's CHECKED control flow for every pair of one limiter pass: limit(bounds_i, U_i,
   * scale_ij P_ij) with U_i the state the pass limits (the low-order update for the first pass, the update after
   * the first pass for the second) and scale_ij = 1 (first pass) or 1 - min(l_ij, l_ji) (second pass: l_first !=
   * NULL). In the checked builds both passes count (:1031-1034, :1155-1161). */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_check_limiter(const typename E::Params P, const DeviceMesh M, DeviceScalars *__restrict__ scalars,
                  const double *__restrict__ U_limited, const double *__restrict__ bounds,
                  const double *__restrict__ pij, const double *__restrict__ l_first)
  {
    constexpr int K = E::K, NB = E::NB;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    bool ok = true;
    if (r.len > 1) {
      double bnd[NB], U[K];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        bnd[b] = bounds[(size_t)b * M.bounds_stride + r.row];
      load_state<K>(U_limited, r.row, U);
      for (uint32_t c = 1; c < r.len; ++c) {
        const uint64_t pos = ((uint64_t)r.base + c) * 64 + r.lane;
        double P_ij[K];
        load_entry<K>(pij, (uint64_t)r.base + c, r.lane, P_ij);
        if (l_first != nullptr) {
          const double l = lmin(l_first[pos], l_first[M.idx_t[pos]]);
#pragma unroll
          for (int q = 0; q < K; ++q)
            P_ij[q] = (1. - l) * P_ij[q];
        }
        bool success;
        E::limit_checked(P, bnd, U, P_ij, success);
        ok = ok && success;
      }
    }
    flag_restart(scalars, ok, r.lane);
  }

  /* ------------------------------------------------------------------ step 5 */

  /* DG: full inverse of the (block-diagonal) consistent mass matrix instead of the Neumann series
   * (hyperbolic_module.template.h:976-986): b_ij = m_i (M^-1)_ij, b_ji = m_j (M^-1)_ij */
  /* WIDE: rows of more than 64 entries (cG Q2 / Q3, dG in 3-D), the columns in blocks of 63 */
#ifndef RYUJIN_PIJ_LIJ_CHAINED
#define RYUJIN_PIJ_LIJ_CHAINED 1 /* step 5 with a stored first part of P_ij (shallow water, stage vectors, scalar, dG): chained gathers */
#endif
  template <typename E, bool DG = false, bool WIDE = false>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_PIJ)
  k_pij_lij(const typename E::Params P, const DeviceMesh M, DeviceScalars *__restrict__ scalars,
            const double *__restrict__ new_U, const double *__restrict__ r_in,
            const double *__restrict__ bounds, double *pij, double *__restrict__ lij,
            double *__restrict__ V_out = nullptr)
  {
    constexpr int K = E::K;
    constexpr int NB = E::NB;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double tau = scalars->tau;
    const uint32_t *__restrict__ cols = M.cols;
    /* DG: the matrix stream is (M^-1)_ij and the gathered per-node value is m_j instead of 1/m_j */
    const double *__restrict__ mij = DG ? M.mass_matrix_inverse : M.mij;
    const double *__restrict__ mi_inv = M.mi_inv;
    const double *__restrict__ node_j = DG ? M.mi : M.mi_inv;

    const size_t stride = M.bounds_stride;
    double bnd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      bnd[b] = bounds[(size_t)b * stride + i];
    const double m_i_inv = mi_inv[i];
    [[maybe_unused]] const double m_i = M.mi[i];
    double U_i_new[K], F_iH[K];
    load_state<K>(new_U, i, U_i_new);
    load_state<K>(r_in, i, F_iH);
    const double lambda_inv = (double)(r.len - 1);
    const double factor = tau * m_i_inv * lambda_inv;
    /* V_i = U_i^low + sum_j lambda P_ij, accumulated as the first high-order pass accumulates it when every l_ij
     * is 1: in slices where nothing was limited that pass takes V_i and skips P_ij (k_high_order_next_cached) */
    const double lambda = 1. / (double)(r.len - 1);
    double V_i[K];
#pragma unroll
    for (int q = 0; q < K; ++q)
      V_i[q] = U_i_new[q];
    bool all_ok = true;
    unsigned long long undecided_mask = 0;

    /* software pipeline: loads of column c+1 are in flight while column c is limited */
    uint32_t j_n = r.width > 1 ? tile_column<tile_map_pays<E::DIMENSION>()>(M, (uint64_t)r.base + 1, r.row, r.lane) : i;
    uint32_t j_nn = r.width > 2 ? tile_column<tile_map_pays<E::DIMENSION>()>(M, (uint64_t)r.base + 2, r.row, r.lane) : i;
    double P_n[K], F_n[K];
    double mjinv_n = 0., mij_n = 0.;
    if (r.width > 1) {
      load_entry<K>(pij, (uint64_t)r.base + 1, r.lane, P_n);
      load_state<K>(r_in, j_n, F_n);
      mjinv_n = node_j[j_n];
      mij_n = ld_stream(mij + (((uint64_t)r.base + 1) * 64 + r.lane));
    }

    /* (rows wider than 64 entries -- cG Q2 / Q3, dG in 3-D -- in blocks of 63 columns: the undecided pairs of a block
     * are finished before the next one starts, the mask has 64 bits) */
    /* the queue of undecided pairs is dynamic LDS, sized by the launch from the widest row of the mesh
     * (tail_queue_bytes(): 1 KB per wave for a 2-D Q1 stencil instead of the 8 KB the widest block needs -- with the
     * 4 KB of TailScratch that was 48 KB per block and capped the sweep at 3 blocks per CU whatever its registers) */
    extern __shared__ uint16_t tail_queue[];
    __shared__ TailScratch<E> tail_rows[kWavesPerBlock];
    const uint32_t queue_stride = M.tail_queue_columns * 64u;
    for (uint32_t c_blk = 1; c_blk < r.width; c_blk += 63) {
    const uint32_t c_end = (WIDE && c_blk + 63 < r.width) ? c_blk + 63 : r.width;
    for (uint32_t c = c_blk; c < c_end; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const bool active = row_active && c < r.len;
      double P_ij[K], F_jH[K];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        P_ij[q] = P_n[q];
        F_jH[q] = F_n[q];
      }
      const double m_j_inv = mjinv_n, m_ij = mij_n;
      if (c + 1 < r.width) {
        j_n = j_nn;
        load_entry<K>(pij, colbase + 1, r.lane, P_n);
        /* chained gathers (kernels_euler.hpp): r_j and the nodal mass of most columns are the previous column's, or
         * the slice's own rows', one lane over */
        constexpr bool kMasks = chain_masks_pay<E::DIMENSION>();
        const TileChain chain = RYUJIN_PIJ_LIJ_CHAINED != 0 ? tile_chain<kMasks>(M, colbase + 1) : TileChain{kChainNone, ~0ull};
        if (chain.kind == kChainNone) {
          load_state<K>(r_in, j_n, F_n);
          mjinv_n = node_j[j_n];
        } else {
          const double own_node = DG ? m_i : m_i_inv;
          if (chain.kind == kChainPrevColumn) {
#pragma unroll
            for (int q = 0; q < K; ++q)
              F_n[q] = lane_next(F_n[q]);
            mjinv_n = lane_next(mjinv_n);
          } else if (chain.kind == kChainOwnNext) {
#pragma unroll
            for (int q = 0; q < K; ++q)
              F_n[q] = lane_next(F_iH[q]);
            mjinv_n = lane_next(own_node);
          } else {
#pragma unroll
            for (int q = 0; q < K; ++q)
              F_n[q] = lane_prev(F_iH[q]);
            mjinv_n = lane_prev(own_node);
          }
          if (chain_lane_loads<kMasks>(chain, r.lane)) {
            load_state<K>(r_in, j_n, F_n);
            mjinv_n = node_j[j_n];
          }
        }
        mij_n = ld_stream(mij + ((colbase + 1) * 64 + r.lane));
        j_nn = (c + 2 < r.width) ? tile_column<tile_map_pays<E::DIMENSION>()>(M, colbase + 2, r.row, r.lane) : i;
      }
      if (!active)
        continue;

      /* Neumann series: b_ij = delta_ij - m_ij/m_j, b_ji = delta_ij - m_ij/m_i (:987-996) */
      double b_ij, b_ji;
      if constexpr (DG) {
        b_ij = m_i * m_ij - 0.;     /* m_ij holds (M^-1)_ij, m_j_inv holds m_j */
        b_ji = m_j_inv * m_ij - 0.;
      } else {
        b_ij = 0. - m_ij * m_j_inv;
        b_ji = 0. - m_ij * m_i_inv;
      }
#pragma unroll
      for (int q = 0; q < K; ++q) {
        P_ij[q] += b_ij * F_jH[q] - b_ji * F_iH[q];
        P_ij[q] *= factor;
      }
      store_entry<K>(pij, colbase, r.lane, P_ij);
#pragma unroll
      for (int q = 0; q < K; ++q)
        V_i[q] += lambda * P_ij[q];

      bool success, undecided;
      const double l_ij =
          E::limit_fast(P, bnd, U_i_new, P_ij, success, undecided);
      if (undecided) {
        undecided_mask |= 1ull << (c - c_blk + 1);
      } else {
        lij[pos] = l_ij;
        all_ok = all_ok && success;
      }
    }
    /* the few pairs of the block that need the Newton iteration */
    const bool tail_ok = limit_undecided_pairs<E>(
        P, r, undecided_mask, bnd, U_i_new, tail_queue + (threadIdx.x >> 6) * queue_stride,
        tail_rows[threadIdx.x >> 6].rows,
        [&](const uint32_t c, const uint32_t owner, double (&out)[K]) {
          load_entry<K>(pij, (uint64_t)r.base + c, owner, out);
        },
        [&](const uint32_t c, const uint32_t owner, const double l_ij) {
          lij[((uint64_t)r.base + c) * 64 + owner] = l_ij;
        },
        c_blk - 1);
    all_ok = all_ok && tail_ok;
    undecided_mask = 0;
    if constexpr (!WIDE)
      break; /* (one block: the compiler sees a straight line, the pipeline registers die before the tail) */
    }
    if (V_out != nullptr && row_active)
      store_state<K>(V_out, i, V_i);
    flag_restart(scalars, all_ok, r.lane);
  }

  /* (Round 6 tried the column pipeline of k_lij_stage0 here -- P_ij formed before the next column's loads are issued,
   * l_ij stored a column late: the shallow-water step 5 ran 4 - 9 % SLOWER on two boxes (0.400 -> 0.437 ms,
   * profiles/r06l_ab_pinned_waits_c5.log; its limiter is a closed form, there is little to hide behind) and the
   * kernel stays as it was.) */

  /* Step 5 for Euler, stages == 0, fused with the first part of P_ij of step 4 (:769-813): instead of
   * loading P_ij it is recomputed from U_i, U_j, alpha, d_ij, c_ij in exactly the operation order of
   * k_low_order, then the mass-matrix correction and the limiter follow as in k_pij_lij. */
  template <int DIM, int NY = 1>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_PIJ)
  k_pij_lij_recompute(const EulerParams P, const DeviceMesh M, DeviceScalars *__restrict__ scalars,
                      const double weight, const double *__restrict__ old_U,
                      const double *__restrict__ alpha, const double *__restrict__ dij,
                      const double *__restrict__ new_U, const double *__restrict__ r_in,
                      const double *__restrict__ bounds, double *__restrict__ pij,
                      double *__restrict__ lij)
  {
    using E = Euler<DIM>;
    constexpr int K = E::K;
    constexpr int NB = E::NB;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double tau = scalars->tau;
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;
    const double *__restrict__ mij = M.mij;
    const double *__restrict__ mi_inv = M.mi_inv;

    const size_t stride = M.bounds_stride;
    double bnd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      bnd[b] = bounds[(size_t)b * stride + i];
    const double m_i_inv = mi_inv[i];
    const double alpha_i = alpha[i];
    double U_i[K], U_i_new[K], F_iH[K];
    load_state<K>(old_U, i, U_i);
    load_state<K>(new_U, i, U_i_new);
    load_state<K>(r_in, i, F_iH);
    double f_i[K][DIM];
    E::flux(P, U_i, f_i);
    const double lambda_inv = (double)(r.len - 1);
    const double factor = tau * m_i_inv * lambda_inv;
    bool all_ok = true;
    unsigned long long undecided_mask = 0;

    /* NY > 1 (small meshes, less than two waves per SIMD: the sweep is one wave's latency chain): the pairs of
     * a row are independent in this sweep, so NY waves (blockIdx.y) share a slice, wave y taking the columns
     * 1 + y, 1 + y + NY, ... -- a chain NY times shorter, the per-row data read NY times from L2 */
    const uint32_t c0 = 1 + (NY > 1 ? blockIdx.y : 0);
    /* software pipeline: loads of the next column are in flight while column c is processed */
    uint32_t j_n = r.width > c0 ? tile_column<tile_map_pays<E::DIMENSION>()>(M, (uint64_t)r.base + c0, r.row, r.lane) : i;
    uint32_t j_nn = r.width > c0 + NY ? tile_column<tile_map_pays<E::DIMENSION>()>(M, (uint64_t)r.base + c0 + NY, r.row, r.lane) : i;
    double c_n[DIM], U_n[K], F_n[K];
    double mjinv_n = 0., mij_n = 0., d_n = 0., alpha_n = 0.;
    if (r.width > c0) {
      load_entry<DIM>(cij, (uint64_t)r.base + c0, r.lane, c_n);
      d_n = dij[((uint64_t)r.base + c0) * 64 + r.lane];
      mij_n = ld_stream(mij + (((uint64_t)r.base + c0) * 64 + r.lane));
      load_state<K>(old_U, j_n, U_n);
      load_state<K>(r_in, j_n, F_n);
      mjinv_n = mi_inv[j_n];
      alpha_n = alpha[j_n];
    }

    for (uint32_t c = c0; c < r.width; c += NY) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const bool active = row_active && c < r.len;
      double c_ij[DIM], U_j[K], F_jH[K];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        c_ij[d] = c_n[d];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        U_j[q] = U_n[q];
        F_jH[q] = F_n[q];
      }
      const double m_j_inv = mjinv_n, m_ij = mij_n, d_ij = d_n, alpha_j = alpha_n;
      if (c + NY < r.width) {
        j_n = j_nn;
        load_entry<DIM>(cij, colbase + NY, r.lane, c_n);
        d_n = dij[(colbase + NY) * 64 + r.lane];
        mij_n = ld_stream(mij + ((colbase + NY) * 64 + r.lane));
        load_state<K>(old_U, j_n, U_n);
        load_state<K>(r_in, j_n, F_n);
        mjinv_n = mi_inv[j_n];
        alpha_n = alpha[j_n];
        j_nn = (c + 2 * NY < r.width) ? tile_column<tile_map_pays<E::DIMENSION>()>(M, colbase + 2 * NY, r.row, r.lane) : i;
      }
      if (!active)
        continue;

      /* first part of P_ij: identical operation sequence to k_low_order (:769-813) */
      const double d_ijH = d_ij * ((alpha_i + alpha_j) * .5);
      double f_j[K][DIM], flux_ij[K], P_ij[K];
      E::flux(P, U_j, f_j);
      E::flux_divergence(f_i, f_j, c_ij, flux_ij);
#pragma unroll
      for (int q = 0; q < K; ++q)
        P_ij[q] = -flux_ij[q];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const double dU = U_j[q] - U_i[q];
        P_ij[q] += (d_ijH - d_ij) * dU;
      }
#pragma unroll
      for (int q = 0; q < K; ++q)
        P_ij[q] += weight * flux_ij[q];

      /* Neumann series: b_ij = delta_ij - m_ij/m_j, b_ji = delta_ij - m_ij/m_i (:987-996) */
      const double b_ij = 0. - m_ij * m_j_inv;
      const double b_ji = 0. - m_ij * m_i_inv;
#pragma unroll
      for (int q = 0; q < K; ++q) {
        P_ij[q] += b_ij * F_jH[q] - b_ji * F_iH[q];
        P_ij[q] *= factor;
      }
      store_entry<K>(pij, colbase, r.lane, P_ij);

      bool success, undecided;
      const double l_ij = E::limit_fast(P, bnd, U_i_new, P_ij, success, undecided);
      if (undecided) {
        undecided_mask |= 1ull << c;
      } else {
        lij[pos] = l_ij;
        all_ok = all_ok && success;
      }
    }

    while (undecided_mask) {
      const uint32_t c = (uint32_t)__builtin_ctzll(undecided_mask);
      undecided_mask &= undecided_mask - 1;
      const uint64_t colbase = (uint64_t)r.base + c;
      double P_ij[K];
      load_entry<K>(pij, colbase, r.lane, P_ij);
      bool success;
      const double l_ij = E::limit(P, bnd, U_i_new, P_ij, success);
      lij[colbase * 64 + r.lane] = l_ij;
      all_ok = all_ok && success;
    }
    flag_restart(scalars, all_ok, r.lane);
  }

  /* ------------------------------------------------------------------ steps 6, 7 */

  /* ---- P_ij without stage vectors, formed from d_ij, m_ij and per-node vectors (see kernels_limiter_stage0.hpp
   * for the derivation): used by step 5 and -- where P_ij is not stored at all -- by the limited slices of steps
   * 6 and 7 */

  /* what a row needs of a neighbour to form P_ij */
  template <int K>
  struct PairData {
    double U_j[K], F_j[K];
    double d_ij, m_ij, alpha_j, m_j_inv;
  };

  /* per-row constants */
  template <int K>
  struct RowData {
    double U_i[K], F_i[K];
    double alpha_i, m_i_inv, factor; /* factor = tau / m_i * (row_length - 1) */
  };

  template <int K>
  RYUJIN_DEV void load_pair(const DeviceMesh &M, const double *__restrict__ old_U,
                            const double *__restrict__ r_in, const double *__restrict__ alpha,
                            const double *__restrict__ dij, const uint64_t pos, const uint32_t j, PairData<K> &p)
  {
    p.d_ij = dij[pos];
    p.m_ij = ld_stream(M.mij + pos);
    load_state<K>(old_U, j, p.U_j);
    load_state<K>(r_in, j, p.F_j);
    p.alpha_j = alpha[j];
    p.m_j_inv = M.mi_inv[j];
  }

  /* The node data of the pair for a CHAINED tile (kernels_euler.hpp, chained gathers): `p` still holds the previous
   * column's (kChainPrevColumn), `row` the slice's own rows' -- or, where step 5 parks them there (PARK as
   * pij_stage0_parked: 1 F_i, 2 and U_i, 3 and the scalars), the wave's rows in LDS, [component][lane], where the
   * neighbouring row is the neighbouring word; d_ij and m_ij are the entry's own and stream as ever. */
  template <int K, int PARK = 0, bool MASKS = false>
  RYUJIN_DEV void load_pair_chained(const DeviceMesh &M, const double *__restrict__ old_U,
                                    const double *__restrict__ r_in, const double *__restrict__ alpha,
                                    const double *__restrict__ dij, const uint64_t pos, const uint32_t j,
                                    const TileChain &chain, const uint32_t lane, const RowData<K> &row,
                                    const double *parked, PairData<K> &p)
  {
    p.d_ij = dij[pos];
    p.m_ij = ld_stream(M.mij + pos);
    if (chain.kind == kChainPrevColumn) { /* wave-uniform */
#pragma unroll
      for (int q = 0; q < K; ++q) {
        p.U_j[q] = lane_next(p.U_j[q]);
        p.F_j[q] = lane_next(p.F_j[q]);
      }
      p.alpha_j = lane_next(p.alpha_j);
      p.m_j_inv = lane_next(p.m_j_inv);
    } else {
      const bool next = chain.kind == kChainOwnNext;
      /* (lane 63 / lane 0 read a word of the neighbouring component / their own: replaced below) */
      const double *src = parked + (next ? lane + 1u : (lane == 0u ? 0u : lane - 1u));
#pragma unroll
      for (int q = 0; q < K; ++q) {
        p.F_j[q] = PARK >= 1 ? src[q * 64] : (next ? lane_next(row.F_i[q]) : lane_prev(row.F_i[q]));
        p.U_j[q] = PARK >= 2 ? src[(K + q) * 64] : (next ? lane_next(row.U_i[q]) : lane_prev(row.U_i[q]));
      }
      p.alpha_j = PARK == 3 ? src[(2 * K + 0) * 64] : (next ? lane_next(row.alpha_i) : lane_prev(row.alpha_i));
      p.m_j_inv = PARK == 3 ? src[(2 * K + 1) * 64] : (next ? lane_next(row.m_i_inv) : lane_prev(row.m_i_inv));
    }
    if (chain_lane_loads<MASKS>(chain, lane)) {
      load_state<K>(old_U, j, p.U_j);
      load_state<K>(r_in, j, p.F_j);
      p.alpha_j = alpha[j];
      p.m_j_inv = M.mi_inv[j];
    }
  }

  /* P_ij for stages == 0 */
  template <int K>
  RYUJIN_DEV void pij_stage0(const RowData<K> &row, const PairData<K> &p, double (&P_ij)[K])
  {
    const double d_ijH = p.d_ij * ((row.alpha_i + p.alpha_j) * .5);
    const double dd = d_ijH - p.d_ij;
    /* Neumann series: b_ij = delta_ij - m_ij/m_j, b_ji = delta_ij - m_ij/m_i (:987-996) */
    const double b_ij = 0. - p.m_ij * p.m_j_inv;
    const double b_ji = 0. - p.m_ij * row.m_i_inv;
#pragma unroll
    for (int q = 0; q < K; ++q) {
      double v = dd * (p.U_j[q] - row.U_i[q]);
      v += b_ij * p.F_j[q] - b_ji * row.F_i[q];
      P_ij[q] = v * row.factor;
    }
  }

  /* The same with the row's F_i (and, PARK_U, U_i) parked in LDS, [component][lane] (conflict free): the same
   * operations on the same operands in the same order -- bit for bit pij_stage0() -- with 2 K (4 K) registers fewer
   * held across the column loop. Step 5 in 3-D: what separates the sweep from 3 waves per SIMD. */
  template <int K, bool PARK_U, bool PARK_SCALARS = false>
  RYUJIN_DEV void pij_stage0_parked(const RowData<K> &row, const double *parked_, const uint32_t lane,
                                    const PairData<K> &p, double (&P_ij)[K])
  {
    /* (the loads are loop invariant: an opaque lane offset keeps the compiler from hoisting them back into
     * registers) */
    uint32_t off = lane;
    asm volatile("" : "+v"(off));
    const double *parked = parked_ + off;
    /* PARK_SCALARS: the row's alpha_i, 1/m_i and factor wait in LDS as well (rows 2 K, 2 K + 1, 2 K + 2) */
    const double alpha_i = PARK_SCALARS ? parked[(2 * K + 0) * 64] : row.alpha_i;
    const double m_i_inv = PARK_SCALARS ? parked[(2 * K + 1) * 64] : row.m_i_inv;
    const double factor = PARK_SCALARS ? parked[(2 * K + 2) * 64] : row.factor;
    const double d_ijH = p.d_ij * ((alpha_i + p.alpha_j) * .5);
    const double dd = d_ijH - p.d_ij;
    const double b_ij = 0. - p.m_ij * p.m_j_inv;
    const double b_ji = 0. - p.m_ij * m_i_inv;
#pragma unroll
    for (int q = 0; q < K; ++q) {
      const double U_iq = PARK_U ? parked[(K + q) * 64] : row.U_i[q];
      double v = dd * (p.U_j[q] - U_iq);
      v += b_ij * p.F_j[q] - b_ji * parked[q * 64];
      P_ij[q] = v * factor;
    }
  }

  /* where the repair launch of step 6 (k_pij_repair) and the debug fetch take P_ij from for the columns step 5 did
   * not store: the operands of pij_stage0 */
#ifndef RYUJIN_TILE_PIJ_MAXDIM
#define RYUJIN_TILE_PIJ_MAXDIM 3 /* per-tile P_ij is BUILT up to this dimension; the host selects it by default up to RYUJIN_TILE_PIJ_DEFAULT_MAXDIM */
#endif
#ifndef RYUJIN_TILE_PIJ_DEFAULT_MAXDIM
#define RYUJIN_TILE_PIJ_DEFAULT_MAXDIM 2 /* 3-D: built, measured and NOT the default (debug_pij_storage = 3 selects it). C4 share, every slice limited,
                                            56 % of the tiles stored: step 5 2.18 against 2.23 - 2.32 ms (its L2-miss traffic falls by 4.5 %, not by
                                            the 24 % of its own bytes the tiles are: profiles/r06f_pmc_c4_*.md), step 6 1.68 against 1.45 ms
                                            -- 7.74 against 7.60 ms per update; C3, 38 % of the slices limited, 22 % of the tiles stored: 12.4 - 12.5
                                            against 12.2 ms per slice (profiles/r06b/d/e_ab_*) */
#endif
#ifndef RYUJIN_TILE_DEFER_MINDIM
#define RYUJIN_TILE_DEFER_MINDIM 3 /* from this dimension on the tiles step 5 did not store are formed OUTSIDE the sweep of step 6: the wave
                                      of a slice that misses a tile puts the slice on a list and retires, a small launch behind the sweep
                                      runs the listed slices with the repair (k_high_order_next_deferred). In 3-D the repair inside the
                                      kernel costs EVERY wave of the sweep: 124 -> 163 registers inlined (the fourth wave per SIMD gone), as
                                      a function call spills in the common path -- step 6 on the C4 share 1.74 - 1.76 ms either way against
                                      1.45 without the repair code (profiles/r06e_ab_repair_variants_c4.log). The launch behind the sweep
                                      takes 156 us for 85 slices (C4) and 316 us for 140 (C3) whether a slice gets a wave or a block of its
                                      own (profiles/r06c/r06d_kernel_trace_*.md): 1.68 ms in all -- the cheapest of the three, still a loss */
#endif
  struct Stage0Src {
    DeviceScalars *scalars; /* tau; and the limited-slice counters of step 6 */
    const double *old_U, *alpha, *dij, *r_in;
    /* != 0: step 5 stored P_ij PER TILE -- a (slice, column) tile iff one of its own l_ij came out limited (or went to
     * the Newton tail) --; step 6 forms and stores what it needs of the rest (next_cached_slice) */
    int tile_store;
  };

  template <int K>
  RYUJIN_DEV void load_row_data(const DeviceMesh &M, const Stage0Src &S0, const uint32_t i, const uint32_t len,
                                RowData<K> &row)
  {
    load_state<K>(S0.old_U, i, row.U_i);
    load_state<K>(S0.r_in, i, row.F_i);
    row.alpha_i = S0.alpha[i];
    row.m_i_inv = M.mi_inv[i];
    row.factor = S0.scalars->tau * row.m_i_inv * (double)(len - 1);
  }

  /* P_ij of column `colbase` of the row, exactly the value step 5 formed (same function, same operands) */
  template <int K>
  RYUJIN_DEV void pij_on_the_fly(const DeviceMesh &M, const Stage0Src &S0, const RowData<K> &row,
                                 const uint64_t colbase, const uint32_t lane, double (&P_ij)[K])
  {
    const uint64_t pos = colbase * 64 + lane;
    const uint32_t j = M.cols[pos];
    PairData<K> pd;
    load_pair<K>(M, S0.old_U, S0.r_in, S0.alpha, S0.dij, pos, j, pd);
    pij_stage0<K>(row, pd, P_ij);
  }

  /* TimeIntegrator::sadd (time_integrator.template.h:18-25) fused into the last sweep of a step:
   * new_U = s * new_U + b * src for the rows the sweep writes (src == nullptr: plain step). */
  struct FusedSadd {
    double s, b;
    const double *src;
  };

  /* The precomputation pass of the NEXT prepare_state_vector() (k_precompute_records: precomputed values and the
   * per-node Riemann record of the row, functions of the row's new state alone) fused into the last sweep of a
   * step, which holds that state in registers: saves the pre-pass its read of U and a launch per Runge-Kutta
   * stage. Armed by the device-resident RK driver, whose next call on the new vector IS prepare_state_vector();
   * boundary rows are redone behind the boundary conditions (k_apply_bc_records). prec == nullptr: plain step. */
  struct FusedPrecompute {
    double *prec, *rec;
  };

  template <typename E>
  RYUJIN_DEV void fused_precompute(const typename E::Params &P, const FusedPrecompute &FP, const uint32_t i,
                                   const double (&U_i)[E::K])
  {
    if constexpr (E::kFusablePrecompute) {
      constexpr int RS = E::RS;
      double r[RS];
      const double2 prec_i = E::precompute(P, U_i);
      reinterpret_cast<double2 *>(FP.prec)[i] = prec_i;
      E::node_record(P, U_i, prec_i, r);
      double2 *out = reinterpret_cast<double2 *>(FP.rec + (size_t)i * RS);
#pragma unroll
      for (int g = 0; g < RS / 2; ++g) {
        double2 t;
        t.x = r[2 * g];
        t.y = r[2 * g + 1];
        out[g] = t;
      }
    }
  }

  /* Generic variant: two passes over the row's stencil (the second one re-reads l_ij, l_ji, P_ij). */
  template <typename E, bool LAST_ROUND, bool WIDE = false>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_HO)
  k_high_order(const typename E::Params P, const DeviceMesh M, double *__restrict__ new_U,
               const double *__restrict__ bounds, const double *__restrict__ pij,
               const double *__restrict__ lij, double *__restrict__ lij_next, const FusedSadd F)
  {
    constexpr int K = E::K;
    constexpr int NB = E::NB;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const uint32_t *__restrict__ idx_t = M.idx_t;

    double U_i_new[K];
    load_state<K>(new_U, i, U_i_new);
    const double lambda = 1. / (double)(r.len - 1);

    for (uint32_t c = 1; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const bool active = row_active && c < r.len;
      const double l_a = lij[pos];
      const double l_b = lij[idx_t[pos]];
      const double l_ij = lmin(l_a, l_b);
      if (LAST_ROUND && !__any(active && !(l_ij == 0.)))
        continue;
      double p_ij[K];
      load_entry<K>(pij, colbase, r.lane, p_ij);
      if (!active)
        continue;
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_i_new[q] += l_ij * lambda * p_ij[q];
    }

    if constexpr (LAST_ROUND) {
      if (F.src) {
        double V[K];
        load_state<K>(F.src, i, V);
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_i_new[q] = F.s * U_i_new[q] + F.b * V[q];
      }
    }
    /* the fused sadd covers every owned entry, constrained rows (row length 1) included, as k_sadd and the
     * reference's sadd do (time_integrator.template.h:18-25) */
    if (row_active || (LAST_ROUND && F.src != nullptr && r.row < M.n_owned))
      store_state<K>(new_U, i, U_i_new);

    if constexpr (!LAST_ROUND) {
      const size_t stride = M.bounds_stride;
      double bnd[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b)
        bnd[b] = bounds[(size_t)b * stride + i];
      unsigned long long undecided_mask = 0;
      __shared__ uint16_t tail_queue[kWavesPerBlock * 63 * 64];
      __shared__ TailScratch<E> tail_rows[kWavesPerBlock];
      /* (blocks of 63 columns: rows wider than 64 entries, see k_pij_lij) */
      for (uint32_t c_blk = 1; c_blk < r.width; c_blk += 63) {
        const uint32_t c_end = (WIDE && c_blk + 63 < r.width) ? c_blk + 63 : r.width;
        for (uint32_t c = c_blk; c < c_end; ++c) {
          const uint64_t colbase = (uint64_t)r.base + c;
          const uint64_t pos = colbase * 64 + r.lane;
          const bool active = row_active && c < r.len;
          const double l_a = lij[pos];
          const double l_b = lij[tile_transposed<tile_map_pays<E::DIMENSION>()>(M, (uint64_t)r.base + c, r.lane)];
          const double old_l_ij = lmin(l_a, l_b);
          if (!__any(active && !(old_l_ij == 1.))) {
            if (active)
              st_stream(lij_next + (pos), 0.);
            continue;
          }
          double p_ij[K];
          load_entry<K>(pij, colbase, r.lane, p_ij);
          if (!active)
            continue;
          double new_p_ij[K];
#pragma unroll
          for (int q = 0; q < K; ++q)
            new_p_ij[q] = (1. - old_l_ij) * p_ij[q];
          bool success, undecided;
          const double new_l_ij =
              E::limit_fast(P, bnd, U_i_new, new_p_ij, success, undecided);
          if (undecided)
            undecided_mask |= 1ull << (c - c_blk + 1);
          else
            st_stream(lij_next + (pos), (1. - old_l_ij) * new_l_ij);
        }
        /* (the second pass's `success` is ignored unless EXPENSIVE_BOUNDS_CHECK, :1148-1161) */
        limit_undecided_pairs<E>(
            P, r, undecided_mask, bnd, U_i_new, tail_queue + (threadIdx.x >> 6) * 63 * 64,
            tail_rows[threadIdx.x >> 6].rows,
            [&](const uint32_t c, const uint32_t owner, double (&out)[K]) {
              const uint64_t pos = ((uint64_t)r.base + c) * 64 + owner;
              const double old_l_ij = lmin(lij[pos], lij[idx_t[pos]]);
              load_entry<K>(pij, (uint64_t)r.base + c, owner, out);
#pragma unroll
              for (int q = 0; q < K; ++q)
                out[q] = (1. - old_l_ij) * out[q];
            },
            [&](const uint32_t c, const uint32_t owner, const double new_l_ij) {
              const uint64_t pos = ((uint64_t)r.base + c) * 64 + owner;
              const double old_l_ij = lmin(lij[pos], lij[idx_t[pos]]);
              st_stream(lij_next + (pos), (1. - old_l_ij) * new_l_ij);
            },
            c_blk - 1);
        undecided_mask = 0;
        if constexpr (!WIDE)
          break;
      }
    }
  }

  /* the tiles `missing` (bit c: column c) of the slice formed exactly as step 5 forms them and stored (next_cached_slice):
   * load_row_data() + pij_on_the_fly() spelled out on plain pointers */
  template <int K>
  RYUJIN_DEV void repair_missing_tiles(const uint32_t *__restrict__ cols, const double *__restrict__ mij,
                                       const double *__restrict__ mi_inv, const double *__restrict__ old_U,
                                       const double *__restrict__ r_in, const double *__restrict__ alpha,
                                       const double *__restrict__ dij, const double tau, const uint32_t i,
                                       const uint32_t len, const uint32_t base, const uint32_t width,
                                       const uint32_t lane, const bool row_active, const uint32_t missing,
                                       double *__restrict__ pij)
  {
    RowData<K> row;
    load_state<K>(old_U, i, row.U_i);
    load_state<K>(r_in, i, row.F_i);
    row.alpha_i = alpha[i];
    row.m_i_inv = mi_inv[i];
    row.factor = tau * row.m_i_inv * (double)(len - 1);
#pragma unroll 1
    for (uint32_t c = 1; c < width; ++c) {
      if (!((missing >> c) & 1u))
        continue;
      const uint64_t pos = ((uint64_t)base + c) * 64 + lane;
      const uint32_t j = cols[pos];
      PairData<K> pd;
      pd.d_ij = dij[pos];
      pd.m_ij = ld_stream(mij + pos);
      load_state<K>(old_U, j, pd.U_j);
      load_state<K>(r_in, j, pd.F_j);
      pd.alpha_j = alpha[j];
      pd.m_j_inv = mi_inv[j];
      double P_t[K];
      pij_stage0<K>(row, pd, P_t);
      if (row_active && c < len)
        store_entry<K>(pij, (uint64_t)base + c, lane, P_t);
    }
  }
  /* ... as a FUNCTION CALL (3-D): the registers of the repair are the callee's, saved and restored around the rare
   * call, instead of being added to what every wave of the sweep holds (inlined: 124 -> 163 registers, the fourth wave
   * per SIMD gone). Plain pointer arguments: a reference to the kernel's DeviceMesh would make every wave write a copy
   * of it to its stack at the top of the kernel. */
  template <int K>
  __device__ __attribute__((noinline)) void
  repair_missing_tiles_out_of_line(const uint32_t *cols, const double *mij, const double *mi_inv, const double *old_U,
                                   const double *r_in, const double *alpha, const double *dij, const double tau,
                                   const uint32_t i, const uint32_t len, const uint32_t base, const uint32_t width,
                                   const uint32_t lane, const bool row_active, const uint32_t missing, double *pij)
  {
    repair_missing_tiles<K>(cols, mij, mi_inv, old_U, r_in, alpha, dij, tau, i, len, base, width, lane, row_active,
                            missing, pij);
  }

  /* Per-slice bookkeeping of the limiter sweeps of an update without stage vectors (kernels_limiter_stage0.hpp):
   * one byte per 64-row slice each, written by exactly one wave per launch.
   *   unlimited     written by step 6: 1 = no pair of the slice was limited in the first high-order pass. EXACT for
   *                 the last sweep of the same update (every l'_ij of such a slice is an exact zero, and so is every
   *                 transposed l'_ji: min(l_ij, l_ji) = 1 is symmetric, (1 - 1) l' = 0 -- it reads neither), and
   *                 the PREDICTION step 5 of the next update stores P_ij by (a limited region moves by less than a
   *                 cell per update);
   *   first_stored  written by step 5 where it stores P_ij per slice: 0 = nothing of the slice is in the matrix, k =
   *                 the columns k, k + 1, ... are (1: all of them -- the slice was predicted limited; k > 1: its
   *                 own l_ij of column k came out limited). The repair launch of step 6 completes the slices that
   *                 need it and sets 1. Exact;
   *   todo          written by the light launch of step 6 for the two behind it: 0 finished (V_i), 1 P_ij complete,
   *                 2 the repair launch has to complete it first. */
  struct SliceFlags {
    uint8_t *unlimited, *first_stored, *todo;
    /* the tiles step 6 read P_ij of in the last updates (an SSPRK33 step is three updates, and what its stages limit
     * differs). Stencils of up to 10 columns (1-D, 2-D Q1): [n_slices] words, bit c: the (slice, column c) tile in the
     * last update; bits 10 + c, 20 + c: in the update before, and the one before that. Wider stencils (3-D Q1: 27
     * columns): one word per generation, [generation][hist_stride] (tiles_predicted / tiles_remember below). */
    uint32_t *needed_tiles;
    uint32_t hist_stride; /* >= n_slices */
    /* where the missing tiles are formed outside the sweep of step 6 (RYUJIN_TILE_DEFER_MINDIM): the slices whose wave
     * retired, [n_slices]; the launch over the slices [slice_begin, slice_end) appends from entry slice_begin on,
     * counted in DeviceScalars::n_deferred[slice_begin != 0] (the export and the interior part of a split sweep) */
    uint32_t *deferred;
  };

#ifndef RYUJIN_GATHER_GROUP_3D
#define RYUJIN_GATHER_GROUP_3D 6 /* columns whose l_ij / l_ji are fetched in one batch where the row has more than 9 (steps 6, 7 in 3-D). C4 share, same process (profiles/r06v_ab_batched_gathers_c4.log): step 6 1.497 -> 1.379 (9) / 1.327 (6) / 1.413 (13) ms, step 7 0.857 -> 0.789 / 0.761 / 0.801 */
#endif
#ifndef RYUJIN_TILE_PIJ_GENERATIONS
#define RYUJIN_TILE_PIJ_GENERATIONS 3
#endif
#ifndef RYUJIN_TILE_REPAIR_CALL
#define RYUJIN_TILE_REPAIR_CALL 1 /* 3-D: the repair of step 6 as a function call (repair_missing_tiles_out_of_line) */
#endif
  /* the tiles step 5 stores on the strength of the last updates; MAXW: the widest row of the stencil family */
  template <int MAXW>
  RYUJIN_DEV uint32_t tiles_predicted(const SliceFlags &W, const uint32_t slice)
  {
    if constexpr (MAXW <= 10) {
      const uint32_t word = W.needed_tiles[slice];
      uint32_t m = word & 0x3ffu;
      if (RYUJIN_TILE_PIJ_GENERATIONS >= 2)
        m |= (word >> 10) & 0x3ffu;
      if (RYUJIN_TILE_PIJ_GENERATIONS >= 3)
        m |= (word >> 20) & 0x3ffu;
      return m;
    } else {
      static_assert(MAXW <= 32, "one bit per column");
      uint32_t m = W.needed_tiles[slice];
      if (RYUJIN_TILE_PIJ_GENERATIONS >= 2)
        m |= W.needed_tiles[(size_t)W.hist_stride + slice];
      if (RYUJIN_TILE_PIJ_GENERATIONS >= 3)
        m |= W.needed_tiles[2 * (size_t)W.hist_stride + slice];
      return m;
    }
  }
  /* step 6 (one lane): `needed` becomes the newest generation, the oldest is forgotten */
  template <int MAXW>
  RYUJIN_DEV void tiles_remember(const SliceFlags &W, const uint32_t slice, const uint32_t needed)
  {
    if constexpr (MAXW <= 10) {
      W.needed_tiles[slice] = ((W.needed_tiles[slice] << 10) | (needed & 0x3ffu)) & 0x3fffffffu;
    } else {
      if (RYUJIN_TILE_PIJ_GENERATIONS >= 3)
        W.needed_tiles[2 * (size_t)W.hist_stride + slice] = W.needed_tiles[(size_t)W.hist_stride + slice];
      if (RYUJIN_TILE_PIJ_GENERATIONS >= 2)
        W.needed_tiles[(size_t)W.hist_stride + slice] = W.needed_tiles[slice];
      W.needed_tiles[slice] = needed;
    }
  }
  /* words of history per slice the host allocates (and sets to all ones: the first update stores every tile) */
  constexpr int tile_history_words(const int maxw) { return maxw <= 10 ? 1 : 3; }

  /* form and store the P_ij of the columns [1, c_end) of the row (the repair launch of step 6,
   * ryujin_hip_debug_fetch): exactly the value step 5 formed (same function, same operands) */
  template <int K>
  RYUJIN_DEV void backfill_pij(const DeviceMesh &M, const Stage0Src &S0, const RowCtx &r,
                               double *__restrict__ pij, const uint32_t c_end)
  {
    if (r.len <= 1)
      return;
    RowData<K> row;
    load_row_data<K>(M, S0, r.row, r.len, row);
    for (uint32_t c = 1; c < c_end && c < r.len; ++c) {
      double P_ij[K];
      pij_on_the_fly<K>(M, S0, row, (uint64_t)r.base + c, r.lane, P_ij);
      store_entry<K>(pij, (uint64_t)r.base + c, r.lane, P_ij);
    }
  }

  /* Last round for stencils of at most MAXW columns: all l_ij = min(l_ij, l_ji) of the row are fetched up
   * front (independent loads), then P_ij is read -- in chunks of CHUNK columns whose loads are issued back to
   * back -- only for the columns in which some row of the slice has l != 0 (see the note at the top).
   * slice_unlimited (SliceFlags::unlimited of this update's step 6, or NULL): slices in which nothing was limited
   * fetch nothing at all. */
  template <typename E, int MAXW, int CHUNK>
  RYUJIN_DEV void last_cached_slice(const typename E::Params &P, const DeviceMesh &M, const RowCtx &r,
                                    double *__restrict__ new_U, const double *__restrict__ pij,
                                    const double *__restrict__ lij, const FusedSadd &F, const FusedPrecompute &FP,
                                    const uint8_t *__restrict__ slice_unlimited = nullptr)
  {
    constexpr int K = E::K;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const uint32_t *__restrict__ idx_t = M.idx_t;

    double U_i_new[K];
    load_state<K>(new_U, i, U_i_new);
    double V[K];
    if (F.src)
      load_state<K>(F.src, i, V);
    const double lambda = 1. / (double)(r.len - 1);

    double l[MAXW];
    uint32_t needed = 0; /* wave-uniform: bit c <=> some row of the slice has l(c) != 0 */
    const bool known_unlimited = slice_unlimited != nullptr && slice_unlimited[r.slice] != 0;
#pragma unroll
    for (int c = 1; c < MAXW; ++c)
      l[c] = 0.;
    if (!known_unlimited) {
      /* in groups of up to 9 columns: all transposed positions, all l_ij and l_ji, then the minima
       * (transposed_positions(), kernels_euler.hpp) */
      constexpr int CH = MAXW - 1 < 9 ? MAXW - 1 : RYUJIN_GATHER_GROUP_3D;
#pragma unroll
      for (int c0 = 1; c0 < MAXW; c0 += CH) {
        uint32_t tpos[CH];
        transposed_positions<tile_map_pays<E::DIMENSION>(), CH>(M, r, c0, tpos);
        batch_fence();
        double l_a[CH], l_b[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int c = c0 + k;
          l_a[k] = l_b[k] = 0.;
          if (c < MAXW && (uint32_t)c < r.width) {
            l_a[k] = lij[(uint32_t)(((uint64_t)r.base + c) * 64 + r.lane)];
            l_b[k] = lij[tpos[k]];
          }
        }
        batch_fence();
#pragma unroll
        for (int k = 0; k < CH; ++k) { /* ONE wait for the group's values (kernels_euler.hpp, k_dij_diag_unrolled) */
          asm volatile("" : "+v"(l_a[k]));
          asm volatile("" : "+v"(l_b[k]));
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int c = c0 + k;
          if (c < MAXW && (uint32_t)c < r.width)
            l[c] = (row_active && (uint32_t)c < r.len) ? lmin(l_a[k], l_b[k]) : 0.;
        }
      }
#pragma unroll
      for (int c = 1; c < MAXW; ++c)
        if (__any(l[c] != 0.))
          needed |= 1u << c;
    }

#pragma unroll
    for (int c0 = 1; c0 < MAXW; c0 += CHUNK) {
      double p[CHUNK][K];
#pragma unroll
      for (int cc = 0; cc < CHUNK; ++cc) {
        const int c = c0 + cc;
        if (c < MAXW && ((needed >> c) & 1u))
          load_entry<K>(pij, (uint64_t)r.base + c, r.lane, p[cc]);
      }
#pragma unroll
      for (int cc = 0; cc < CHUNK; ++cc) {
        const int c = c0 + cc;
        if (c < MAXW && ((needed >> c) & 1u)) {
          if (row_active && (uint32_t)c < r.len) { /* padding entries of P_ij are never read into U */
#pragma unroll
            for (int q = 0; q < K; ++q)
              U_i_new[q] += l[c] * lambda * p[cc][q];
          }
        }
      }
    }

    if (F.src) {
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_i_new[q] = F.s * U_i_new[q] + F.b * V[q];
    }
    if (row_active || (F.src != nullptr && r.row < M.n_owned))
      store_state<K>(new_U, i, U_i_new);
    if (FP.prec != nullptr && row_active) /* rows of length 1 are skipped by the pre-pass as well */
      fused_precompute<E>(P, FP, i, U_i_new);
  }

  RYUJIN_DEV RowCtx row_context_of_slice(const DeviceMesh &M, const uint32_t slice)
  {
    RowCtx r;
    r.lane = threadIdx.x & 63;
    r.slice = slice;
    r.valid = true;
    r.row = slice * 64 + r.lane;
    r.len = M.row_len[r.row];
    r.base = M.slice_off[slice];
    r.width = M.slice_off[slice + 1] - r.base;
    return r;
  }

  template <typename E, int MAXW, int CHUNK>
  __global__ void __launch_bounds__(kBlock, (MAXW > 9 ? RYUJIN_OCC_LAST_3D : 1))
  k_high_order_last_cached(const typename E::Params P, const DeviceMesh M, double *__restrict__ new_U,
                           const double *__restrict__ pij, const double *__restrict__ lij, const FusedSadd F,
                           const FusedPrecompute FP, const uint8_t *__restrict__ slice_unlimited = nullptr)
  {
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    last_cached_slice<E, MAXW, CHUNK>(P, M, r, new_U, pij, lij, F, FP, slice_unlimited);
  }

  /* Register-cached variant for stencils of at most MAXW columns (2-D Q1: 9, 1-D: 3): the row's
   * l_ij = min(l_ij, l_ji) and P_ij stay in registers between the update and the next limiter pass,
   * so step 6 reads every array exactly once (the generic variant fetches ~2x the algorithmic bytes)
   * and all loads of a row are independent and issued up front.
   * CP < MAXW (3-D Q1: 27 columns of 5 components do not fit the register file at a useful occupancy):
   * all l_ij but only the P_ij of columns < CP are cached, the others are fetched a second time. */
  /* SPLIT (small meshes, the sweep is one wave's latency chain): the four waves of a block share ONE slice; all
   * of them form the new U_i (bitwise the same sum), the first one stores it -- behind a block barrier, the update
   * is in place -- and wave w runs the second limiter pass for the columns 1 + w, 5 + w, ... only. */
  /* MODE (updates whose step 5 stored P_ij per slice, kernels_limiter_stage0.hpp; V_unlimited and the flags are
   * required then): the sweep runs as three launches over all slices,
   *   kHoLight   slices of which step 5 stored nothing: fetch l_ij / l_ji; if nothing is limited take V_i and finish
   *              -- a kernel of 20 registers at full occupancy --, otherwise leave the slice to the launches behind
   *              (todo = 2: limited through a neighbour's l_ji alone). Slices with a complete P_ij are passed on
   *              unseen (todo = 1), slices stored from some column on are limited for sure (todo = 2);
   *   k_pij_repair  forms and stores what is missing of the P_ij of the slices with todo = 2;
   *   kHoHeavy   waves of finished slices retire at once; the others run the sweep on the stored P_ij.
   * kHoPlain: the whole sweep in one launch (P_ij stored everywhere).
   * With V_i the new state is formed as V_i - sum_j (1 - l_ij) lambda P_ij over the (slice, column) tiles in which
   * some pair is limited -- the terms of all other tiles are exact zeros, and P_ij is read for those tiles only,
   * once, for the sum and for the second limiter pass. Against U_i^low + sum_j l_ij lambda P_ij in column order
   * (the reference, :1107-1131, and the variant without V_i below) this is another rounding of the same sum:
   * differences of a few ulp of lambda |P_ij|, orders inside the 1e-11 contract on the new state. Rows whose
   * columns are ALL in limited tiles (the strongly limited ones) take the reference's form in the reference's
   * order instead: see the branch below. */
  /* P_ij stored per tile and the missing tiles formed outside the sweep (RYUJIN_TILE_DEFER_MINDIM):
   *   kHoDefer   the whole sweep in one launch, but the wave of a slice that needs a tile step 5 did not store appends
   *              the slice to SliceFlags::deferred and retires before it has written anything;
   *   kHoRepair  the launch behind it (k_high_order_next_deferred): forms and stores the missing tiles of the listed
   *              slices -- as kHoPlain does inside the sweep up to two dimensions -- and runs the sweep on them. */
  constexpr int kHoPlain = 0, kHoLight = 1, kHoHeavy = 2, kHoDefer = 3, kHoRepair = 4;
  template <int DIM, int MODE>
  constexpr bool forms_missing_tiles()
  {
    return DIM <= RYUJIN_TILE_PIJ_MAXDIM && ((MODE == kHoPlain && DIM < RYUJIN_TILE_DEFER_MINDIM) || MODE == kHoRepair);
  }

  template <typename E, int MAXW, int CP, bool SPLIT, int MODE>
  RYUJIN_DEV void next_cached_slice(const typename E::Params &P, const DeviceMesh &M, const RowCtx &r,
                                    const uint32_t group, double *__restrict__ new_U,
                                    const double *__restrict__ bounds, double *__restrict__ pij,
                                    const double *__restrict__ lij, double *__restrict__ lij_next,
                                    const double *__restrict__ V_unlimited, const Stage0Src &S0, const SliceFlags &W,
                                    const uint32_t todo = 1)
  {
    constexpr int K = E::K;
    constexpr int NB = E::NB;
    static_assert(!SPLIT || CP == MAXW, "the split variant caches the whole row");
    static_assert(!SPLIT || MODE == kHoPlain || MODE == kHoRepair, "small meshes keep the stored P_ij");
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const uint32_t *__restrict__ idx_t = M.idx_t;

    double l[MAXW];
    if constexpr (MODE == kHoLight) {
      bool limited = false;
      constexpr int CH = MAXW - 1 < 9 ? MAXW - 1 : RYUJIN_GATHER_GROUP_3D;
#pragma unroll
      for (int c0 = 1; c0 < MAXW; c0 += CH) {
        uint32_t tpos[CH];
        transposed_positions<tile_map_pays<E::DIMENSION>(), CH>(M, r, c0, tpos);
        batch_fence();
        double l_a[CH], l_b[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int c = c0 + k;
          l_a[k] = l_b[k] = 1.;
          if (c < MAXW && (uint32_t)c < r.width) {
            l_a[k] = lij[(uint32_t)(((uint64_t)r.base + c) * 64 + r.lane)];
            l_b[k] = lij[tpos[k]];
          }
        }
        batch_fence();
#pragma unroll
        for (int k = 0; k < CH; ++k) { /* ONE wait for the group's values (kernels_euler.hpp, k_dij_diag_unrolled) */
          asm volatile("" : "+v"(l_a[k]));
          asm volatile("" : "+v"(l_b[k]));
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int c = c0 + k;
          /* (NaN counts as limited: !(l == 1), not l != 1 through fmin, which drops a NaN operand) */
          if (c < MAXW && (uint32_t)c < r.width)
            limited = limited || (row_active && (uint32_t)c < r.len && !(l_a[k] == 1. && l_b[k] == 1.));
        }
      }
      const bool slice_limited = __any(limited);
      if ((r.slice & 15u) == 0 && r.lane == 0) {
        atomicAdd(&S0.scalars->n_sampled_slices, 1u);
        if (slice_limited)
          atomicAdd(&S0.scalars->n_sampled_limited, 1u);
      }
      if (r.lane == 0) {
        W.todo[r.slice] = slice_limited ? 3 : 0; /* 3: limited, nothing stored, and counted above */
        if (!slice_limited)
          W.unlimited[r.slice] = 1;
      }
      if (!slice_limited && row_active) {
        double V_i[K];
        load_state<K>(V_unlimited, i, V_i);
        store_state<K>(new_U, i, V_i);
#pragma unroll
        for (int c = 1; c < MAXW; ++c)
          if ((uint32_t)c < r.len)
            st_stream(lij_next + ((r.base + c) * 64 + r.lane), 0.);
      }
      return;
    }

    double U_i_new[K];
    const double lambda = 1. / (double)(r.len - 1);
    const size_t stride = M.bounds_stride;
    double bnd[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
      bnd[b] = bounds[(size_t)b * stride + i];

    double p[CP][K];
    uint32_t needed = 0; /* wave-uniform: bit c <=> some pair of the (slice, column) tile is limited */
    uint32_t own_limited = 0; /* ... <=> one of the tile's OWN l_ij is (the tiles step 5 stored with Stage0Src::tile_store) */
    if (V_unlimited != nullptr) {
      /* what step 5 stored besides: the tiles this sweep needed in the previous updates (read before it is replaced) */
      constexpr bool kTileMode = ((MODE == kHoPlain || MODE == kHoDefer) && !SPLIT) || MODE == kHoRepair;
      const bool tiles = kTileMode && S0.tile_store != 0 && W.needed_tiles != nullptr;
      const uint32_t predicted = tiles ? tiles_predicted<MAXW>(W, r.slice) : 0u;
      bool limited = false;
      /* the row's l = min(l_ij, l_ji) and the tile masks (`lane`: see the second call below) */
      auto fetch_l = [&](const uint32_t lane) {
        needed = own_limited = 0u;
        /* in groups of up to 9 columns: all transposed positions, all l_ij and l_ji, then the masks
         * (transposed_positions(), kernels_euler.hpp) */
        constexpr int CH = MAXW - 1 < 9 ? MAXW - 1 : RYUJIN_GATHER_GROUP_3D;
        RowCtx rr = r;
        rr.lane = lane;
#pragma unroll
        for (int c0 = 1; c0 < MAXW; c0 += CH) {
          uint32_t tpos[CH];
          transposed_positions<tile_map_pays<E::DIMENSION>(), CH>(M, rr, c0, tpos);
          batch_fence();
          double l_a[CH], l_b[CH];
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            const int c = c0 + k;
            l_a[k] = l_b[k] = 1.;
            if (c < MAXW && (uint32_t)c < r.width) {
              l_a[k] = lij[(uint32_t)(((uint64_t)r.base + c) * 64 + lane)];
              l_b[k] = lij[tpos[k]];
            }
          }
          batch_fence();
#pragma unroll
          for (int k = 0; k < CH; ++k) { /* ONE wait for the group's values */
            asm volatile("" : "+v"(l_a[k]));
            asm volatile("" : "+v"(l_b[k]));
          }
#pragma unroll
          for (int k = 0; k < CH; ++k) {
            const int c = c0 + k;
            if (c >= MAXW)
              continue;
            l[c] = 1.;
            if ((uint32_t)c < r.width) {
              const bool lane_on = row_active && (uint32_t)c < r.len;
              const bool lim = lane_on && !(l_a[k] == 1. && l_b[k] == 1.);
              l[c] = lane_on ? lmin(l_a[k], l_b[k]) : 1.;
              limited = limited || lim;
              if (__any(lim))
                needed |= 1u << c;
              if (__any(lane_on && !(l_a[k] == 1.)))
                own_limited |= 1u << c;
            }
          }
        }
      };
      fetch_l(r.lane);
      const bool slice_limited = needed != 0u;
      (void)limited;
      /* P_ij stored per tile by step 5 (Stage0Src::tile_store): a tile this sweep needs -- some pair limited after
       * the symmetrisation -- that step 5 did not store -- none of its OWN l_ij limited and not read in the last
       * updates: the limit came from the neighbour's l_ji, which step 5 cannot see. Wave-uniform. */
      const uint32_t missing = (kTileMode && S0.tile_store != 0) ? (needed & ~(own_limited | predicted)) : 0u;
      if constexpr (MODE == kHoDefer) {
        /* formed outside this sweep: the slice goes on the list before anything of it is written */
        if (missing != 0u) {
          if (r.lane == 0)
            W.deferred[M.slice_begin + atomicAdd(&S0.scalars->n_deferred[M.slice_begin != 0u ? 1 : 0], 1u)] = r.slice;
          return;
        }
      }
      /* (slices the light launch found limited, todo = 3, were counted there) */
      if (S0.scalars != nullptr && (r.slice & 15u) == 0 && r.lane == 0 && (!SPLIT || group == 0) &&
          !(MODE == kHoHeavy && todo == 3)) {
        atomicAdd(&S0.scalars->n_sampled_slices, 1u);
        if (slice_limited)
          atomicAdd(&S0.scalars->n_sampled_limited, 1u);
      }
      if (W.unlimited != nullptr && r.lane == 0 && (!SPLIT || group == 0))
        W.unlimited[r.slice] = slice_limited ? 0 : 1;
      if constexpr (SPLIT && MODE == kHoRepair)
        __syncthreads(); /* every wave of the block has read the history */
      if (tiles && r.lane == 0 && (!SPLIT || group == 0))
        tiles_remember<MAXW>(W, r.slice, needed);
      load_state<K>(V_unlimited, i, U_i_new);
      if (!slice_limited) {
        if (row_active) {
          if (!SPLIT || group == 0)
            store_state<K>(new_U, i, U_i_new);
#pragma unroll
          for (int c = 1; c < MAXW; ++c)
            if ((uint32_t)c < r.len && (!SPLIT || (uint32_t)(c - 1) % kWavesPerBlock == group))
              st_stream(lij_next + ((r.base + c) * 64 + r.lane), 0.);
        }
        return;
      }
      if (kTileMode && tiles && (r.slice & 15u) == 0 && r.lane == 0 && (!SPLIT || group == 0)) {
        atomicAdd(&S0.scalars->n_sampled_tiles_needed, (unsigned int)__popc(needed));
        atomicAdd(&S0.scalars->n_sampled_tiles_formed, (unsigned int)__popc(missing));
      }
      if constexpr (kTileMode && forms_missing_tiles<E::DIMENSION, MODE>()) {
        /* the missing tiles are formed here, exactly as step 5 forms them (pij_stage0 on the same operands: the same
         * bits), and stored: the loads below, the second limiter pass, its Newton tail and step 7 then find them in
         * the matrix like every other tile. (A tile whose own pairs all went to the tail and came back with l = 1 is
         * stored already and merely written again. SPLIT: each of the block's waves forms and stores them -- the same
         * bits four times -- and reads back what it stored itself.)
         * The rare wave that comes here FETCHES ITS l AGAIN afterwards (through a lane index the compiler cannot see
         * through, or it would keep the first copy): the 2 (MAXW - 1) registers of l are then dead across the repair
         * and serve it -- in 3-D the repair would otherwise raise the kernel from 124 to 156 registers and cost every
         * wave of the sweep its fourth wave per SIMD (rounds 4 - 5 kept 3-D off per-tile storage for that reason). */
        if (missing != 0u) {
          if constexpr (MAXW > 9 && RYUJIN_TILE_REPAIR_CALL != 0)
            repair_missing_tiles_out_of_line<K>(M.cols, M.mij, M.mi_inv, S0.old_U, S0.r_in, S0.alpha, S0.dij,
                                                S0.scalars->tau, i, r.len, r.base, r.width, r.lane, row_active,
                                                missing, pij);
          else
            repair_missing_tiles<K>(M.cols, M.mij, M.mi_inv, S0.old_U, S0.r_in, S0.alpha, S0.dij, S0.scalars->tau, i,
                                    r.len, r.base, r.width, r.lane, row_active, missing, pij);
          if constexpr (MAXW > 9) {
            uint32_t lane_again = r.lane;
            asm volatile("" : "+v"(lane_again));
            fetch_l(lane_again);
          }
        }
      }
      if constexpr (E::kLimitedUpdateFromV) {
        /* U_i = V_i - sum over the limited tiles of (1 - l_ij) lambda P_ij: P_ij of those tiles only.
         * Per row, however: a row ALL of whose columns lie in limited tiles -- every strongly limited row: behind a
         * shock, next to vacuum -- has all of its P_ij at hand and forms the reference's sum in the reference's order,
         * U_i^low + sum_j l_ij lambda P_ij (hyperbolic_module.template.h:1107-1131): with l = 0 that IS U_i^low, and
         * with small l its rounding error scales with |l lambda P_ij|, not with the unlimited flux |lambda P_ij| the
         * V_i form cancels (which, where |lambda P| >> |U^low|, can leave the state outside its bounds by more than the
         * limiter's relaxation). One accumulator serves both forms: it starts from U_i^low or V_i and adds
         * w lambda P_ij with w = l or w = l - 1 = -(1 - l) (exact), the same bits as the subtraction. */
        const uint32_t row_cols = row_active ? ((r.len >= 32u ? 0xffffffffu : ((1u << r.len) - 1u)) & ~1u) : 0u;
        const bool ref_order = row_active && (needed & row_cols) == row_cols;
        if (__any(ref_order)) {
          double U_low[K];
          load_state<K>(new_U, i, U_low);
#pragma unroll
          for (int q = 0; q < K; ++q)
            U_i_new[q] = ref_order ? U_low[q] : U_i_new[q];
        }
#pragma unroll
        for (int c = 1; c < CP; ++c) {
#pragma unroll
          for (int q = 0; q < K; ++q)
            p[c][q] = 0.;
          if ((needed >> c) & 1u)
            load_entry<K>(pij, (uint64_t)r.base + c, r.lane, p[c]);
        }
#pragma unroll
        for (int c = 1; c < MAXW; ++c) {
          if (!((needed >> c) & 1u))
            continue;
          /* (columns beyond the row's length: l = 1 was set above, the V_i form adds an exact zero; the
           * reference-order form must not add the padding entry) */
          const double w = (row_active && (uint32_t)c < r.len) ? (ref_order ? l[c] : l[c] - 1.) : 0.;
          if (c < CP) {
#pragma unroll
            for (int q = 0; q < K; ++q)
              U_i_new[q] += w * lambda * p[c < CP ? c : 0][q];
          } else {
            double pt[K];
            load_entry<K>(pij, (uint64_t)r.base + c, r.lane, pt);
#pragma unroll
            for (int q = 0; q < K; ++q)
              U_i_new[q] += w * lambda * pt[q];
          }
        }
      } else {
        /* (shallow water, scalar equations) the reference's sum in the reference's order: with l = 0 it returns
         * U_i^low exactly, which a dry node relies on */
        load_state<K>(new_U, i, U_i_new);
#pragma unroll
        for (int c = 1; c < CP; ++c) {
#pragma unroll
          for (int q = 0; q < K; ++q)
            p[c][q] = 0.;
          if ((uint32_t)c < r.width)
            load_entry<K>(pij, (uint64_t)r.base + c, r.lane, p[c]);
        }
#pragma unroll
        for (int c = 1; c < MAXW; ++c) {
          if (c < CP) {
            if (row_active && (uint32_t)c < r.len) {
#pragma unroll
              for (int q = 0; q < K; ++q)
                U_i_new[q] += l[c] * lambda * p[c < CP ? c : 0][q];
            }
          } else if ((uint32_t)c < r.width) {
            double pt[K];
            load_entry<K>(pij, (uint64_t)r.base + c, r.lane, pt);
            if (row_active && (uint32_t)c < r.len) {
#pragma unroll
              for (int q = 0; q < K; ++q)
                U_i_new[q] += l[c] * lambda * pt[q];
            }
          }
        }
      }
    } else {
      load_state<K>(new_U, i, U_i_new);
#pragma unroll
      for (int c = 1; c < MAXW; ++c) {
        l[c] = 0.;
        if (c < CP) {
#pragma unroll
          for (int q = 0; q < K; ++q)
            p[c < CP ? c : 0][q] = 0.;
        }
        if ((uint32_t)c < r.width) {
          const uint64_t colbase = (uint64_t)r.base + c;
          const uint32_t pos = (uint32_t)(colbase * 64 + r.lane);
          const double l_a = lij[pos];
          const double l_b = lij[tile_transposed<tile_map_pays<E::DIMENSION>()>(M, (uint64_t)r.base + c, r.lane)];
          l[c] = lmin(l_a, l_b);
          if (c < CP)
            load_entry<K>(pij, colbase, r.lane, p[c < CP ? c : 0]);
          if (__any(row_active && (uint32_t)c < r.len && !(l[c] == 1.)))
            needed |= 1u << c;
        }
      }
#pragma unroll
      for (int c = 1; c < MAXW; ++c) {
        if (c < CP) {
          if (row_active && (uint32_t)c < r.len) {
#pragma unroll
            for (int q = 0; q < K; ++q)
              U_i_new[q] += l[c] * lambda * p[c < CP ? c : 0][q];
          }
        } else if ((uint32_t)c < r.width) {
          double pt[K];
          load_entry<K>(pij, (uint64_t)r.base + c, r.lane, pt);
          if (row_active && (uint32_t)c < r.len) {
#pragma unroll
            for (int q = 0; q < K; ++q)
              U_i_new[q] += l[c] * lambda * pt[q];
          }
        }
      }
    }
    if constexpr (SPLIT) {
      __syncthreads(); /* every wave of the block has read the old new_U[i] */
      if (row_active && group == 0)
        store_state<K>(new_U, i, U_i_new);
    } else {
      if (row_active)
        store_state<K>(new_U, i, U_i_new);
    }

    unsigned long long undecided_mask = 0;
#pragma unroll
    for (int c = 1; c < MAXW; ++c) {
      if ((uint32_t)c >= r.width)
        continue;
      if (SPLIT && (uint32_t)(c - 1) % kWavesPerBlock != group)
        continue;
      const bool lane_on = row_active && (uint32_t)c < r.len;
      if (!((needed >> c) & 1u)) {
        if (lane_on)
          st_stream(lij_next + ((r.base + c) * 64 + r.lane), 0.);
        continue;
      }
      double pc[K];
      if (c < CP) {
#pragma unroll
        for (int q = 0; q < K; ++q)
          pc[q] = p[c < CP ? c : 0][q];
      } else {
        load_entry<K>(pij, (uint64_t)r.base + c, r.lane, pc);
      }
      if (lane_on) {
        double new_p_ij[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          new_p_ij[q] = (1. - l[c]) * pc[q];
        bool success, undecided;
        const double new_l_ij =
            E::limit_fast(P, bnd, U_i_new, new_p_ij, success, undecided);
        if (undecided)
          undecided_mask |= 1ull << c;
        else
          st_stream(lij_next + ((r.base + c) * 64 + r.lane), (1. - l[c]) * new_l_ij);
      }
    }
    /* (the second pass's `success` is ignored unless EXPENSIVE_BOUNDS_CHECK, :1148-1161) */
    __shared__ uint16_t tail_queue[kWavesPerBlock * (MAXW - 1) * 64];
    __shared__ TailScratch<E> tail_rows[kWavesPerBlock];
    limit_undecided_pairs<E>(
        P, r, undecided_mask, bnd, U_i_new, tail_queue + (threadIdx.x >> 6) * (MAXW - 1) * 64,
        tail_rows[threadIdx.x >> 6].rows,
        [&](const uint32_t c, const uint32_t owner, double (&out)[K]) {
          const uint64_t pos = ((uint64_t)r.base + c) * 64 + owner;
          const double old_l_ij = lmin(lij[pos], lij[idx_t[pos]]);
          load_entry<K>(pij, (uint64_t)r.base + c, owner, out);
#pragma unroll
          for (int q = 0; q < K; ++q)
            out[q] = (1. - old_l_ij) * out[q];
        },
        [&](const uint32_t c, const uint32_t owner, const double new_l_ij) {
          const uint64_t pos = ((uint64_t)r.base + c) * 64 + owner;
          const double old_l_ij = lmin(lij[pos], lij[idx_t[pos]]);
          st_stream(lij_next + (pos), (1. - old_l_ij) * new_l_ij);
        });
  }

  /* the launch between the light and the heavy one: completes the P_ij of the slices the light launch marked */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_pij_repair(const DeviceMesh M, const Stage0Src S0, double *__restrict__ pij, const SliceFlags W)
  {
    const uint32_t slice = M.slice_begin + blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (slice >= M.slice_end || W.todo[slice] < 2)
      return;
    const uint32_t fs = W.first_stored[slice];
    const RowCtx r = row_context_of_slice(M, slice);
    backfill_pij<E::K>(M, S0, r, pij, fs == 0 ? 0xffffffffu : fs);
    if (r.lane == 0)
      W.first_stored[slice] = 1;
  }

  template <typename E, int MAXW, int CP = MAXW, bool SPLIT = false, int MODE = kHoPlain>
  __global__ void __launch_bounds__(kBlock, (MAXW > 9 ? (CP < MAXW ? RYUJIN_OCC_HO_3D : 1) : RYUJIN_OCC_HO))
  k_high_order_next_cached(const typename E::Params P, const DeviceMesh M, double *__restrict__ new_U,
                           const double *__restrict__ bounds, double *__restrict__ pij,
                           const double *__restrict__ lij, double *__restrict__ lij_next,
                           const double *__restrict__ V_unlimited = nullptr, const Stage0Src S0 = Stage0Src{},
                           const SliceFlags W = SliceFlags{})
  {
    RowCtx r;
    const uint32_t group = SPLIT ? (threadIdx.x >> 6) : 0u;
    uint32_t todo = 1;
    if constexpr (SPLIT) { /* one slice per block (uniform over the block: the barrier in the body is safe) */
      if (M.slice_begin + blockIdx.x >= M.slice_end)
        return;
      r = row_context_of_slice(M, M.slice_begin + blockIdx.x);
    } else if constexpr (MODE == kHoLight || MODE == kHoHeavy) {
      /* the flag first: most waves of a developed flow retire on it */
      const uint32_t slice = M.slice_begin + mapped_block(M) * kWavesPerBlock + (threadIdx.x >> 6);
      if (slice >= M.slice_end)
        return;
      if constexpr (MODE == kHoLight) {
        const uint32_t fs = W.first_stored[slice];
        if (fs != 0) { /* complete (1), or stored from the column on whose l_ij came out limited */
          if ((threadIdx.x & 63) == 0)
            W.todo[slice] = fs == 1 ? 1 : 2;
          return;
        }
      } else {
        todo = W.todo[slice];
        if (todo == 0)
          return;
      }
      r = row_context_of_slice(M, slice);
    } else {
      r = row_context(M);
      if (!r.valid)
        return;
    }
    next_cached_slice<E, MAXW, CP, SPLIT, MODE>(P, M, r, group, new_U, bounds, pij, lij, lij_next, V_unlimited, S0,
                                                W, todo);
  }

  /* the launch behind the sweep of step 6 where the tiles step 5 did not store are formed outside it (kHoDefer): the
   * listed slices, with the repair. A few dozen to a few hundred slices of a developed flow on an otherwise idle device:
   * what counts is the length of one slice's chain of dependent loads, not throughput (one wave per slice through the
   * kernel of the sweep: 157 us for 85 slices, profiles/r06c_kernel_trace_cylinder3d.md). Hence ONE BLOCK PER SLICE in
   * the SPLIT form of the small meshes -- every wave forms the update, wave w runs the second limiter pass of every
   * fourth column -- with ALL of the row's P_ij in registers (one wave per SIMD: the register file is the wave's) so
   * that its loads are issued back to back. */
  template <typename E, int MAXW>
  __global__ void __launch_bounds__(kBlock, 1)
  k_high_order_next_deferred(const typename E::Params P, const DeviceMesh M, double *__restrict__ new_U,
                             const double *__restrict__ bounds, double *__restrict__ pij,
                             const double *__restrict__ lij, double *__restrict__ lij_next,
                             const double *__restrict__ V_unlimited, const Stage0Src S0, const SliceFlags W)
  {
    const uint32_t n = S0.scalars->n_deferred[M.slice_begin != 0u ? 1 : 0];
    const uint32_t group = threadIdx.x >> 6;
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) { /* (uniform over the block: the barriers in the body are safe) */
      const RowCtx r = row_context_of_slice(M, W.deferred[M.slice_begin + q]);
      next_cached_slice<E, MAXW, MAXW, true, kHoRepair>(P, M, r, group, new_U, bounds, pij, lij, lij_next,
                                                        V_unlimited, S0, W);
    }
  }
} // namespace ryujin_hip

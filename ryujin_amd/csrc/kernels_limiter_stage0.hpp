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
// PER_SLICE: P_ij IS STORED ONLY WHERE STEPS 6 AND 7 WILL READ IT. A developed flow limits something in a few
// per cent (Mach-3 step early on, 3-D radial contrast) up to nearly all (cylinder channel) of its 64-row slices;
// everywhere else steps 6 and 7 never look at P_ij (V_i; l' = 0) and the 8 k S bytes per row step 5 would write --
// a third of its traffic -- are wasted. Round 3 chose between "store everywhere" and "store nowhere, form it again
// in steps 6/7" for the whole mesh from the measured fraction of limited slices: a cliff at 20 - 25 %, and a chain
// of six dependent gathers per column in every limited slice below it. Now the wave of a slice decides for itself:
//   * the slice held a limited pair in the previous update (SliceFlags::unlimited of that update's step 6 -- a
//     limited region moves by less than a cell per update): store as P_ij is formed;
//   * otherwise do not -- unless one of the slice's own l_ij comes out limited (or undecided) at column c: store
//     from c on;
//   * SliceFlags::first_stored says which. What is missing where step 6 needs P_ij -- the columns in front of c, or
//     all of a slice that turns out limited only through a neighbour's l_ji (step 6 sees that, step 5 cannot) --
//     is formed by the small repair launch inside step 6 (kernels_limiter.hpp).
// The cost of steps 5 - 7 therefore follows the limited fraction smoothly (profiles/r04*_ab_limited_fraction*); once
// most slices are limited -- a developed Mach-3 flow: 70 - 95 % -- the bookkeeping buys nothing and the host runs the
// plain kernels (P_ij stored everywhere, step 6 in one launch; ryujin_hip_ctx::step, RYUJIN_PER_SLICE_MAX_LIMITED).
// Same bits whatever is stored: whoever reads P_ij reads pij_stage0() of the same operands.
// ryujin_hip_params::debug_pij_storage: < 0 always the plain kernels, > 0 always per slice with no slice predicted
// limited (everything through the trigger / the repair launch).

#pragma once

#include "kernels_limiter.hpp"

namespace ryujin_hip
{
#ifndef RYUJIN_OCC_LIJ0
#define RYUJIN_OCC_LIJ0 3 /* waves per SIMD asked of the register allocator */
#endif
#ifndef RYUJIN_LIJ0_UNCOND
#define RYUJIN_LIJ0_UNCOND 1 /* step 5 where P_ij is stored everywhere: unconditional stores (see kUnconditionalStores) */
#endif
#ifndef RYUJIN_LIJ0_DELAY_L_MAXDIM
#define RYUJIN_LIJ0_DELAY_L_MAXDIM 3 /* step 5: l_ij of column c stored in iteration c + 1 up to this dimension. 3-D: 2 until the chained
                                         gathers took the scratch out of the kernel (the three registers came back as spill reloads inside
                                         the loop); now C3's per-slice kernel 3.17 -> 2.88 ms, C4's (unconditional stores) unchanged
                                         (profiles/r06aw_ab_delay_l_3d_c{3,4}.log) */
#endif
#ifndef RYUJIN_LIJ0_PARK_3D
#define RYUJIN_LIJ0_PARK_3D 3 /* step 5 in 3-D: 1 = the row's F_i in LDS, 2 = F_i and U_i, 3 = and alpha_i, 1 / m_i, factor; 0 = all in registers */
#endif
#ifndef RYUJIN_LIJ0_CHAIN_3D
#define RYUJIN_LIJ0_CHAIN_3D 3 /* step 5 in 3-D: chained gathers from 1 = the previous column, 2 = the slice's own rows (kernels_euler.hpp) */
#endif
#ifndef RYUJIN_OCC_LIJ0_3D
#define RYUJIN_OCC_LIJ0_3D 3 /* rounds 1-4: 2 waves (at 3 the kernel spilled 28-56 B per lane and lost). Round 5: slice context in scalar
                                 registers + F_i / U_i parked in LDS leave 12 B per lane outside the column loop: 2.22 -> 1.97 ms on the
                                 C4 share (profiles/r05c_ab_3d.log) */
#endif

#ifndef RYUJIN_OCC_LIJ0_AEOS
#define RYUJIN_OCC_LIJ0_AEOS 3 /* EulerAEOS (a fourth bound, gamma_min, powers in psi): rounds 3-4 52 B/lane of scratch at 3 waves and 2 waves faster;
                                   with the slice context in scalar registers 36 B and 0.312 -> 0.276 ms (profiles/r05e_ab_aeos.log) */
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
   * PER_SLICE: see the head of the file (NY == 1 only); otherwise P_ij is stored everywhere. */
  /* TILE: P_ij is stored per (slice, column) tile (below; NY == 1, not PER_SLICE). A template parameter, not a kernel
   * argument: with both storage schemes in one instantiation the 3-D kernel ran 5 % slower than the per-slice one at
   * equal stores (profiles/r06b_ab_tile_pij_c4.log). */
  template <typename E, int NY = 1, bool PER_SLICE = false, bool TILE = false>
  __global__ void __launch_bounds__(kBlock, lij0_waves_per_simd<E>())
  k_lij_stage0(const typename E::Params P, const DeviceMesh M, DeviceScalars *__restrict__ scalars,
               const double *__restrict__ old_U, const double *__restrict__ alpha,
               const double *__restrict__ dij, const double *__restrict__ new_U,
               const double *__restrict__ r_in, const double *__restrict__ bounds, double *__restrict__ pij,
               double *__restrict__ lij, double *__restrict__ V_out, const SliceFlags W = SliceFlags{},
               const int predict_override = 0)
  {
    static_assert(!PER_SLICE || NY == 1, "one wave per slice decides");
    static_assert(!TILE || (NY == 1 && !PER_SLICE), "per tile: the plain kernel, one wave per slice");
    constexpr int K = E::K;
    constexpr int NB = E::NB;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);

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
    /* 3-D: F_i (and U_i; and the row's alpha_i, 1 / m_i and factor) live in LDS across the column loop
     * (pij_stage0_parked). The rows share their LDS with the scratch of the Newton tail behind the loop
     * (TailScratch: the two are never alive together), which is what makes room for the three scalars: without them
     * the kernel reloads a spilled register INSIDE the loop, and a scratch load counts in vmcnt like any other --
     * the compiler follows it with vmcnt(0), which drains the prefetch of the next column a third of the way
     * into the iteration (scripts/isa_loop_waits.sh). */
    constexpr int kPark = (E::DIMENSION == 3 && NY == 1) ? RYUJIN_LIJ0_PARK_3D : 0; /* 0 none, 1 F_i, 2 F_i and U_i, 3 and the scalars */
    constexpr int kParkedDoubles = kPark == 0 ? 0 : (2 * K + (kPark == 3 ? 3 : 0)) * 64;
    constexpr int kRowDoubles = kParkedDoubles > TailScratch<E>::kRowDoubles ? kParkedDoubles : TailScratch<E>::kRowDoubles;
    __shared__ double lds_rows[kWavesPerBlock * kRowDoubles];
    double *const parked = lds_rows + (threadIdx.x >> 6) * kRowDoubles;
    row.alpha_i = alpha[i];
    row.m_i_inv = M.mi_inv[i];
    row.factor = scalars->tau * row.m_i_inv * (double)(r.len - 1);
    if constexpr (kPark != 0) {
#pragma unroll
      for (int q = 0; q < K; ++q) {
        parked[q * 64 + r.lane] = row.F_i[q];
        if (kPark >= 2)
          parked[(K + q) * 64 + r.lane] = row.U_i[q];
      }
      if (kPark == 3) {
        parked[(2 * K + 0) * 64 + r.lane] = row.alpha_i;
        parked[(2 * K + 1) * 64 + r.lane] = row.m_i_inv;
        parked[(2 * K + 2) * 64 + r.lane] = row.factor;
      }
    }
    const double lambda = 1. / (double)(r.len - 1);
    bool all_ok = true;
    unsigned long long undecided_mask = 0;

    const uint32_t c0 = 1 + (NY > 1 ? blockIdx.y : 0);
    /* wave-uniform: P_ij of this slice goes to the matrix; from column first_stored on */
    bool storing = true;
    uint32_t first_stored = c0;
    if constexpr (PER_SLICE)
      storing = predict_override < 0 || (predict_override == 0 && W.unlimited[r.slice] == 0);
    /* PER TILE (plain kernels, NY == 1; the host's choice, Stage0Src::tile_store): a (slice, column) tile is stored
     * iff one of its own l_ij comes out limited or goes to the Newton tail -- behind the limiter, wave-uniformly.
     * Steps 6/7 read P_ij only in tiles that hold a limited pair after the symmetrisation; the tiles that are limited
     * through the neighbour's l_ji alone are formed by step 6 (next_cached_slice). On the developed C2 flow 93 % of the
     * slices but 43 % of the tiles hold a limited pair: most of the 8 k S bytes per row this sweep used to write were
     * never read. */
    constexpr bool tile_mode = TILE;
    /* gfx9 has ONE counter for vector loads and stores (vmcnt), decremented in issue order. A store the compiler sees
     * on some paths of the column loop only -- `if (active) store` -- makes it wait for vmcnt(0) before the first use of
     * the next column's operands: every column then pays the round trip of the stores of the column before
     * (profiles/r06i_*). Where everything is stored anyway, every lane stores, every column: the number of
     * stores behind the loads is the same on every path and the wait becomes vmcnt(n > 0). */
    constexpr bool kUnconditionalStores = RYUJIN_LIJ0_UNCOND != 0 && !PER_SLICE && !TILE && NY == 1;
    uint32_t tiles_stored = 0;
    /* ... or step 6 of the PREVIOUS update needed it (SliceFlags::needed_tiles, bit c of the slice's word): fronts
     * move a fraction of a cell per update, so this predicts nearly every tile that is limited through l_ji alone,
     * and step 6 forms what is left (a tile predicted in vain costs its store, as before). */
    constexpr int kMaxWidth = E::DIMENSION == 1 ? 3 : (E::DIMENSION == 2 ? 9 : 27);
    const uint32_t predicted = (tile_mode && W.needed_tiles != nullptr) ? tiles_predicted<kMaxWidth>(W, r.slice) : 0u;

    /* SOFTWARE PIPELINE, arranged around gfx9's single in-order counter for vector loads and stores (vmcnt): whatever
     * the wave waits for, it waits for everything it issued before that as well, and where the compiler cannot count
     * the operations behind a load on every path -- conditional stores, the index load of an irregular tile -- it
     * waits for vmcnt(0): all of it. Rounds 1 - 5 issued the l_ij store of column c as the last thing of its iteration
     * and, in 3-D, the index load of column c + 1 right in front of the gathers that need it: the first wait of every
     * iteration then covered a store and an index load that had just been issued -- two round trips per column
     * (scripts/isa_loop_waits.sh k_lij_stage0). Now
     *   - the column index runs TWO columns ahead (j_nn), the operands of the pair one column ahead (next);
     *   - the l_ij of column c is stored in iteration c + 1, BEHIND the loads of that iteration, so that it has a whole
     *     limiter to retire behind before anything waits for it;
     *   - where everything is stored anyway, every lane stores in every column (kUnconditionalStores).
     * (the column index of a structured tile is row + delta of the tile's descriptor, kernels_euler.hpp) */
    constexpr bool kTileMap = tile_map_pays<E::DIMENSION>();
    constexpr bool kChained = NY == 1; /* (one wave walks all the columns of the slice; the chain codes exist in every dimension) */
    uint32_t j_n = r.width > c0 ? tile_column<kTileMap>(M, (uint64_t)r.base + c0, i, r.lane) : i;
    uint32_t j_nn = r.width > c0 + NY ? tile_column<kTileMap>(M, (uint64_t)r.base + c0 + NY, i, r.lane) : i;
    PairData<K> next;
    if (r.width > c0)
      load_pair<K>(M, old_U, r_in, alpha, dij, ((uint64_t)r.base + c0) * 64 + r.lane, j_n, next);
    /* the l_ij of the previous column, not stored yet (l_pending_on: there is one). (3-D: until the chained gathers
     * the kernel sat at the register limit of three waves per SIMD and the three registers this takes came back as
     * spill reloads INSIDE the loop -- scratch loads count in vmcnt as well; RYUJIN_LIJ0_DELAY_L_MAXDIM) */
    constexpr bool kDelayL = RYUJIN_LIJ0_DELAY_L_MAXDIM >= E::DIMENSION;
    double l_pending = 1.;
    bool l_pending_on = false;

    for (uint32_t c = c0; c < r.width; c += NY) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const bool active = row_active && c < r.len;
      double P_ij[K];
      if constexpr (kPark != 0)
        pij_stage0_parked<K, kPark >= 2, kPark == 3>(row, parked, r.lane, next, P_ij);
      else
        pij_stage0<K>(row, next, P_ij);
      if (c + NY < r.width) {
        j_n = j_nn;
        /* (chained gathers, kernels_euler.hpp: the node data of most columns is the previous column's, or the
         * slice's own rows', moved by a lane) */
        constexpr bool kMasks = chain_masks_pay<E::DIMENSION>();
        TileChain chain = kChained ? tile_chain<kMasks>(M, colbase + NY) : TileChain{kChainNone, ~0ull};
        if constexpr (E::DIMENSION == 3) {
          if ((RYUJIN_LIJ0_CHAIN_3D & 1) == 0 && chain.kind == kChainPrevColumn)
            chain.kind = kChainNone;
          if ((RYUJIN_LIJ0_CHAIN_3D & 2) == 0 && chain.kind != kChainPrevColumn)
            chain.kind = kChainNone;
        }
        if (chain.kind == kChainNone)
          load_pair<K>(M, old_U, r_in, alpha, dij, (colbase + NY) * 64 + r.lane, j_n, next);
        else {
          uint32_t lane_c = r.lane;
          if constexpr (kPark != 0) {
            /* (the parked rows are read once P_ij is complete and `next` is free: read earlier -- the scheduler would
             * -- the new operands need registers of their own next to the old ones, and the kernel has none) */
#pragma unroll
            for (int q = 0; q < K; ++q)
              asm volatile("" : "+v"(P_ij[q]));
            asm volatile("" : "+v"(lane_c));
          }
          load_pair_chained<K, kPark, kMasks>(M, old_U, r_in, alpha, dij, (colbase + NY) * 64 + r.lane, j_n, chain, lane_c, row,
                                      parked, next);
        }
        j_nn = c + 2 * NY < r.width ? tile_column<kTileMap>(M, colbase + 2 * NY, i, r.lane) : i;
      }
      /* the l_ij of the column before */
      if constexpr (kDelayL) {
        if constexpr (kUnconditionalStores) {
          if (c > c0)
            lij[pos - NY * 64] = l_pending;
        } else if (l_pending_on)
          lij[pos - NY * 64] = l_pending;
      }
      /* a slice that stores already: as soon as P_ij is formed (the store overlaps the limiter) */
      const bool stored_early = storing && (!tile_mode || ((predicted >> c) & 1u) != 0u);
      if (tile_mode && stored_early)
        ++tiles_stored;
      if constexpr (kUnconditionalStores)
        store_entry<K>(pij, colbase, r.lane, P_ij); /* (every lane: the padding slots of the slice absorb the inactive ones) */
      else if (stored_early && active)
        store_entry<K>(pij, colbase, r.lane, P_ij);
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
      if constexpr (PER_SLICE) {
        if (!storing && __any(active && (undecided || !(l_ij == 1.)))) {
          storing = true;
          first_stored = c;
        }
      }
      if (tile_mode && !stored_early && __any(active && (undecided || !(l_ij == 1.)))) {
        ++tiles_stored;
        if (active)
          store_entry<K>(pij, colbase, r.lane, P_ij);
      }
      if constexpr (PER_SLICE) {
        if (active && storing && !stored_early)
          store_entry<K>(pij, colbase, r.lane, P_ij);
      }
      /* (an undecided pair's entry: a placeholder in the unconditional form, which the Newton tail overwrites behind
       * its fence; nothing otherwise) */
      if constexpr (kDelayL) {
        l_pending = (active && !undecided) ? l_ij : 1.;
        l_pending_on = active && !undecided;
      } else if constexpr (kUnconditionalStores)
        lij[pos] = (active && !undecided) ? l_ij : 1.;
      else if (active && !undecided)
        lij[pos] = l_ij;
      if (active && undecided)
        undecided_mask |= 1ull << c;
      all_ok = all_ok && (!active || undecided || success);
    }
    /* the l_ij of the last column */
    if (kDelayL && r.width > c0) {
      const uint32_t c_last = c0 + ((r.width - 1 - c0) / NY) * NY;
      if (kUnconditionalStores || l_pending_on)
        lij[((uint64_t)r.base + c_last) * 64 + r.lane] = l_pending;
    }
    if (NY == 1 && V_out != nullptr && row_active)
      store_state<K>(V_out, i, V_i);

    if (tile_mode && (r.slice & 15u) == 0 && r.lane == 0 && r.width > 1) {
      atomicAdd(&scalars->n_sampled_tiles, r.width - 1);
      atomicAdd(&scalars->n_sampled_tiles_stored, tiles_stored);
    }
    if constexpr (PER_SLICE) {
      if (r.lane == 0) {
        W.first_stored[r.slice] = storing ? (uint8_t)first_stored : 0;
        if ((r.slice & 15u) == 0 && storing)
          atomicAdd(&scalars->n_sampled_stored, 1u);
      }
    }

    /* the few pairs that need the Newton iteration (an undecided pair makes its slice store: P_ij is there),
     * compacted over the wave (limit_undecided_pairs, kernels_limiter.hpp) */
    constexpr int kTailColumns = E::DIMENSION == 1 ? 2 : (E::DIMENSION == 2 ? 8 : 26);
    __shared__ uint16_t tail_queue[kWavesPerBlock * kTailColumns * 64];
    const bool tail_ok = limit_undecided_pairs<E>(
        P, r, undecided_mask, bnd, U_i_new, tail_queue + (threadIdx.x >> 6) * kTailColumns * 64,
        parked /* (the rows of TailScratch: the column loop is done with them) */,
        [&](const uint32_t c, const uint32_t owner, double (&out)[K]) {
          load_entry<K>(pij, (uint64_t)r.base + c, owner, out);
        },
        [&](const uint32_t c, const uint32_t owner, const double l_ij) {
          lij[((uint64_t)r.base + c) * 64 + owner] = l_ij;
        });
    flag_restart(scalars, all_ok && tail_ok, r.lane);
  }

  /* ryujin_hip_debug_fetch(P_ij) behind a step that did not store all of it: the same pij_stage0() on the same
   * operands (all of them outlive the step), written to the matrix the parity tests read for the columns the
   * sweeps left out */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_pij_stage0_store(const DeviceMesh M, const Stage0Src S0, double *__restrict__ pij,
                     const uint8_t *__restrict__ first_stored)
  {
    const RowCtx r = row_context(M);
    if (!r.valid || r.len <= 1)
      return;
    const uint32_t fs = first_stored != nullptr ? first_stored[r.slice] : 0u; /* NULL: stored per tile, form all */
    if (fs == 1)
      return;
    backfill_pij<E::K>(M, S0, r, pij, fs == 0 ? 0xffffffffu : fs);
  }
} // namespace ryujin_hip

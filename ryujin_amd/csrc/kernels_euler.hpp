// HIP kernels for HyperbolicModule::prepare_state_vector / ::step (Euler), gfx950.
//
// Mapping: one thread per DoF row, one wavefront (64 lanes) per SELL-64 slice. For every col_idx
// the 64 lanes of a wave read 64 consecutive matrix entries (512 B / 1 KiB per load instruction),
// so the c_ij / m_ij / d_ij / l_ij / p_ij streams are coalesced without an LDS transpose; states
// U_j, r_j, alpha_j are gathered through L1/L2 (for a locality-preserving numbering consecutive
// lanes gather consecutive j). No MFMA: the path is a bandwidth-bound stencil sweep.
//
// Sweeps follow the reference one to one (source/hyperbolic_module.template.h):
//   k_precompute_records (boundary conditions      step 1  :96-193
//     folded in: apply_bc_row)
//     (+ k_ghost_precompute_records on the ghost rows)
//   k_dij_alpha_records (k_alpha + k_dij_records,  step 2  :341-424
//     k_dij_alpha for rows wider than 32)
//   k_dij_boundary + k_dij_diag[_unrolled]         step 3  :432-564
//   k_low_order (finalize_tau)                     step 4  :597-884
//   k_lij_stage0 (no stage vectors: P_ij formed    step 5  :892-1041
//     here, kernels_limiter_stage0.hpp) /
//     k_pij_lij[_recompute]
//   k_high_order_next_cached (a light and a heavy  step 6  :1053-1182
//     launch where P_ij is stored per slice)
//     / k_high_order<false>
//   k_high_order_last_cached / k_high_order<true>  step 7  :1053-1182

#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "euler_device.hpp"

namespace ryujin_hip
{
  struct DeviceScalars;

  /* Start of a step (tau_max := tau_max_in, flags := 0, the restart flag of the previous Runge-Kutta stage
   * folded into its accumulator, the arguments finalize_tau() needs), carried by the first sweep of step 2 and
   * executed by the first thread of its launches: the kernels of step 2 do not touch the scalars, a kernel
   * boundary separates them from the previous step and from step 3. Idempotent (a split sweep runs it in
   * its export and in its interior launch). A launch of its own cost ~10 us of dependency latency per update. */
  struct StepBegin {
    DeviceScalars *scalars; /* NULL: this launch does not begin a step */
    double tau_max_in, tau_in;
    int reset_accumulators;
    int accumulate_stage; /* < 0: nothing to fold */
    int use_device_tau, stage;
  };

  struct DeviceMesh {
    StepBegin begin;
    uint32_t n_owned, n_relevant, n_slices;
    uint32_t bounds_stride; /* limiter bounds are SoA [n_bounds][bounds_stride], bounds_stride >= n_relevant */
    uint32_t slice_begin, slice_end; /* slice range of this launch (export rows first, then interior) */
    const uint32_t *slice_off; /* [n_slices+1] */
    const uint16_t *row_len;    /* [n_slices*64] */
    const uint32_t *cols;      /* [nnz_total] */
    const uint32_t *idx_t;     /* [nnz_total] */
    const TileDesc *tiles;     /* [slice_off[n_slices]] the tile map (host_layout.hpp), or NULL: explicit arrays only */
    const uint64_t *chain_loads; /* [slice_off[n_slices]] with the tile map: the lanes of a chained tile that fetch their node (TileDesc::chain) */
    uint32_t tail_queue_columns; /* min(63, widest row - 1): columns of the dynamic-LDS queue of undecided pairs (k_pij_lij) */
    /* > 1: the four waves of a block take slices that are band_stride apart (one lattice row / plane of a structured
     * patch) instead of four consecutive ones: row_context() */
    uint32_t band_stride;
    /* > 0: blocks renumbered in chunks of 8 * xcd_chunk so that each XCD takes xcd_chunk consecutive blocks of a chunk
     * (ryujin_hip_params::debug_xcd_chunk): row_context() */
    uint32_t xcd_chunk;
    const double *cij;         /* paired layout, DIM comps */
    const double *mij;
    const double *mi, *mi_inv;
    double measure_of_omega_inverse;
    /* discontinuous ansatz only (NULL otherwise): incidence matrix, full inverse mass matrix, SELL-64 */
    const double *incidence, *mass_matrix_inverse;
  };

  /* device scalars shared between sweeps */
  struct DeviceScalars {
    unsigned long long tau_max_bits; /* atomicMin over positive doubles */
    double tau;                      /* the tau used by steps 4,5 */
    int restart_needed;
    int tau_invalid;
    /* device-resident Runge-Kutta driver (ryujin_hip_time_step): the tau of the first stage and the
     * flags of all stages stay on the device until the end of the RK step */
    double tau_rk;
    /* 0 = never raised, otherwise kStageCode - (index of the FIRST stage that raised it): a max-reduction
     * over stages and ranks then yields the earliest stage, which the host needs to give a Restart raised
     * at the end of stage s precedence over an invalid tau_max of a later stage (the reference never runs
     * that later stage: hyperbolic_module.template.h:1194-1207 throws first) */
    int restart_accum;
    int tau_invalid_accum;
    /* arguments of the running step, stored by step_begin() for finalize_tau() */
    double tau_in;
    int use_device_tau;
    int stage;
    /* running counters (never reset on the device; the host takes differences): every 16th slice that passes the
     * first high-order sweep, those of them in which some pair was limited, and every 16th slice whose P_ij step 5
     * stored (diagnostics: ryujin_hip_limiter_statistics) */
    unsigned int n_sampled_slices, n_sampled_limited, n_sampled_stored;
    /* ... and, where step 5 stores P_ij per tile, the tiles of every 16th slice and those of them it stored */
    unsigned int n_sampled_tiles, n_sampled_tiles_stored;
    /* ... those step 6 read, and those of them it had to form itself (step 5 had not stored them) */
    unsigned int n_sampled_tiles_needed, n_sampled_tiles_formed;
    /* slices on the list of the launch behind step 6 (SliceFlags::deferred; [0] the export or only part of the sweep,
     * [1] the interior part of a split sweep); reset by step_begin() */
    unsigned int n_deferred[2];
  };
  constexpr int kStageCode = 100;

  RYUJIN_DEV void step_begin(const DeviceMesh &M)
  {
    DeviceScalars *scalars = M.begin.scalars;
    if (!scalars || blockIdx.x != 0 || threadIdx.x != 0)
      return;
    if (M.begin.reset_accumulators) {
      scalars->restart_accum = 0;
      scalars->tau_invalid_accum = 0;
    } else if (M.begin.accumulate_stage >= 0) {
      if (scalars->restart_needed && scalars->restart_accum < kStageCode - M.begin.accumulate_stage)
        scalars->restart_accum = kStageCode - M.begin.accumulate_stage;
    }
    scalars->tau_max_bits = (unsigned long long)__double_as_longlong(M.begin.tau_max_in);
    scalars->restart_needed = 0;
    scalars->tau_invalid = 0;
    scalars->tau_in = M.begin.tau_in;
    scalars->use_device_tau = M.begin.use_device_tau;
    scalars->stage = M.begin.stage;
    scalars->n_deferred[0] = scalars->n_deferred[1] = 0;
  }


  /* minimum waves per SIMD requested from the register allocator for the heavy sweeps (second
   * __launch_bounds__ argument): 512 registers / waves. Tuned on MI355X, see DESIGN.md. */
#ifndef RYUJIN_PIN_WAITS
#define RYUJIN_PIN_WAITS 1 /* arrived() in the column loops (0: A/B, the compiler's own placement of the waits) */
#endif
#ifndef RYUJIN_HO_CP_3D
#define RYUJIN_HO_CP_3D 2 /* step 6 in 3-D: 0 = two-pass kernel, n = l_ij and the first n P_ij columns in registers (the others are read a second time, unless the whole tile is unlimited). Round 1, all at 2 waves/SIMD: 3.07 ms (0), 2.43 (8), 2.24 (14), 2.36 (18); round 2 see RYUJIN_OCC_HO_3D; round 4 (developed C4 state, all slices limited, limited update from V_i with tile-predicated P loads, 3 waves): whole update 8.30 ms (6), 8.06 (3), 8.02 (2), 8.13 (1) -- the second read of an unlimited tile is skipped anyway, fewer held columns leave the registers to the loads in flight */
#endif
#ifndef RYUJIN_OCC_DIJ_NODE_RECORD
#define RYUJIN_OCC_DIJ_NODE_RECORD 3 /* step 2 on the combined node record (3-D Euler): 168 registers without scratch; C4 share 1.47 ms at 2 waves (184 registers), 1.35 at 3 */
#endif
#ifndef RYUJIN_DIJ_PREFETCH_RECORD
#define RYUJIN_DIJ_PREFETCH_RECORD 1 /* step 2 on node records: the next neighbour's record is loaded one column ahead */
#endif
#ifndef RYUJIN_HO_CP_2D
#define RYUJIN_HO_CP_2D 9 /* step 6 in 2-D: P_ij columns kept in registers between the update and the second limiter pass (9: all) */
#endif
#ifndef RYUJIN_OCC_HO_3D
#define RYUJIN_OCC_HO_3D 3 /* step 6 in 3-D: waves per SIMD asked of the register allocator. A/B on MI355X (4.2 M gridpoints): CP 14 at 2 waves 1.96 ms, CP 6 at 3 waves 1.72 ms */
#endif
#ifndef RYUJIN_OCC_LAST_3D
#define RYUJIN_OCC_LAST_3D 4 /* step 7 in 3-D. A/B (4.2 M gridpoints): chunk 9 at 2 waves 0.99 ms, 5 at 3 waves 0.76 ms, 3 at 4 waves 0.68 ms */
#endif
#ifndef RYUJIN_LAST_CHUNK_3D
#define RYUJIN_LAST_CHUNK_3D 3 /* step 7 in 3-D: P_ij columns whose loads are issued back to back */
#endif
#ifndef RYUJIN_LAST_CHUNK_2D
#define RYUJIN_LAST_CHUNK_2D 3 /* A/B on C2: step 7 0.118 -> 0.104 ms (6 instead of 4 waves per SIMD) */
#endif
#ifndef RYUJIN_OCC_DIJ
#define RYUJIN_OCC_DIJ 2
#endif
#ifndef RYUJIN_LOW_PARK
#define RYUJIN_LOW_PARK 1 /* step 4 (1-D, 2-D, no stage vectors): f(U_i) in LDS across the column loop */
#endif
#ifndef RYUJIN_OCC_LOW
#define RYUJIN_OCC_LOW 3 /* C2, profiles/r05e_ab_low_order_2d.log: 0.2496 ms at 2 waves per SIMD, 0.2751 at 3 (36 B per lane of
                            scratch), 0.2408 at 3 with f(U_i) parked in LDS (two 8-byte spills per column left) */
#endif
#ifndef RYUJIN_OCC_LOW_AEOS
#define RYUJIN_OCC_LOW_AEOS 2 /* k_low_order_aeos, kernels_euler_aeos.hpp */
#endif
#ifndef RYUJIN_OCC_LOW_SW
#define RYUJIN_OCC_LOW_SW 2 /* k_low_order_sw, kernels_shallow_water.hpp */
#endif
#ifndef RYUJIN_OCC_LOW_3D_ALL
#define RYUJIN_OCC_LOW_3D_ALL 0
#endif
#ifndef RYUJIN_OCC_LOW_3D_STAGES
#define RYUJIN_OCC_LOW_3D_STAGES 1 /* 3-D multi-stage step 4: 1 wave/SIMD without spills instead of 2 with 212 B/lane of scratch */
#endif
#ifndef RYUJIN_OCC_PIJ
#define RYUJIN_OCC_PIJ 2
#endif
#ifndef RYUJIN_OCC_HO
#define RYUJIN_OCC_HO 2
#endif
#ifndef RYUJIN_STAGE0_PIJ
#define RYUJIN_STAGE0_PIJ 1 /* Euler, stages == 0: P_ij formed once, in step 5 (kernels_limiter_stage0.hpp) */
#endif
#ifndef RYUJIN_PER_SLICE_PIJ
#define RYUJIN_PER_SLICE_PIJ 1 /* stages == 0, two limiter passes: step 5 stores P_ij only in the slices steps 6/7 will read it in (kernels_limiter_stage0.hpp); 0: everywhere */
#endif
#ifndef RYUJIN_PER_SLICE_MAX_LIMITED
#define RYUJIN_PER_SLICE_MAX_LIMITED 0.8 /* ... while at most this fraction of the slices held a limited pair in the latest measured update. Break-even on the developed Mach-3 step (profiles/r04f_ab_per_slice_vs_plain_real_flow_2d.log): per slice -1.9 % per update at 53 % limited slices, -0.5 % at 71 %, +1.2 % at 93 % */
#endif
#ifndef RYUJIN_FUSE_PRECOMPUTE
#define RYUJIN_FUSE_PRECOMPUTE 1 /* device-resident RK driver: the last sweep of a stage leaves the precomputed values and Riemann records of the next one (FusedPrecompute) */
#endif

  constexpr int kBlock = 256;
  constexpr int kWavesPerBlock = kBlock / 64;

  template <int K>
  struct StatePad {
    static constexpr int KP = (K + 1) / 2 * 2;
  };

  template <int K>
  RYUJIN_DEV void load_state(const double *__restrict__ U, const uint32_t i, double (&v)[K])
  {
    constexpr int KP = StatePad<K>::KP;
    const double2 *b = reinterpret_cast<const double2 *>(U + (size_t)i * KP);
#pragma unroll
    for (int g = 0; g < KP / 2; ++g) {
      const double2 t = b[g];
      v[2 * g] = t.x;
      if (2 * g + 1 < K)
        v[2 * g + 1] = t.y;
    }
  }

  template <int K>
  RYUJIN_DEV void store_state(double *__restrict__ U, const uint32_t i, const double (&v)[K])
  {
    constexpr int KP = StatePad<K>::KP;
    double2 *b = reinterpret_cast<double2 *>(U + (size_t)i * KP);
#pragma unroll
    for (int g = 0; g < KP / 2; ++g) {
      double2 t;
      t.x = v[2 * g];
      t.y = (2 * g + 1 < K) ? v[2 * g + 1] : 0.;
      b[g] = t;
    }
  }

#ifndef RYUJIN_NT
#define RYUJIN_NT 3 /* non-temporal hints on the single-use multi-component matrix streams (c_ij, P_ij):
                       bit 0 loads, bit 1 stores. A/B on MI355X: -2..3 % per update (the streams no longer
                       evict the gathered U_j / l_ji lines from L2) */
#endif
  typedef double v2d_t __attribute__((ext_vector_type(2)));

  /* single-use scalar streams (column indices, m_ij, l'_ij): bits 2 (loads) and 3 (stores) */
  template <typename T>
  RYUJIN_DEV T ld_stream(const T *p)
  {
#if RYUJIN_NT & 4
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
  }
  template <typename T, typename V>
  RYUJIN_DEV void st_stream(T *p, const V v)
  {
#if RYUJIN_NT & 8
    __builtin_nontemporal_store((T)v, p);
#else
    *p = (T)v;
#endif
  }

  /* entry of an NC-component matrix in the paired SELL layout; colbase = slice_off + col_idx */
  template <int NC>
  RYUJIN_DEV void load_entry(const double *__restrict__ m, const uint64_t colbase,
                             const uint32_t lane, double (&v)[NC])
  {
    const double *b = m + colbase * 64 * NC;
#pragma unroll
    for (int g = 0; g < NC / 2; ++g) {
#if RYUJIN_NT & 1
      const v2d_t t = __builtin_nontemporal_load(reinterpret_cast<const v2d_t *>(b + g * 128 + lane * 2));
#else
      const double2 t = *reinterpret_cast<const double2 *>(b + g * 128 + lane * 2);
#endif
      v[2 * g] = t.x;
      v[2 * g + 1] = t.y;
    }
    if (NC & 1)
      v[NC - 1] = b[(NC / 2) * 128 + lane];
  }

  /* the same entry with an ordinary (temporal) load: for streams a kernel reads a second time */
  template <int NC>
  RYUJIN_DEV void load_entry_cached(const double *__restrict__ m, const uint64_t colbase,
                                    const uint32_t lane, double (&v)[NC])
  {
    const double *b = m + colbase * 64 * NC;
#pragma unroll
    for (int g = 0; g < NC / 2; ++g) {
      const double2 t = *reinterpret_cast<const double2 *>(b + g * 128 + lane * 2);
      v[2 * g] = t.x;
      v[2 * g + 1] = t.y;
    }
    if (NC & 1)
      v[NC - 1] = b[(NC / 2) * 128 + lane];
  }

  template <int NC>
  RYUJIN_DEV void store_entry(double *__restrict__ m, const uint64_t colbase, const uint32_t lane,
                              const double (&v)[NC])
  {
    double *b = m + colbase * 64 * NC;
#pragma unroll
    for (int g = 0; g < NC / 2; ++g) {
#if RYUJIN_NT & 2
      v2d_t t;
      t.x = v[2 * g];
      t.y = v[2 * g + 1];
      __builtin_nontemporal_store(t, reinterpret_cast<v2d_t *>(b + g * 128 + lane * 2));
#else
      double2 t;
      t.x = v[2 * g];
      t.y = v[2 * g + 1];
      *reinterpret_cast<double2 *>(b + g * 128 + lane * 2) = t;
#endif
    }
    if (NC & 1)
      b[(NC / 2) * 128 + lane] = v[NC - 1];
  }


  RYUJIN_DEV double wave_min(double x)
  {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
      x = fmin(x, __shfl_xor(x, off, 64));
    return x;
  }

  /* tau_max: one device-scope atomicMin per wave would serialise ~40k atomics on one address
   * (~12 ns each = the whole sweep); the running minimum only decreases, so a wave first peeks at
   * it with a relaxed load and skips the atomic unless it can lower it. */
  RYUJIN_DEV void publish_tau_min(DeviceScalars *scalars, const double tau, const uint32_t lane)
  {
    if (lane == 0 && tau < DBL_MAX) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(tau);
      const unsigned long long cur =
          __hip_atomic_load(&scalars->tau_max_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (bits < cur)
        atomicMin(&scalars->tau_max_bits, bits);
    }
  }

  struct RowCtx {
    uint32_t slice, lane, row, len, base, width;
    bool valid;
  };

  /* ---- the tile map (TileDesc, host_layout.hpp): column index and transposed position of the entry of row `row`
   * (lane `lane` of its slice) in the tile `colbase`, from the tile's 16-byte descriptor where the tile is regular,
   * from the explicit arrays where it is not. The descriptor load has a wave-uniform address.
   * WHERE IT IS USED is decided by measurement (profiles/r05c_ab_tile_2d.log, r05c_ab_3d.log): the bandwidth-bound
   * sweeps of the 1-D / 2-D stencils (steps 3, 5, 6, 7: -10 %, -4.9 %, -3.8 %, -7.0 % on C2). The sweeps that are
   * bound by FP64 issue lose on the handful of extra instructions per column -- steps 2 and 4 in 2-D (+3 %, 0),
   * every sweep in 3-D (step 2 +5 %, step 6 +8 %) -- and keep streaming the explicit indices: USE = false compiles
   * the map out. ---- */
  template <int DIM>
  constexpr bool tile_map_pays()
  {
    return DIM <= 2;
  }
  inline bool tile_map_pays(const int dim) { return dim <= 2; } /* host: whether create() builds the map at all */

  /* A tile descriptor through the CONSTANT address space (the map is written at create() and by no kernel): with a
   * wave-uniform address that is a scalar load -- no vector-memory round trip in front of the gathers that need it,
   * nothing in vmcnt -- whatever the kernel stores or pins around it. RYUJIN_TILE_DESC_SCALAR = 0: the global loads of
   * rounds 5 - 6 (A/B). */
#ifndef RYUJIN_DIAG_PINS
#define RYUJIN_DIAG_PINS 1 /* step 3: 1 = the gathers of a row in flight together, 2 = its stores as well. Same process
                             (profiles/r06ar_ab_step3_pins_c{2,5}.log): C2 0.0767 -> 0.0710 (1) / 0.0727 (2) / 0.0911 ms (3: a
                             row's eight reads and four writes at once), C5 0.1024 -> 0.0951 / 0.0989 / 0.0960 */
#endif
#ifndef RYUJIN_TILE_DESC_SCALAR
#define RYUJIN_TILE_DESC_SCALAR 1
#endif
  RYUJIN_DEV int4 load_tile_desc(const TileDesc *t)
  {
#if RYUJIN_TILE_DESC_SCALAR
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef const v4i __attribute__((address_space(4))) *const_ptr;
    const v4i v = *(const_ptr)(uintptr_t)t;
    return int4{v.x, v.y, v.z, v.w};
#else
    return *reinterpret_cast<const int4 *>(t);
#endif
  }

  template <bool USE = true>
  RYUJIN_DEV TileDesc tile_desc(const DeviceMesh &M, const uint64_t colbase)
  {
    TileDesc t;
    t.delta = kTileIrregular;
    t.ta = t.tb = t.chain = 0;
    if constexpr (USE) {
      if (M.tiles != nullptr) {
        const int4 raw = load_tile_desc(M.tiles + colbase);
        t.delta = __builtin_amdgcn_readfirstlane(raw.x);
        t.ta = (uint32_t)__builtin_amdgcn_readfirstlane(raw.y);
        t.tb = (uint32_t)__builtin_amdgcn_readfirstlane(raw.z);
      }
    }
    return t;
  }

  /* (the column index needs the first word of the descriptor only) */
  template <bool USE = true>
  RYUJIN_DEV uint32_t tile_column(const DeviceMesh &M, const uint64_t colbase, const uint32_t row, const uint32_t lane)
  {
    if constexpr (USE) {
      if (M.tiles != nullptr) {
        const int32_t delta = __builtin_amdgcn_readfirstlane(M.tiles[colbase].delta);
        if (delta != kTileIrregular) /* wave-uniform */
          return (uint32_t)((int32_t)row + delta);
      }
    }
    return ld_stream(M.cols + (colbase * 64 + lane));
  }

  /* ---- CHAINED GATHERS (TileDesc::chain and chain_loads, host_layout.hpp). On a lattice the node data of the columns
   * i + d - 1, i + d, i + d + 1 are the same 64 nodes shifted by a lane, so one of three is fetched and the others are
   * moved across the wave (v_mov_b32_dpp wave_shl:1 / wave_shr:1: one instruction per dword, no memory); the
   * neighbours i - 1 and i + 1 are the slice's own rows. The lanes for which that does not hold -- the end of the
   * wave, the end of a lattice row, boundary rows: the tile's mask -- fetch their node as ever. Same values, same
   * bits. ---- */
#ifndef RYUJIN_CHAINED_GATHERS
#define RYUJIN_CHAINED_GATHERS 1
#endif
  struct TileChain {
    uint32_t kind;  /* kChainNone / kChainPrevColumn / kChainOwnPrev / kChainOwnNext, wave-uniform */
    uint64_t loads; /* the lanes that fetch their node from memory */
  };
  /* (scalar loads through the constant address space: the map is written at create() and by no kernel)
   * MASKS (3-D): every chained tile, with its mask of loading lanes; otherwise only the tiles whose mask is the lane at
   * the end of the wave, and no mask is read (host_layout.hpp). */
  template <int DIM>
  constexpr bool chain_masks_pay()
  {
    return DIM == 3;
  }
  template <bool MASKS>
  RYUJIN_DEV TileChain tile_chain(const DeviceMesh &M, const uint64_t colbase)
  {
    TileChain t{kChainNone, ~0ull};
    if constexpr (RYUJIN_CHAINED_GATHERS != 0) {
      if (M.tiles != nullptr) {
        typedef const uint32_t __attribute__((address_space(4))) *const_ptr;
        typedef const uint64_t __attribute__((address_space(4))) *const_ptr64;
        const uint32_t code = __builtin_amdgcn_readfirstlane(*(const_ptr)(uintptr_t)&M.tiles[colbase].chain);
        if constexpr (MASKS) {
          t.kind = code & kChainKindMask;
          if (t.kind != kChainNone) {
            const uint64_t m = *(const_ptr64)(uintptr_t)(M.chain_loads + colbase);
            t.loads = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) |
                      (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)m);
          }
        } else {
          t.kind = (code & kChainEndLaneOnly) != 0u ? (code & kChainKindMask) : kChainNone;
        }
      }
    }
    return t;
  }

  /* lane l takes the value of lane l + 1 (lane 63 keeps its own) / of lane l - 1 (lane 0 keeps its own) */
  RYUJIN_DEV double lane_next(const double x)
  {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x130 /* wave_shl:1 */, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false));
  }
  RYUJIN_DEV double lane_prev(const double x)
  {
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x138 /* wave_shr:1 */, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false));
  }
  /* whether this lane of a chained tile fetches its node from memory */
  template <bool MASKS>
  RYUJIN_DEV bool chain_lane_loads(const TileChain &t, const uint32_t lane)
  {
    if constexpr (MASKS)
      return ((t.loads >> lane) & 1ull) != 0ull;
    else
      return lane == (t.kind == kChainOwnPrev ? 0u : 63u);
  }

  RYUJIN_DEV uint32_t tile_transposed(const DeviceMesh &M, const TileDesc &t, const uint64_t colbase,
                                      const uint32_t lane)
  {
    if (t.delta != kTileIrregular) /* wave-uniform */
      return (lane < 64u - ((uint32_t)t.delta & 63u) ? t.ta : t.tb) + lane;
    /* (an irregular tile: the position comes from memory. It ARRIVES inside this branch: left pending, the wait moves
     * to where the two paths meet -- every tile's path -- and, behind a load the compiler cannot count, becomes
     * vmcnt(0): each column of step 3 then waited for the gather of the column before, scripts/isa_loop_waits.sh) */
    uint32_t pos = M.idx_t[colbase * 64 + lane];
#if (RYUJIN_DIAG_PINS & 1)
    asm volatile("" : "+v"(pos));
#endif
    return pos;
  }

  template <bool USE = true>
  RYUJIN_DEV uint32_t tile_transposed(const DeviceMesh &M, const uint64_t colbase, const uint32_t lane)
  {
    return tile_transposed(M, tile_desc<USE>(M, colbase), colbase, lane);
  }

  /* TRANSPOSED POSITIONS OF SEVERAL COLUMNS AT ONCE (steps 6, 7: the row's l_ji gathers). Every one of them is the end
   * of a chain -- tile descriptor (or the index array) -> position -> value -- and written column by column the
   * compiler emits the chain column by column: descriptor load, vmcnt(0), gather, vmcnt(0), eight (2-D) or 26 (3-D) times
   * in a row, sixteen to fifty-two round trips where the sweep needs two (scripts/isa_loop_waits.sh on the round-5
   * kernels; the wave-uniform branch on the descriptor and the ballots behind each gather keep it from batching them).
   * Here all descriptors / indices of a group of columns are loaded first, then all positions are formed; the callers
   * then issue all gathers of the group before they look at the first value. batch_fence() keeps the scheduler from
   * sinking the loads of a group back down to their uses. Measured (profiles/r06u_ab_batched_gathers_*.log): step 6
   * -12 %, step 7 -9 % on C2, -5 % / -2 % on the C4 share. Step 3 (k_dij_diag_unrolled) does NOT take it: its eight
   * chains already overlap as written (there is no ballot behind its gathers), and the two fenced groups cost it
   * 0.076 -> 0.126 ms on C2. */
  RYUJIN_DEV void batch_fence() { __builtin_amdgcn_sched_barrier(0); }

  template <bool USE, int N>
  RYUJIN_DEV void transposed_positions(const DeviceMesh &M, const RowCtx &r, const int c0, uint32_t (&tpos)[N])
  {
    if constexpr (USE) {
      if (M.tiles != nullptr) {
        int4 raw[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
          raw[k] = int4{kTileIrregular, 0, 0, 0};
          if ((uint32_t)(c0 + k) < r.width)
            raw[k] = load_tile_desc(M.tiles + ((uint64_t)r.base + c0 + k));
        }
        batch_fence();
        bool any_irregular = false;
#pragma unroll
        for (int k = 0; k < N; ++k) {
          tpos[k] = 0;
          if ((uint32_t)(c0 + k) < r.width) {
            const int32_t delta = __builtin_amdgcn_readfirstlane(raw[k].x);
            const uint32_t ta = (uint32_t)__builtin_amdgcn_readfirstlane(raw[k].y);
            const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane(raw[k].z);
            tpos[k] = (r.lane < 64u - ((uint32_t)delta & 63u) ? ta : tb) + r.lane;
            any_irregular = any_irregular || delta == kTileIrregular;
          }
        }
        if (any_irregular) { /* wave-uniform, rare (boundary slices): the explicit array for those columns */
#pragma unroll
          for (int k = 0; k < N; ++k)
            if ((uint32_t)(c0 + k) < r.width && __builtin_amdgcn_readfirstlane(raw[k].x) == kTileIrregular)
              tpos[k] = M.idx_t[((uint64_t)r.base + c0 + k) * 64 + r.lane];
        }
        /* (one wait for whatever the branch above loaded, here: otherwise the compiler waits -- for everything in
         * flight -- in front of every gather that uses one of the positions) */
#pragma unroll
        for (int k = 0; k < N; ++k)
          asm volatile("" : "+v"(tpos[k]));
        return;
      }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      tpos[k] = 0;
      if ((uint32_t)(c0 + k) < r.width)
        tpos[k] = M.idx_t[((uint64_t)r.base + c0 + k) * 64 + r.lane];
    }
#pragma unroll
    for (int k = 0; k < N; ++k)
      asm volatile("" : "+v"(tpos[k]));
  }

  /* ---- the column pipeline and gfx9's single memory counter. Vector loads AND stores count in one counter (vmcnt),
   * retired in issue order. A sweep prefetches the operands of column c + 1 while it works on column c; the compiler
   * places the wait for column c's operands at their first use -- and where the loop holds a memory operation it
   * cannot count on every path (the store of an active lane, the index load of an irregular tile), that wait is
   * vmcnt(0): EVERYTHING in flight. If the first use sits behind the prefetch of column c + 1 the wave then waits
   * for that prefetch as well and nothing overlaps (scripts/isa_loop_waits.sh shows the pattern: loads, then
   * vmcnt(0) straight away). arrived() is a use the compiler cannot see through: called on the operands of column c
   * BEFORE the loads of column c + 1 are issued it pins the wait in front of them. The stores of a column are kept
   * back and issued behind the next column's loads for the same reason (a store issued at the end of an iteration
   * is the first thing the next iteration waits for). ---- */
  RYUJIN_DEV void arrived(double &x) { asm volatile("" : "+v"(x)); }
  RYUJIN_DEV void arrived(uint32_t &x) { asm volatile("" : "+v"(x)); }
  template <int N>
  RYUJIN_DEV void arrived(double (&x)[N])
  {
#pragma unroll
    for (int q = 0; q < N; ++q)
      arrived(x[q]);
  }

  /* XCD-LOCAL BLOCK RANGES. Block b runs on XCD b % 8 (observed; speed only): consecutive blocks -- consecutive
   * slices, whose rows gather from the same lattice rows and planes -- land on eight different L2s, and every L2
   * ends up fetching every node's data. Renumbered in chunks of 8 C blocks, XCD x takes the C consecutive blocks
   * [x C, (x + 1) C) of a chunk: the rows one L2 serves at a time are a contiguous range. The tail of the launch
   * that does not fill a chunk keeps its numbering. */
  RYUJIN_DEV uint32_t mapped_block(const DeviceMesh &M)
  {
    uint32_t block = blockIdx.x;
    if (M.xcd_chunk > 0u) {
      const uint32_t span = 8u * M.xcd_chunk;
      if (block < gridDim.x - gridDim.x % span) {
        const uint32_t rem = block % span;
        block = block - rem + (rem & 7u) * M.xcd_chunk + (rem >> 3);
      }
    }
    return block;
  }

  RYUJIN_DEV RowCtx row_context(const DeviceMesh &M)
  {
    RowCtx r;
    r.lane = threadIdx.x & 63;
    const uint32_t block = mapped_block(M);
    /* slice, base and width are the same in all lanes of the wave: say so (scalar registers, scalar loop control) */
    uint32_t id = block * kWavesPerBlock + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    /* STACKED BLOCKS. On a structured patch the rows a slice gathers from sit one lattice row (2-D) / one lattice
     * plane (3-D) of the mesh above and below it -- band_stride slices away, served by whichever XCD's L2 happens to
     * run those slices: every node's data is fetched ~3 times per sweep (profiles/r05k_pmc.md: step 5 moves 1.28x, step
     * 2 1.9x its own bytes). With band_stride = G > 1 the four waves of a block take the slices s, s + G, s + 2G,
     * s + 3G -- four rows (planes) stacked on top of each other -- so that three of the four vertical neighbour
     * relations stay inside one workgroup, i.e. one L2: bands of 4G consecutive slices, block q of a band takes
     * {q, q + G, q + 2G, q + 3G}; the remainder of the range keeps consecutive slices. Every slice is still taken by
     * exactly one wave; nothing else depends on which one. */
    if (M.band_stride > 1u) {
      const uint32_t band = kWavesPerBlock * M.band_stride;
      const uint32_t n = M.slice_end - M.slice_begin;
      if (id < n - n % band) {
        const uint32_t rem = id % band;
        id = id - rem + rem / kWavesPerBlock + (rem % kWavesPerBlock) * M.band_stride;
      }
    }
    r.slice = __builtin_amdgcn_readfirstlane(M.slice_begin + id);
    r.valid = r.slice < M.slice_end;
    if (!r.valid) {
      r.row = r.len = r.base = r.width = 0;
      return r;
    }
    r.row = r.slice * 64 + r.lane;
    r.len = M.row_len[r.row];
    r.base = __builtin_amdgcn_readfirstlane(M.slice_off[r.slice]);
    r.width = __builtin_amdgcn_readfirstlane(M.slice_off[r.slice + 1]) - r.base;
    return r;
  }

  /* ------------------------------------------------------------------ step 1 */

  /* Boundary conditions (hyperbolic_module.template.h:102-146), folded into the first pre-pass sweep: the thread
   * of a boundary row applies the row's boundary_map entries in the reference's (serial) order before anything
   * else reads U_i. slice_mask[s] bit l <=> row 64 s + l is a boundary DoF; its entries are those of group
   * slice_first[s] + (number of boundary rows below it in the slice), groups being sorted by DoF. (A launch of
   * its own costs ~10 us of dependency latency per update -- a tenth of the update on the 43 k gridpoint mesh;
   * used below kBcFoldMaxSlices, k_apply_bc above.) */
  struct BcFold {
    const unsigned long long *slice_mask; /* NULL: no boundary DoFs */
    const uint32_t *slice_first;
    const uint32_t *grp_start;
    const double *b_normal;
    const uint8_t *b_id;
    const double *dirichlet;
  };

  /* the boundary_map entries of group g (DoF i), in the reference's serial order */
  template <typename E>
  RYUJIN_DEV void apply_bc_group(const typename E::Params &P, const BcFold &B, const uint32_t g, const uint32_t i,
                                 double *U)
  {
    constexpr int K = E::K;
    constexpr int DIM = E::DIMENSION;
    const uint32_t e0 = B.grp_start[g], e1 = B.grp_start[g + 1];
    double U_i[K];
    load_state<K>(U, i, U_i);
    bool touched = false;
    for (uint32_t e = e0; e < e1; ++e) {
      const int id = B.b_id[e];
      if (id == RYUJIN_BC_DO_NOTHING)
        continue;
      double normal[DIM], U_D[K], result[K];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        normal[d] = B.b_normal[(size_t)e * DIM + d];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_D[q] = B.dirichlet ? B.dirichlet[(size_t)e * K + q] : 0.;
      E::apply_boundary_conditions(P, id, U_i, normal, U_D, result);
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_i[q] = result[q];
      touched = true;
    }
    if (touched)
      store_state<K>(U, i, U_i);
  }

  /* folded form: called by the thread of row i at the top of the first pre-pass kernel */
  template <typename E>
  RYUJIN_DEV void apply_bc_row(const typename E::Params &P, const BcFold &B, const uint32_t i, double *U)
  {
    if (!B.slice_mask)
      return;
    const unsigned long long m = B.slice_mask[i >> 6];
    const uint32_t lane = i & 63u;
    if (!((m >> lane) & 1ull))
      return;
    apply_bc_group<E>(P, B, B.slice_first[i >> 6] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), i, U);
  }

  /* launch of its own (large meshes: the boundary-condition code would cost the streaming pre-pass kernel half
   * of its occupancy -- 108 instead of 40 registers for Euler -- and a ~5 us launch is invisible there): one
   * thread per boundary DoF */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_apply_bc(const typename E::Params P, const uint32_t n_groups, const uint32_t *__restrict__ b_i,
             const BcFold B, double *U)
  {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups)
      return;
    apply_bc_group<E>(P, B, g, b_i[B.grp_start[g]], U);
  }

  /* the same for a state vector whose precomputed values and Riemann records were left behind by the last sweep of
   * the step that produced it (FusedPrecompute, kernels_limiter.hpp): those of the boundary rows are redone from
   * the state the boundary conditions leave */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_apply_bc_records(const typename E::Params P, const uint32_t n_groups, const uint32_t *__restrict__ b_i,
                     const uint16_t *__restrict__ row_len, const BcFold B, double *U, double *__restrict__ prec,
                     double *__restrict__ rec)
  {
    constexpr int K = E::K, RS = E::RS;
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups)
      return;
    const uint32_t i = b_i[B.grp_start[g]];
    apply_bc_group<E>(P, B, g, i, U);
    if (row_len[i] == 1)
      return;
    double U_i[K], r[RS];
    load_state<K>(U, i, U_i);
    const double2 prec_i = E::precompute(P, U_i);
    reinterpret_cast<double2 *>(prec)[i] = prec_i;
    E::node_record(P, U_i, prec_i, r);
    double2 *out = reinterpret_cast<double2 *>(rec + (size_t)i * RS);
#pragma unroll
    for (int q = 0; q < RS / 2; ++q) {
      double2 t;
      t.x = r[2 * q];
      t.y = r[2 * q + 1];
      out[q] = t;
    }
  }

  /* precomputation_loop (source/euler/hyperbolic_system.h:702-737, shallow_water/hyperbolic_system.h:676-716):
   * the precomputed values AND the per-node Riemann record (E::riemann_record) in one pass over the owned rows */
  template <typename E, bool WITH_BC>
  __global__ void __launch_bounds__(kBlock)
  k_precompute_records(const typename E::Params P, const DeviceMesh M, const BcFold B, double *U,
                       double *__restrict__ prec, double *__restrict__ rec)
  {
    constexpr int K = E::K, RS = E::RS;
    const uint32_t i = M.slice_begin * 64 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M.n_owned || i >= M.slice_end * 64)
      return;
    if constexpr (WITH_BC)
      apply_bc_row<E>(P, B, i, U);
    if (M.row_len[i] == 1)
      return;
    double U_i[K], r[RS];
    load_state<K>(U, i, U_i);
    const double2 prec_i = E::precompute(P, U_i);
    reinterpret_cast<double2 *>(prec)[i] = prec_i;
    E::node_record(P, U_i, prec_i, r);
    double2 *out = reinterpret_cast<double2 *>(rec + (size_t)i * RS);
#pragma unroll
    for (int g = 0; g < RS / 2; ++g) {
      double2 t;
      t.x = r[2 * g];
      t.y = r[2 * g + 1];
      out[g] = t;
    }
  }

  /* the same pair for the ghost rows [first, last): computed locally from the exchanged ghost states (both
   * are functions of U_j alone, so the reference's update_ghost_values() on the precomputed vector,
   * hyperbolic_module.template.h:157-160, moves nothing that is not already here -- one exchange less per
   * update) */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_ghost_precompute_records(const typename E::Params P, const uint32_t first, const uint32_t last,
                             const double *__restrict__ U, double *__restrict__ prec,
                             double *__restrict__ rec)
  {
    constexpr int K = E::K, RS = E::RS;
    const uint32_t i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= last)
      return;
    double U_i[K], r[RS];
    load_state<K>(U, i, U_i);
    const double2 prec_i = E::precompute(P, U_i);
    reinterpret_cast<double2 *>(prec)[i] = prec_i;
    E::node_record(P, U_i, prec_i, r);
#pragma unroll
    for (int g = 0; g < RS; ++g)
      rec[(size_t)i * RS + g] = r[g];
  }

  /* a node record (E::node_record / E::riemann_record) as 16-byte loads */
  template <int RS>
  RYUJIN_DEV void load_record(const double *__restrict__ rec, const uint32_t i, double (&r)[RS])
  {
    const double2 *b = reinterpret_cast<const double2 *>(rec + (size_t)i * RS);
#pragma unroll
    for (int g = 0; g < RS / 2; ++g) {
      const double2 t = b[g];
      r[2 * g] = t.x;
      r[2 * g + 1] = t.y;
    }
  }

  /* ------------------------------------------------------------------ step 2 */

  template <typename E>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_DIJ)
  k_dij_alpha(const typename E::Params P, const DeviceMesh M, const double *__restrict__ U,
              const double *__restrict__ prec, double *__restrict__ dij, double *__restrict__ alpha)
  {
    constexpr int K = E::K;
    constexpr int DIM = E::DIMENSION;
    step_begin(M);
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double2 *__restrict__ prec2 = reinterpret_cast<const double2 *>(prec);

    double U_i[K];
    load_state<K>(U, i, U_i);
    typename E::Indicator indicator;
    indicator.reset(P, U_i, prec2[i]);

    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;

    for (uint32_t c = 0; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const bool active = row_active && c < r.len;
      double c_ij[DIM], U_j[K];
      const uint32_t j = ld_stream(cols + (pos));
      load_entry<DIM>(cij, colbase, r.lane, c_ij);
      load_state<K>(U, j, U_j);
      const double2 prec_j = prec2[j];

      if (active) {
        indicator.accumulate(P, U_j, prec_j, c_ij);
        /* upper triangle only (:394-408) */
        if (c > 0 && j > i)
          dij[pos] = E::dij_from_states(P, U_i, U_j, c_ij);
      }
    }

    if (row_active)
      alpha[i] = indicator.alpha(P, M.mi[i] * M.measure_of_omega_inverse);
  }

  /* Step 2 as two kernels (shallow water; Euler with the general Riemann path): the streaming indicator sweep and the compute-bound
   * Riemann sweep have very different register needs; split, the Riemann kernel only touches the
   * upper-triangle columns (known from the per-row bitmask, no loads for the others) and runs at a
   * higher occupancy. */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_alpha(const typename E::Params P, const DeviceMesh M, const double *__restrict__ U,
          const double *__restrict__ prec, double *__restrict__ alpha)
  {
    constexpr int K = E::K;
    constexpr int DIM = E::DIMENSION;
    step_begin(M);
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double2 *__restrict__ prec2 = reinterpret_cast<const double2 *>(prec);
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;

    double U_i[K];
    load_state<K>(U, i, U_i);
    typename E::Indicator indicator;
    indicator.reset(P, U_i, prec2[i]);

    uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
    uint32_t j_nn = r.width > 1 ? ld_stream(cols + (((uint64_t)r.base + 1) * 64 + r.lane)) : i;
    double c_n[DIM], U_n[K];
    load_entry<DIM>(cij, r.base, r.lane, c_n);
    load_state<K>(U, j_n, U_n);
    double2 prec_n = prec2[j_n];
    for (uint32_t c = 0; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      double c_ij[DIM], U_j[K];
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        c_ij[d] = c_n[d];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_j[q] = U_n[q];
      const double2 prec_j = prec_n;
      if (c + 1 < r.width) {
        j_n = j_nn;
        load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
        load_state<K>(U, j_n, U_n);
        prec_n = prec2[j_n];
        j_nn = (c + 2 < r.width) ? ld_stream(cols + ((colbase + 2) * 64 + r.lane)) : i;
      }
      if (row_active && c < r.len)
        indicator.accumulate(P, U_j, prec_j, c_ij);
    }
    if (row_active)
      alpha[i] = indicator.alpha(P, M.mi[i] * M.measure_of_omega_inverse);
  }

  /* The Riemann sweep on per-node records: no state loads, no per-pair pow (see euler_device.hpp) */
  template <typename E, bool GENERAL>
  __global__ void __launch_bounds__(kBlock, RYUJIN_OCC_DIJ)
  k_dij_records(const typename E::Params P, const DeviceMesh M, const uint32_t *__restrict__ lower_mask,
                const double *__restrict__ rec, double *__restrict__ dij)
  {
    constexpr int DIM = E::DIMENSION;
    constexpr int RS = E::RS;
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;
    const uint32_t upper =
        row_active ? (~lower_mask[r.row] & (r.len >= 32 ? 0xFFFFFFFFu : ((1u << r.len) - 1u)) & ~1u) : 0u;

    double rec_i[RS];
    {
      const double2 *b = reinterpret_cast<const double2 *>(rec + (size_t)i * RS);
#pragma unroll
      for (int g = 0; g < RS / 2; ++g) {
        const double2 t = b[g];
        rec_i[2 * g] = t.x;
        rec_i[2 * g + 1] = t.y;
      }
    }
    for (uint32_t c = 1; c < r.width; ++c) {
      const bool mine = (upper >> c) & 1u;
      if (!__any(mine))
        continue; /* wave-uniform: no loads at all for lower-triangle columns */
      const uint64_t colbase = (uint64_t)r.base + c;
      const uint64_t pos = colbase * 64 + r.lane;
      const uint32_t j = ld_stream(cols + (pos));
      double c_ij[DIM], rec_j[RS];
      load_entry<DIM>(cij, colbase, r.lane, c_ij);
      const double2 *b = reinterpret_cast<const double2 *>(rec + (size_t)j * RS);
#pragma unroll
      for (int g = 0; g < RS / 2; ++g) {
        const double2 t = b[g];
        rec_j[2 * g] = t.x;
        rec_j[2 * g + 1] = t.y;
      }
      if (mine)
        dij[pos] = E::template dij_from_records<GENERAL>(P, rec_i, rec_j, c_ij);
    }
  }

  /* Step 2 in ONE kernel on top of the node records: the indicator sweep streams the stencil at HBM speed and
   * leaves the VALU mostly idle, the Riemann sweep on records is short and needs few registers -- fused, its
   * arithmetic hides behind the indicator's loads and the column indices / c_ij are read once instead of twice
   * (the round-1 fusion lost because the old Riemann solver held 200+ registers). */
  template <typename E, bool GENERAL>
  __global__ void __launch_bounds__(kBlock, E::kRecordHoldsState ? RYUJIN_OCC_DIJ_NODE_RECORD : RYUJIN_OCC_DIJ)
  k_dij_alpha_records(const typename E::Params P, const DeviceMesh M, const double *__restrict__ U,
                      const double *__restrict__ prec, const double *__restrict__ rec,
                      double *__restrict__ dij, double *__restrict__ alpha)
  {
    constexpr int K = E::K, RS = E::RS, DIM = E::DIMENSION;
    step_begin(M);
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const double2 *__restrict__ prec2 = reinterpret_cast<const double2 *>(prec);
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;

    double rec_i[RS];
    load_record<RS>(rec, i, rec_i);

    if constexpr (E::kRecordHoldsState) {
      /* Euler: the record is all the indicator needs of a node as well (E::node_record) -- one gather of RS doubles
       * per neighbour, nothing from U and prec */
      typename E::Indicator indicator;
      indicator.reset_record(P, rec_i);

      uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
      uint32_t j_nn = r.width > 1 ? ld_stream(cols + (((uint64_t)r.base + 1) * 64 + r.lane)) : i;
      double c_n[DIM], rec_n[RS];
      load_entry<DIM>(cij, r.base, r.lane, c_n);
      if constexpr (RYUJIN_DIJ_PREFETCH_RECORD)
        load_record<RS>(rec, j_n, rec_n);
      double d_pending = 0.; /* the d_ij of the column before, stored behind this column's loads (see arrived()) */
      bool d_pending_on = false;
      for (uint32_t c = 0; c < r.width; ++c) {
        const uint64_t colbase = (uint64_t)r.base + c;
        uint32_t j = j_n;
        double c_ij[DIM], rec_j[RS];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          c_ij[d] = c_n[d];
        if constexpr (RYUJIN_DIJ_PREFETCH_RECORD) {
#pragma unroll
          for (int q = 0; q < RS; ++q)
            rec_j[q] = rec_n[q];
        } else {
          load_record<RS>(rec, j, rec_j);
        }
        if constexpr (RYUJIN_PIN_WAITS) {
          arrived(c_ij);
          arrived(rec_j);
          arrived(j_nn);
        }
        if (c + 1 < r.width) {
          j_n = j_nn;
          load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
          if constexpr (RYUJIN_DIJ_PREFETCH_RECORD)
            load_record<RS>(rec, j_n, rec_n);
          j_nn = (c + 2 < r.width) ? ld_stream(cols + ((colbase + 2) * 64 + r.lane)) : i;
        }
        if (d_pending_on)
          dij[(colbase - 1) * 64 + r.lane] = d_pending;
        const bool active = row_active && c < r.len;
        /* column 0 is the row itself: eta_j / rho_j = eta_i / rho_i and f_j = f_i bit for bit, its terms are exact
         * zeros added to sums that start at zero */
        if (active && (c > 0 || !E::kIndicatorDiagonalIsZero))
          indicator.accumulate_record(rec_j, c_ij);
        /* upper triangle only (:394-408) */
        d_pending_on = active && c > 0 && j > i;
        if (d_pending_on)
          d_pending = E::template dij_from_records<GENERAL>(P, rec_i, rec_j, c_ij);
      }
      if (d_pending_on)
        dij[((uint64_t)r.base + r.width - 1) * 64 + r.lane] = d_pending;
      if (row_active)
        alpha[i] = indicator.alpha(P, M.mi[i] * M.measure_of_omega_inverse);
      return;
    } else {
      double U_i[K];
      load_state<K>(U, i, U_i);
      typename E::Indicator indicator;
      indicator.reset(P, U_i, prec2[i]);

      uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
      uint32_t j_nn = r.width > 1 ? ld_stream(cols + (((uint64_t)r.base + 1) * 64 + r.lane)) : i;
      double c_n[DIM], U_n[K];
      load_entry<DIM>(cij, r.base, r.lane, c_n);
      load_state<K>(U, j_n, U_n);
      double2 prec_n = prec2[j_n];
      /* (1-D, 2-D, shallow water: 124 registers and four waves per SIMD. Pinning the waits -- arrived() -- and keeping
       * the d_ij store back costs 16 registers and the fourth wave here: 0.241 -> 0.247 ms on C2, with the Riemann
       * record prefetched as well 0.268, profiles/r06l_ab_pinned_waits_c2.log; the loop stays as it was) */
      for (uint32_t c = 0; c < r.width; ++c) {
        const uint64_t colbase = (uint64_t)r.base + c;
        const uint32_t j = j_n;
        double c_ij[DIM], U_j[K];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          c_ij[d] = c_n[d];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_j[q] = U_n[q];
        const double2 prec_j = prec_n;
        if (c + 1 < r.width) {
          j_n = j_nn;
          load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
          load_state<K>(U, j_n, U_n);
          prec_n = prec2[j_n];
          j_nn = (c + 2 < r.width) ? ld_stream(cols + ((colbase + 2) * 64 + r.lane)) : i;
        }
        const bool active = row_active && c < r.len;
        if (active && (c > 0 || !E::kIndicatorDiagonalIsZero))
          indicator.accumulate(P, U_j, prec_j, c_ij);
        /* upper triangle only (:394-408) */
        const bool mine = active && c > 0 && j > i;
        if (__any(mine)) {
          double rec_j[RS];
          load_record<RS>(rec, j, rec_j);
          if (mine)
            dij[colbase * 64 + r.lane] = E::template dij_from_records<GENERAL>(P, rec_i, rec_j, c_ij);
        }
      }
      if (row_active)
        alpha[i] = indicator.alpha(P, M.mi[i] * M.measure_of_omega_inverse);
    }
  }

  /* ------------------------------------------------------------------ step 3 */

  /* boundary pairs (:462-490): d_ij = max(d_ij, |c_ji| lambda_max(U_j, U_i, n_ji)) for j >= i */
  template <typename E>
  __global__ void __launch_bounds__(kBlock)
  k_dij_boundary(const typename E::Params P, const uint32_t n_pairs, const uint32_t *__restrict__ p_i,
                 const uint32_t *__restrict__ p_j, const uint32_t *__restrict__ p_pos,
                 const double *__restrict__ cji,
                 const double *__restrict__ U, double *__restrict__ dij)
  {
    constexpr int K = E::K;
    constexpr int DIM = E::DIMENSION;
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_pairs)
      return;
    const uint32_t i = p_i[q], j = p_j[q];
    if (j < i)
      return;
    double U_i[K], U_j[K], c_ji[DIM];
    load_state<K>(U, i, U_i);
    load_state<K>(U, j, U_j);
#pragma unroll
    for (int d = 0; d < DIM; ++d)
      c_ji[d] = cji[(size_t)q * DIM + d];
    const double d_ji = E::dij_from_states(P, U_j, U_i, c_ji);
    const uint32_t pos = p_pos[q];
    dij[pos] = fmax(dij[pos], d_ji);
  }

  /* symmetrise, diagonal, tau_max (:494-560) */
  __global__ void __launch_bounds__(kBlock)
  k_dij_diag(const DeviceMesh M, const double cfl, double *__restrict__ dij,
             DeviceScalars *__restrict__ scalars)
  {
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = r.row;
    double d_sum = 0.;
    for (uint32_t c = 1; c < r.width; ++c) {
      const uint64_t pos = ((uint64_t)r.base + c) * 64 + r.lane;
      if (row_active && c < r.len) {
        const uint32_t j = ld_stream(M.cols + (pos));
        double d;
        if (j < i) {
          d = dij[M.idx_t[pos]];
          dij[pos] = d;
        } else {
          d = dij[pos];
        }
        d_sum -= d;
      }
    }
    double tau = DBL_MAX;
    if (row_active) {
      d_sum = fmin(d_sum, -1.e6 * DBL_MIN);
      dij[(uint64_t)r.base * 64 + r.lane] = d_sum;
      tau = cfl * M.mi[i] / (-2. * d_sum);
    }
    publish_tau_min(scalars, wave_min(tau), r.lane);
  }

  /* Same sweep with memory-level parallelism: the row's "lower triangle" columns are known at setup
   * (bit c of lower_mask[row] <=> cols(row,c) < row), so neither the column indices nor a dependent
   * load chain are needed: all d_ij / idx_t loads of a row are issued back to back (the generic
   * kernel above spends 98 % of its wave cycles waiting on 3 dependent loads per column). */
  template <int MAXW>
  __global__ void __launch_bounds__(kBlock)
  k_dij_diag_unrolled(const DeviceMesh M, const uint32_t *__restrict__ lower_mask, const double cfl,
                      double *__restrict__ dij, DeviceScalars *__restrict__ scalars)
  {
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    uint32_t mask = row_active ? lower_mask[r.row] : 0u;
#if (RYUJIN_DIAG_PINS & 1)
    asm volatile("" : "+v"(mask)); /* (arrived: the per-column blocks below must not each wait for it, and with it for the gather of the column before) */
#endif
    double d[MAXW];
#pragma unroll
    for (int c = 1; c < MAXW; ++c) {
      d[c] = 0.;
      if ((uint32_t)c < r.width) {
        const uint32_t pos = (r.base + c) * 64 + r.lane;
        const TileDesc t = tile_desc<(MAXW <= 9)>(M, (uint64_t)r.base + c);
        const uint32_t src = ((mask >> c) & 1u) ? tile_transposed(M, t, (uint64_t)r.base + c, r.lane) : pos;
        d[c] = dij[src];
      }
    }
    /* ONE wait for the row's values: left to the compiler, every column of the loop below waits for the one it uses --
     * behind conditional stores it cannot count, with vmcnt(0): for the store of the column before as well */
#if (RYUJIN_DIAG_PINS & 2)
#pragma unroll
    for (int c = 1; c < MAXW; ++c)
      asm volatile("" : "+v"(d[c]));
#endif
    double d_sum = 0.;
#pragma unroll
    for (int c = 1; c < MAXW; ++c) {
      if ((uint32_t)c < r.width) {
        if ((mask >> c) & 1u)
          dij[(r.base + c) * 64 + r.lane] = d[c];
        if (row_active && (uint32_t)c < r.len)
          d_sum -= d[c];
      }
    }
    double tau = DBL_MAX;
    if (row_active) {
      d_sum = fmin(d_sum, -1.e6 * DBL_MIN);
      dij[(uint64_t)r.base * 64 + r.lane] = d_sum;
      tau = cfl * M.mi[r.row] / (-2. * d_sum);
    }
    publish_tau_min(scalars, wave_min(tau), r.lane);
  }

  /* tau = (tau_in == 0 ? tau_max : tau_in) with the validity check of :571-578, evaluated by EVERY thread of
   * the step-4 kernel from the finished (and, on several ranks, all-reduced) tau_max -- the kernel boundary
   * behind step 3 orders it. The first thread of the launch also publishes tau / the validity flags for the
   * later sweeps and the host (idempotent when a sweep runs as an export and an interior launch).
   * use_device_tau: later stages of a device-resident RK step reuse the tau of the first stage. */
  RYUJIN_DEV double finalize_tau(DeviceScalars *scalars)
  {
    const double tau_max = __longlong_as_double((long long)scalars->tau_max_bits);
    const int use_device_tau = scalars->use_device_tau;
    const double tau_in = scalars->tau_in;
    const double tau = use_device_tau ? scalars->tau_rk : (tau_in == 0. ? tau_max : tau_in);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const int invalid = (isnan(tau_max) || isinf(tau_max) || !(tau_max > 0.)) ? 1 : 0;
      const int stage = scalars->stage;
      scalars->tau_invalid = invalid;
      if (invalid && scalars->tau_invalid_accum < kStageCode - stage)
        scalars->tau_invalid_accum = kStageCode - stage;
      scalars->tau = tau;
      if (!use_device_tau)
        scalars->tau_rk = tau;
    }
    return tau;
  }

  /* end of a device-resident RK step: fold the restart flag of the last stage into the accumulator */
  __global__ void k_accumulate_flags(const int stage, DeviceScalars *__restrict__ scalars)
  {
    if (scalars->restart_needed && scalars->restart_accum < kStageCode - stage)
      scalars->restart_accum = kStageCode - stage;
  }

  /* ------------------------------------------------------------------ step 4 */

  template <int DIM>
  struct StageArgs {
    int stages;
    const double *U[4];
    const double *prec[4]; /* precomputed block of the stage vectors (shallow-water sources) */
    double w[4];
  };

  /* STORE_P = false: the first part of P_ij is not written here but recomputed -- with the identical
   * operation sequence, hence bit-identical -- by k_pij_lij_recompute (saves the 8kS B/row store of
   * this sweep and the 8kS B/row load of step 5 for 8dS+8S B/row of c_ij, d_ij loads there). */
  template <int DIM, bool HAS_STAGES, bool STORE_P = true, bool DG = false>
  __global__ void __launch_bounds__(kBlock, (DIM == 3 && (HAS_STAGES || RYUJIN_OCC_LOW_3D_ALL) && RYUJIN_OCC_LOW_3D_STAGES) ? 1 : ((DIM == 3 || HAS_STAGES || DG) ? 2 : RYUJIN_OCC_LOW))
  k_low_order(const EulerParams P, const DeviceMesh M, DeviceScalars *scalars,
              const double weight, const StageArgs<DIM> S, const double *__restrict__ U,
              const double *__restrict__ prec, const double *__restrict__ alpha,
              const double *__restrict__ dij, double *__restrict__ new_U, double *__restrict__ r_out,
              double *__restrict__ bounds, double *__restrict__ pij)
  {
    using E = Euler<DIM>;
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
    double f_i[K][DIM];
    E::flux(P, U_i, f_i);
    /* RYUJIN_LOW_PARK: the row's flux f(U_i) -- K x DIM doubles that are only read, once per column -- lives in LDS
     * across the column loop, [entry][lane] (conflict free): 16 registers (2-D) that separate the sweep from 3 waves
     * per SIMD. The same operands into the same operations: the same bits. */
    constexpr bool kParkFlux = RYUJIN_LOW_PARK && !HAS_STAGES && !DG && DIM >= 2;
    __shared__ double parked_flux[kParkFlux ? kWavesPerBlock * K * DIM * 64 : 1];
    double *const parked = parked_flux + (kParkFlux ? (threadIdx.x >> 6) * K * DIM * 64 : 0);
    if constexpr (kParkFlux) {
#pragma unroll
      for (int q = 0; q < K; ++q)
#pragma unroll
        for (int d = 0; d < DIM; ++d)
          parked[(q * DIM + d) * 64 + r.lane] = f_i[q][d];
    }

    /* Limiter::reset (limiter.h:255-276) */
    double rho_min = DBL_MAX, rho_max = 0., s_min = DBL_MAX;
    double rho_relaxation_numerator = 0., rho_relaxation_denominator = 0., s_interp_max = 0.;

    /* software pipeline (see k_dij_alpha) */
    const uint32_t *__restrict__ cols = M.cols;
    const double *__restrict__ cij = M.cij;
    uint32_t j_n = ld_stream(cols + ((uint64_t)r.base * 64 + r.lane));
    uint32_t j_nn = r.width > 1 ? ld_stream(cols + (((uint64_t)r.base + 1) * 64 + r.lane)) : i;
    double c_n[DIM], U_n[K];
    load_entry<DIM>(cij, r.base, r.lane, c_n);
    double d_n = dij[(uint64_t)r.base * 64 + r.lane];
    load_state<K>(U, j_n, U_n);
    double alpha_n = alpha[j_n];
    double s_n = prec[(size_t)j_n * 2 + 0];

    for (uint32_t c = 0; c < r.width; ++c) {
      const uint64_t colbase = (uint64_t)r.base + c;
      const bool active = row_active && c < r.len;
      double c_ij[DIM], U_j[K];
      const uint32_t j = j_n;
#pragma unroll
      for (int d = 0; d < DIM; ++d)
        c_ij[d] = c_n[d];
#pragma unroll
      for (int q = 0; q < K; ++q)
        U_j[q] = U_n[q];
      const double d_ij = d_n, alpha_j = alpha_n, s_j = s_n;
      if (c + 1 < r.width) {
        j_n = j_nn;
        load_entry<DIM>(cij, colbase + 1, r.lane, c_n);
        d_n = dij[(colbase + 1) * 64 + r.lane];
        load_state<K>(U, j_n, U_n);
        alpha_n = alpha[j_n];
        s_n = prec[(size_t)j_n * 2 + 0];
        j_nn = (c + 2 < r.width) ? ld_stream(cols + ((colbase + 2) * 64 + r.lane)) : i;
      }

      if (!active)
        continue;

      double factor = (alpha_i + alpha_j) * .5;
      if constexpr (DG) /* hyperbolic_module.template.h:733-737 */
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
      E::flux(P, U_j, f_j);
      double flux_ij[K];
      if constexpr (kParkFlux) {
        /* flux_divergence() with f_i read from LDS (an opaque lane offset keeps the loop-invariant loads from being
         * hoisted back into registers) */
        uint32_t off = r.lane;
        asm volatile("" : "+v"(off));
        const double *const f_i_parked = parked + off;
#pragma unroll
        for (int q = 0; q < K; ++q) {
          double s = (f_i_parked[(q * DIM + 0) * 64] + f_j[q][0]) * c_ij[0];
#pragma unroll
          for (int d = 1; d < DIM; ++d)
            s += (f_i_parked[(q * DIM + d) * 64] + f_j[q][d]) * c_ij[d];
          flux_ij[q] = -s;
        }
      } else {
        E::flux_divergence(f_i, f_j, c_ij, flux_ij);
      }

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

      /* Limiter::accumulate (limiter.h:279-327) */
      {
        const double rho_i = U_i[0], rho_j = U_j[0];
        double dm_c = (U_i[1] - U_j[1]) * scaled_c_ij[0];
#pragma unroll
        for (int d = 1; d < DIM; ++d)
          dm_c += (U_i[1 + d] - U_j[1 + d]) * scaled_c_ij[d];
        const double rho_ij_bar = 0.5 * (rho_i + rho_j + dm_c) + 0.;
        rho_min = fmin(rho_min, rho_ij_bar);
        rho_max = fmax(rho_max, rho_ij_bar);
        s_min = fmin(s_min, s_j);
        rho_relaxation_numerator += 1. * (rho_i + rho_j);
        rho_relaxation_denominator += 1.;
        double U_avg[K];
#pragma unroll
        for (int q = 0; q < K; ++q)
          U_avg[q] = (U_i[q] + U_j[q]) * .5;
        /* column 0 (j = i, wave-uniform): U_avg = U_i exactly and s(U_i) is the precomputed s_i = s_j (the same
         * function on the same arguments): one power and one division less per row, the same bits */
        const double s_interp = c == 0 ? s_j : E::specific_entropy(P, U_avg);
        s_interp_max = fmax(s_interp_max, s_interp);
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
          E::flux(P, U_iHs, f_iHs);
          E::flux(P, U_jHs, f_jHs);
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

    /* Limiter::bounds (limiter.h:330-363) */
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
    const double rho_max_r = fmin((1. + r_i) * rho_max, rho_max + relaxation);
    const double entropy_relaxation = P.lim_relaxation_factor * (s_interp_max - s_min);
    const double s_min_r = fmax((1. - r_i) * s_min, s_min - entropy_relaxation);

    /* bounds stored SoA: [NB][rows_padded] */
    const size_t stride = M.bounds_stride;
    bounds[i] = rho_min_r;
    bounds[stride + i] = rho_max_r;
    bounds[2 * stride + i] = s_min_r;
  }

  /* The same for a Description whose Limiter::combine_bounds is a plain component-wise minimum / maximum: bit q of
   * MAX_MASK <=> bound q is a maximum. EulerAEOS (rho_min, rho_max, s_min, gamma_min: euler_aeos/limiter.h:435-445)
   * NB = 4, mask 0b0010; scalar conservation (u_min, u_max: scalar_conservation/limiter.h:302-309) NB = 2, mask 0b10. */
  template <int NB, unsigned MAX_MASK>
  __global__ void __launch_bounds__(kBlock)
  k_bounds_combine_minmax(const DeviceMesh M, const double *__restrict__ in, double *__restrict__ out)
  {
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const size_t stride = M.bounds_stride;
    double b[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q)
      b[q] = in[(size_t)q * stride + i];
    for (uint32_t c = 1; c < r.width; ++c) {
      const uint32_t j = M.cols[((uint64_t)r.base + c) * 64 + r.lane];
      if (row_active && c < r.len) {
#pragma unroll
        for (int q = 0; q < NB; ++q) {
          const double v = in[(size_t)q * stride + j];
          b[q] = ((MAX_MASK >> q) & 1u) ? fmax(b[q], v) : fmin(b[q], v);
        }
      }
    }
    if (row_active) {
#pragma unroll
      for (int q = 0; q < NB; ++q)
        out[(size_t)q * stride + i] = b[q];
    }
  }

  /* Discontinuous ansatz: extend the limiter bounds over the stencil (hyperbolic_module.template.h:938-948
   * with Limiter::combine_bounds, euler/limiter.h:366-377: min, max, min). Reads the ORIGINAL bounds of
   * the neighbours and writes a second buffer (the reference combines in place with a benign race). */
  __global__ void __launch_bounds__(kBlock)
  k_bounds_combine_euler(const DeviceMesh M, const double *__restrict__ in, double *__restrict__ out)
  {
    const RowCtx r = row_context(M);
    if (!r.valid)
      return;
    const bool row_active = r.len > 1;
    const uint32_t i = row_active ? r.row : (r.row < M.n_owned ? r.row : M.n_owned - 1);
    const size_t stride = M.bounds_stride;
    double rho_min = in[i], rho_max = in[stride + i], s_min = in[2 * stride + i];
    for (uint32_t c = 1; c < r.width; ++c) {
      const uint32_t j = M.cols[((uint64_t)r.base + c) * 64 + r.lane];
      if (row_active && c < r.len) {
        rho_min = fmin(rho_min, in[j]);
        rho_max = fmax(rho_max, in[stride + j]);
        s_min = fmin(s_min, in[2 * stride + j]);
      }
    }
    if (row_active) {
      out[i] = rho_min;
      out[stride + i] = rho_max;
      out[2 * stride + i] = s_min;
    }
  }

  /* steps 5, 6, 7: kernels_limiter.hpp */

  /* ------------------------------------------------------------------ helpers */

  /* sadd: dst = s*dst + b*src over the whole local vector (time_integrator.template.h:18-25) */
  __global__ void __launch_bounds__(kBlock)
  k_sadd(const size_t n, const double s, const double b, double *__restrict__ dst,
         const double *__restrict__ src)
  {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n / 2; q += stride) {
      double2 d = reinterpret_cast<double2 *>(dst)[q];
      const double2 v = reinterpret_cast<const double2 *>(src)[q];
      d.x = s * d.x + b * v.x;
      d.y = s * d.y + b * v.y;
      reinterpret_cast<double2 *>(dst)[q] = d;
    }
  }

  /* Quantities-style conservation monitor (source/quantities.template.h: interior "mass" integrals):
   * partial[b][q] = sum over the rows of block b of m_i U_i[q]; fixed summation tree (wave shuffles,
   * then the 4 waves in order), second pass adds the blocks in order: bitwise reproducible. */
  template <int K>
  __global__ void __launch_bounds__(kBlock)
  k_integrals_partial(const uint32_t n_owned, const double *__restrict__ mi, const double *__restrict__ U,
                      double *__restrict__ partial)
  {
    __shared__ double lds[kWavesPerBlock][K];
    double acc[K];
#pragma unroll
    for (int q = 0; q < K; ++q)
      acc[q] = 0.;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_owned; i += gridDim.x * blockDim.x) {
      double U_i[K];
      load_state<K>(U, i, U_i);
      const double m = mi[i];
#pragma unroll
      for (int q = 0; q < K; ++q)
        acc[q] += m * U_i[q];
    }
#pragma unroll
    for (int q = 0; q < K; ++q) {
      double v = acc[q];
      for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
      if ((threadIdx.x & 63) == 0)
        lds[threadIdx.x >> 6][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < K) {
      double v = 0.;
      for (int w = 0; w < kWavesPerBlock; ++w)
        v += lds[w][threadIdx.x];
      partial[(size_t)blockIdx.x * K + threadIdx.x] = v;
    }
  }

  template <int K>
  __global__ void k_integrals_final(const uint32_t n_blocks, const double *__restrict__ partial,
                                    double *__restrict__ out)
  {
    if (threadIdx.x < K) {
      double v = 0.;
      for (uint32_t b = 0; b < n_blocks; ++b)
        v += partial[(size_t)b * K + threadIdx.x];
      out[threadIdx.x] = v;
    }
  }

  __global__ void __launch_bounds__(kBlock)
  k_debug_pow(const size_t n, const double *__restrict__ x, const double *__restrict__ y,
              double *__restrict__ out)
  {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n)
      out[q] = dev_pow(x[q], y[q]);
  }

  /* in-process transport (test facility), all-reduce on device scalars: every rank copies its value(s)
   * into its slot, then -- behind the events of all ranks -- reduces all slots. op 0: min over positive
   * doubles (compared through their bit patterns, as tau_max_bits), op 1: max over `count` <= 2 ints. */
  __global__ void k_slot_write(const void *__restrict__ value, const int op, const int count,
                               unsigned long long *__restrict__ slot)
  {
    if (op == 0) {
      slot[0] = *static_cast<const unsigned long long *>(value);
    } else {
      for (int q = 0; q < count; ++q)
        slot[q] = (unsigned long long)(long long)static_cast<const int *>(value)[q];
    }
  }

  __global__ void k_slot_reduce(const unsigned long long *__restrict__ slots, const int n_ranks, const int op,
                                const int count, void *__restrict__ value)
  {
    if (op == 0) {
      unsigned long long m = slots[0];
      for (int r = 1; r < n_ranks; ++r)
        m = slots[2 * r] < m ? slots[2 * r] : m;
      *static_cast<unsigned long long *>(value) = m;
    } else {
      for (int q = 0; q < count; ++q) {
        long long m = (long long)slots[q];
        for (int r = 1; r < n_ranks; ++r)
          m = (long long)slots[2 * r + q] > m ? (long long)slots[2 * r + q] : m;
        static_cast<int *>(value)[q] = (int)m;
      }
    }
  }

  /* pack owned entries of an n_comp-strided AoS vector into a contiguous send buffer */
  __global__ void __launch_bounds__(kBlock)
  k_pack_vector(const uint32_t n, const uint32_t *__restrict__ idx, const int stride,
                const double *__restrict__ v, double *__restrict__ out)
  {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n * (uint32_t)stride)
      return;
    const uint32_t e = q / stride, d = q % stride;
    out[q] = v[(size_t)idx[e] * stride + d];
  }

  __global__ void __launch_bounds__(kBlock)
  k_pack_matrix(const uint32_t n, const uint32_t *__restrict__ pos, const double *__restrict__ m,
                double *__restrict__ out)
  {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n)
      out[q] = m[pos[q]];
  }
} // namespace ryujin_hip

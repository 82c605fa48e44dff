/*
 * ryujin_hip.h -- C ABI of the MI355X-native hyperbolic update.
 *
 * This is the drop-in boundary for ONE hot path of conservation-laws/ryujin:
 * HyperbolicModule::prepare_state_vector() + HyperbolicModule::step<stages>()
 * (reference: source/hyperbolic_module.h:110-278,
 *  source/hyperbolic_module.template.h:96-193,234-1211).
 *
 * The reference has no FFI layer; its boundary is the C++ class template
 * ryujin::HyperbolicModule<Description,dim,Number>. A maintainer binds this
 * header from a thin C++ shim with the same member signatures (shown in
 * INTEGRATION.md, shipped as ryujin_amd/csrc/hyperbolic_module_shim.hpp).
 *
 * Conventions
 *  - plain C, no torch / HIP types in any signature;
 *  - every function returns an int status (never throws across the ABI);
 *  - all arrays are HOST pointers in the reference's own layouts (cited per
 *    field); the library converts them once to its device layout in create();
 *  - state vectors live in device memory behind integer handles; the caller
 *    moves data explicitly with state_upload()/state_download().
 */
#ifndef RYUJIN_HIP_H
#define RYUJIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
#define RYUJIN_OK 0
/* invariant-domain violation with id_violation_strategy == warn
 * (hyperbolic_module.template.h:1198-1202: n_warnings_++) */
#define RYUJIN_WARN 1
/* ... with id_violation_strategy == raise_exception; the C++ shim converts
 * this into `throw Restart()` (hyperbolic_module.template.h:1203-1205) */
#define RYUJIN_RESTART 2
/* tau_max is NaN/inf/<=0 (AssertThrow at hyperbolic_module.template.h:573) */
#define RYUJIN_ERR_TAU (-1)
#define RYUJIN_ERR_ARG (-2)
#define RYUJIN_ERR_HIP (-3)
#define RYUJIN_ERR_COMM (-4)
#define RYUJIN_ERR_UNSUPPORTED (-5)

/* ---- enums -------------------------------------------------------------- */
/* Description (source/<eq>/description.h) */
enum {
  RYUJIN_EQ_EULER = 0,
  RYUJIN_EQ_SHALLOW_WATER = 1,
  RYUJIN_EQ_EULER_AEOS = 2,
  RYUJIN_EQ_SCALAR_CONSERVATION = 3
};

/* FluxLibrary (source/scalar_conservation/flux_library.h). "function" takes a muparser expression in
 * the reference; here its polynomial subset sum_n c_n u^n per direction (covers "u", "0.5*u*u", ...),
 * with the same central-difference gradient (flux_function.h:80-84). */
enum { RYUJIN_FLUX_BURGERS = 0, RYUJIN_FLUX_KPP = 1, RYUJIN_FLUX_POLYNOMIAL = 2 };

/* EquationOfStateLibrary (source/euler_aeos/equation_of_state_library.h): the closed-form members.
 * "sesame" (tabulated, needs EOSPAC) and "function" (muparser) are host-library bound and not offered. */
enum {
  RYUJIN_EOS_POLYTROPIC_GAS = 0,             /* equation_of_state_polytropic_gas.h */
  RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS = 1,   /* equation_of_state_noble_abel_stiffened_gas.h */
  RYUJIN_EOS_VAN_DER_WAALS = 2,              /* equation_of_state_van_der_waals.h */
  RYUJIN_EOS_JONES_WILKINS_LEE = 3           /* equation_of_state_jones_wilkins_lee.h */
};

/* ryujin::Boundary (source/discretization.h:28-112) */
enum {
  RYUJIN_BC_DO_NOTHING = 0,
  RYUJIN_BC_PERIODIC = 1,
  RYUJIN_BC_SLIP = 2,
  RYUJIN_BC_NO_SLIP = 3,
  RYUJIN_BC_DIRICHLET = 4,
  RYUJIN_BC_DYNAMIC = 5,
  RYUJIN_BC_DIRICHLET_MOMENTUM = 6
};

/* IDViolationStrategy (source/hyperbolic_module.h:32-47) */
enum { RYUJIN_IDV_WARN = 0, RYUJIN_IDV_RAISE_EXCEPTION = 1 };

/* ---- run-time parameters ------------------------------------------------ */
/*
 * Everything the reference keeps in ParameterAcceptor subsections that the
 * hot path reads. Defaults (ryujin_hip_default_params) are the reference's.
 */
typedef struct ryujin_hip_params {
  int equation; /* RYUJIN_EQ_* */
  int dim;      /* 1, 2, 3 */

  /* "B - Equation" Euler: source/euler/hyperbolic_system.h:665-699 */
  double gamma;                          /* 7/5 */
  double reference_density;              /* 1 */
  double vacuum_state_relaxation_small;  /* 1e2 */
  double vacuum_state_relaxation_large;  /* 1e4 */

  /* "B - Equation" shallow water: source/shallow_water/hyperbolic_system.h:643-673 */
  double gravity;                        /* 9.81 */
  double manning_friction_coefficient;   /* 0 */
  double reference_water_depth;          /* 1 */
  double dry_state_relaxation_factor;    /* 0.2 */
  double dry_state_relaxation_small;     /* 1e2 */
  double dry_state_relaxation_large;     /* 1e4 */

  /* HyperbolicModule: hyperbolic_module.template.h:36,45 */
  double cfl;                /* 0.2 until the TimeIntegrator sets it */
  int id_violation_strategy; /* RYUJIN_IDV_WARN */

  /* "/indicator": source/euler/indicator.h:28 */
  double indicator_evc_factor; /* 1 */

  /* "/limiter": source/euler/limiter.h:26-49, shallow_water/limiter.h:27-59 */
  int limiter_iterations;             /* 2, must be in [0,2] */
  double limiter_newton_tolerance;    /* 1e-10 */
  int limiter_newton_max_iterations;  /* 2 */
  double limiter_relaxation_factor;   /* 1 */
  int limiter_limit_on_kinetic_energy;  /* SW: 0 */
  int limiter_limit_on_square_velocity; /* SW: 1 */

  /* "/riemann solver": source/euler/riemann_solver.h:27-38 */
  int riemann_newton_max_iterations; /* 0 */
  double riemann_newton_tolerance;   /* 1e-10 */

  /* "B - Equation" Euler with arbitrary equation of state (RYUJIN_EQ_EULER_AEOS):
   * source/euler_aeos/hyperbolic_system.h:766-812; `gamma`, `reference_density`,
   * `vacuum_state_relaxation_*` above are shared with Euler */
  int eos;                    /* RYUJIN_EOS_*: "equation of state", polytropic gas */
  int compute_strict_bounds;  /* 1 */
  double eos_covolume_b;      /* NASG, van der Waals: "covolume b", 0 */
  double eos_q;               /* NASG: "reference specific internal energy", 0 */
  double eos_pinf;            /* NASG: "reference pressure", 0 */
  double eos_vdw_a;           /* van der Waals: "vdw a", 0 */
  double eos_gas_constant_R;  /* 287.052874 (van der Waals: 0.4); only temperature() uses it */
  /* Jones-Wilkins-Lee (equation_of_state_jones_wilkins_lee.h:38-66) */
  double jwl_A, jwl_B, jwl_R1, jwl_R2, jwl_omega, jwl_rho_0, jwl_q_0, jwl_cv;

  /* "B - Equation" scalar conservation (RYUJIN_EQ_SCALAR_CONSERVATION):
   * source/scalar_conservation/hyperbolic_system.h:510-540 and "/riemann solver" riemann_solver.h:26-55 */
  int sc_flux;                                /* RYUJIN_FLUX_*: burgers */
  double sc_flux_polynomial[3][4];            /* RYUJIN_FLUX_POLYNOMIAL: c_0..c_3 per direction */
  double sc_derivative_approximation_delta;   /* "function": 1e-10 */
  int sc_use_greedy_wavespeed;                /* 0 */
  int sc_use_averaged_entropy;                /* 0 */
  int sc_random_entropies;                    /* 0; > 0 is not reproducible in the reference: rejected */

  /* Run-time switches of the library itself (no ParameterAcceptor counterpart; all default to 0).
   * system_scope_events: the events that tie the compute stream and the exchange stream of ONE device are
   *   created without the system-scope fence by default (the RCCL kernels carry their own system-scope
   *   semantics); nonzero selects plain hipEventDisableTiming events -- the conservative choice, selectable
   *   without a rebuild should a multi-GPU run ever show stale ghost data (bench.py --gpus N runs both and
   *   compares bitwise before it times anything).
   * debug_*: test hooks that select code branches the mesh would otherwise select (results are identical):
   *   debug_join_exchanges != 0: every sweep joins the ghost exchanges (the choreography of a non-symmetric
   *   stencil); debug_bc_fold_max_slices: boundary conditions ride on the pre-pass kernel up to n slices of
   *   64 rows (0 = default 4096, < 0 = always a launch of their own); debug_no_small_mesh_split != 0: meshes
   *   that do not fill the device run the same step-5/6 kernels as large ones; debug_pij_storage: the matrix
   *   P_ij of an update without stage vectors is stored only where steps 6 and 7 read it (0, the default): up to
   *   two dimensions per (slice, column) tile of 64 entries -- where one of the tile's own l_ij comes out limited
   *   or step 6 read the tile in one of the last updates; step 6 forms the few that are missing --, in 3-D per
   *   64-row slice while few slices hold a limited pair (where the slice held one in the previous update or one of
   *   its own l_ij comes out limited; the rest completed by the repair launch of step 6) and everywhere after
   *   that. 1: always per slice and no slice predicted limited (every stored slice goes through the trigger in
   *   step 5 or the repair launch of step 6); 2: per tile and no tile predicted (every tile the neighbour's l_ji
   *   limits is formed by step 6; per slice as 1 in 3-D); 3: per tile in ANY dimension (3-D: the tiles step 5 did
   *   not store are formed by a launch behind step 6 -- built, measured as a small loss, not the default); 4: as 3
   *   with no tile predicted; < 0: always stored everywhere. */
  int system_scope_events;
  int debug_join_exchanges;
  int debug_bc_fold_max_slices;
  int debug_no_small_mesh_split;
  int debug_pij_storage;
  /* != 0: the reference's EXPENSIVE_BOUNDS_CHECK build (CMakeLists.txt / compile_time_options.h.in:13-16) as a
   * run-time option, for every Description: View::is_admissible() on every new state after steps 4, 6 and 7
   * (hyperbolic_module.template.h:851-855,1121-1126), the limiter's checked control flow -- its additional
   * high-order density and entropy checks (limiter.template.h:110-134,244-252,291-322) -- and the second limiter
   * pass's `success` counted (:1155-1161): any of them raises the restart flag. Evaluated by separate kernels
   * between the sweeps (P_ij is stored in full then); the l_ij themselves are the same in both control flows.
   * (limiter.template.h of euler/, euler_aeos/, shallow_water/ and scalar_conservation/; is_admissible() of the
   * Description's view: trivially true for a scalar equation.) */
  int debug_expensive_bounds_check;
  /* The tile map: column indices and transposed positions of structured 64-row tiles come from a 16-byte descriptor
   * per tile instead of the explicit index arrays (ryujin_amd/csrc/host_layout.hpp, TileDesc; the 1-D / 2-D sweeps 3,
   * 5, 6, 7), and step 5 takes the node data of runs of consecutive neighbours from a neighbouring lane instead of
   * gathering it (TileDesc::chain, every dimension). 0: on (default); < 0: off -- every sweep reads the explicit
   * arrays and gathers every neighbour, as for an unstructured mesh. Results are identical. */
  int debug_tile_map;
  /* Stacked blocks: the four waves of a workgroup take 64-row slices that are one lattice row (2-D) / lattice plane
   * (3-D) of a structured mesh apart instead of four consecutive ones, so that their vertical neighbour rows are
   * served by one L2 (ryujin_amd/csrc/kernels_euler.hpp, row_context()). 0: chosen from the mesh (or off, see
   * DESIGN.md section 3); > 0: that many slices; < 0: off. Results are identical: every slice is processed by exactly
   * one wave either way. */
  int debug_band_stride;
  /* XCD-local block ranges: the hardware deals the workgroups of a launch out to the eight XCDs round robin (block b
   * runs on XCD b % 8 -- observed, relied on for speed only), so that the rows one L2 serves are scattered over the
   * whole mesh and every L2 fetches every node's data. With debug_xcd_chunk = C > 0 the blocks of a launch are
   * renumbered in chunks of 8 C: inside a chunk XCD x takes the C consecutive blocks [x C, (x + 1) C) -- each L2
   * then serves a contiguous range of 4 C slices at a time, and the eight XCDs still advance through the mesh
   * together, chunk by chunk (one contiguous eighth of the mesh per XCD would put a shock on one XCD).
   * 0: chosen by the library; < 0: off. Results are identical: every slice is processed by
   * exactly one wave either way (ryujin_amd/csrc/kernels_euler.hpp, row_context()). */
  int debug_xcd_chunk;
} ryujin_hip_params;

/* ---- offline data (input contract) ------------------------------------- */
/*
 * Read-only stencil/geometry data, the output of OfflineData::prepare()
 * (source/offline_data.h:121-264), in the reference's local numbering
 *    [0,n_export) c [0,n_internal) c [0,n_owned) c [0,n_relevant)
 * (source/offline_data.template.h:210-272) and in the SparsityPatternSIMD
 * storage scheme (source/sparse_matrix_simd.h:311-350,403-418):
 *   rows [0,n_internal): groups of simd_length rows of equal length,
 *     entry (row,col_idx,comp) at
 *       data[(row_starts[row/sl] + col_idx*sl)*n_comp + comp*sl + row%sl]
 *     column index at columns[row_starts[row/sl] + col_idx*sl + row%sl];
 *     row_starts has one entry per group, i.e. indices 0..n_internal/sl;
 *   rows [n_internal,n_relevant): plain CSR,
 *       data[(row_starts[row] + col_idx)*n_comp + comp],
 *     where row_starts is indexed by the row itself (the reference re-bases
 *     row_starts[n_internal] := row_starts[n_internal/sl],
 *     sparse_matrix_simd.template.h:127).
 *   row_starts therefore has n_relevant+1 entries of which only
 *   [0, n_internal/sl] and [n_internal, n_relevant] are meaningful.
 *   simd_length == 1 (or n_internal == 0) is plain CSR.
 * Column 0 of every row is the diagonal, the rest ascend by local index
 * (deal.II square SparsityPattern; relied on at
 * hyperbolic_module.template.h:394-396). Ghost rows [n_owned,n_relevant)
 * hold the diagonal and those entries whose transpose is locally owned
 * (sparse_matrix_simd.template.h:61-74).
 */
typedef struct ryujin_hip_offline {
  uint32_t n_export, n_internal, n_owned, n_relevant;
  uint32_t simd_length;
  const uint64_t *row_starts; /* [n_relevant+1], see above */
  const uint32_t *columns;    /* [nnz] */
  const double *cij;          /* [nnz*dim]  c_ij = int phi_i grad phi_j  (offline_data.template.h:566-576) */
  const double *mij;          /* [nnz]      m_ij = int phi_i phi_j */
  const double *mi;           /* [n_relevant] lumped mass  (offline_data.template.h:790-802) */
  const double *mi_inv;       /* [n_relevant] */
  double measure_of_omega;

  /* boundary_map() (offline_data.h:66-72), SoA, in application order */
  uint32_t n_bdry;
  const uint32_t *b_i;      /* [n_bdry] local index (owned) */
  const double *b_normal;   /* [n_bdry*dim] unit normal */
  const uint8_t *b_id;      /* [n_bdry] RYUJIN_BC_* */

  /* coupling_boundary_pairs() (offline_data.h:74-82) */
  uint32_t n_pairs;
  const uint32_t *p_i, *p_col, *p_j; /* [n_pairs] each */

  /* initial_precomputed (shallow water bathymetry Z_i); NULL for Euler.
   * hyperbolic_module.template.h:84-85 */
  const double *initial_precomputed; /* [n_relevant * n_initial_precomputed] */

  /*
   * Exchange pattern (all zero / NULL for a single rank). Neighbour ranks in
   * ascending order. Vector exchange = dealii Partitioner semantics
   * (ghost range sorted by owner, receive in place):
   *   send to nbr p the owned entries send_idx[send_off[p]..send_off[p+1]),
   *   receive from nbr p into ghost rows [recv_off[p], recv_off[p+1])
   *   (recv_off[0] == n_owned, recv_off[n_nbr] == n_relevant).
   * Matrix ghost-row exchange (sparse_matrix_simd.h:649-763):
   *   send to nbr p the entries (row_send_row[e], row_send_col[e]),
   *   e in [row_send_off[p], row_send_off[p+1]); received values fill the
   *   ghost rows of that neighbour contiguously in storage order.
   */
  int n_nbr;
  const int *nbr_rank;       /* [n_nbr] */
  const uint32_t *send_off;  /* [n_nbr+1] */
  const uint32_t *send_idx;  /* [send_off[n_nbr]] */
  const uint32_t *recv_off;  /* [n_nbr+1] */
  const uint32_t *row_send_off; /* [n_nbr+1] */
  const uint32_t *row_send_row; /* [row_send_off[n_nbr]] */
  const uint32_t *row_send_col; /* [row_send_off[n_nbr]] */

  /*
   * Discontinuous finite element ansatz (Discretization::have_discontinuous_ansatz(), SURVEY.md
   * section 8 f-4). When nonzero the step uses the incidence matrix in the high-order viscosity
   * (hyperbolic_module.template.h:733-737), the full block-diagonal inverse mass matrix instead of the
   * Neumann series (:976-986) and extends the limiter bounds over the stencil (:938-948).
   * Both matrices in the storage scheme of mij. Any number of ranks (the limiter bounds are exchanged,
   * :601-613); Euler, EulerAEOS and shallow water. Scalar conservation is refused with RYUJIN_ERR_UNSUPPORTED:
   * the reference's scalar Riemann solver is 0/0 on the structural zeros of a dG stencil (n_ij = c_ij / |c_ij|
   * with c_ij = 0; scalar_conservation/riemann_solver.template.h:63,95-96), tests/test_dg_q1.py.
   */
  int discontinuous_ansatz;
  const double *incidence;           /* [nnz] OfflineData::incidence_matrix() */
  const double *mass_matrix_inverse; /* [nnz] OfflineData::mass_matrix_inverse() */
} ryujin_hip_offline;

typedef struct ryujin_hip_ctx ryujin_hip_ctx; /* opaque; one per (rank, GPU) */

/* ---- communicator (RCCL over xGMI) -------------------------------------- */
/*
 * One process per GPU. Rank 0 calls comm_unique_id(), the caller broadcasts
 * the 128 bytes by whatever means it has (MPI_Bcast in ryujin, a
 * torch.distributed broadcast in bench.py), every rank calls comm_init().
 * Replaces the reference's MPI calls listed in SURVEY.md section 2.2.
 */
#define RYUJIN_HIP_UNIQUE_ID_BYTES 128
typedef struct ryujin_hip_comm ryujin_hip_comm;
int ryujin_hip_comm_unique_id(char id[RYUJIN_HIP_UNIQUE_ID_BYTES]);
int ryujin_hip_comm_init(ryujin_hip_comm **comm, const char id[RYUJIN_HIP_UNIQUE_ID_BYTES],
                         int rank, int n_ranks, int device);
/* number of HIP devices visible to this process (a host application picks rank % count) */
int ryujin_hip_device_count(int *n_devices);
/* Test facility: n_ranks communicators of ONE process (one host thread per rank, all on `device`)
 * that exchange ghost data with device-to-device copies instead of RCCL. comms: [n_ranks]. */
int ryujin_hip_comm_init_local(ryujin_hip_comm **comms, int n_ranks, int device);
/* Measurement facility: a communicator for ONE rank of an n_ranks slab partition that is run alone; every
 * neighbour is replaced by the rank itself (what it packs for the opposite neighbour arrives in its ghost range,
 * i.e. a periodic channel). All launches, pack kernels, copies, events and reductions of a middle rank of a real
 * run, without the network: what the multi-rank choreography costs per rank (scripts/overhead_loopback.py). The
 * offline data must have two neighbours with equally sized send / ghost ranges (uniform cross-section). */
int ryujin_hip_comm_init_loopback(ryujin_hip_comm **comm, int rank, int n_ranks, int device);
void ryujin_hip_comm_destroy(ryujin_hip_comm *comm);
/* What the communicator itself reports: rank / n_ranks as passed to comm_init and -- RCCL communicators only,
 * -1 otherwise -- ncclCommUserRank / ncclCommCount / ncclCommCuDevice as RCCL sees them (bench.py prints them
 * so that "RCCL saw N ranks" can be read off the JSON line). Any pointer may be NULL. */
int ryujin_hip_comm_info(const ryujin_hip_comm *comm, int *rank, int *n_ranks, int *rccl_rank,
                         int *rccl_count, int *rccl_device);

/* ---- lifecycle ----------------------------------------------------------- */
void ryujin_hip_default_params(ryujin_hip_params *params, int equation, int dim);

/* HyperbolicModule ctor + prepare() (hyperbolic_module.template.h:28-86).
 * `comm` may be NULL for a single rank. `device` is the HIP device ordinal. The library reads no
 * environment variables; its run-time switches are the last four fields of ryujin_hip_params. */
int ryujin_hip_create(ryujin_hip_ctx **ctx, const ryujin_hip_offline *offline,
                      const ryujin_hip_params *params, ryujin_hip_comm *comm, int device);
/* The exchange pattern of a context and what it has done so far: neighbour ranks (at most max_nbr written),
 * number of point-to-point ghost exchanges (vector or matrix; one grouped send/recv set per neighbour each)
 * and of scalar all-reduces enqueued since create(). Any pointer may be NULL. */
int ryujin_hip_exchange_info(ryujin_hip_ctx *ctx, int *n_nbr, int *nbr_rank, int max_nbr,
                             unsigned long long *n_exchanges, unsigned long long *n_allreduces);
void ryujin_hip_destroy(ryujin_hip_ctx *ctx);

/* ---- state vectors (StateVector = U + precomputed, source/state_vector.h:47-51) */
int ryujin_hip_state_alloc(ryujin_hip_ctx *ctx, int *handle);
int ryujin_hip_state_free(ryujin_hip_ctx *ctx, int handle);
/* U_aos: MultiComponentVector layout U[i*k + d], i < n_relevant
 * (source/multicomponent_vector.h:55-183) */
int ryujin_hip_state_upload(ryujin_hip_ctx *ctx, int handle, const double *U_aos);
int ryujin_hip_state_download(ryujin_hip_ctx *ctx, int handle, double *U_aos);
int ryujin_hip_state_download_precomputed(ryujin_hip_ctx *ctx, int handle, double *prec_aos);

/* ---- host-mirrored state vectors: the unmodified reference caller ----------------------------------------
 * The reference's StateVector lives on the HOST and is read and written there by its caller between the calls
 * into HyperbolicModule (TimeIntegrator: sadd(), swap(); time_integrator.template.h:18-25,300-328). An adapter
 * that keeps such a caller unchanged mirrors every call over PCIe; these entry points move only what the
 * contract of the call can have changed, straight between the caller's arrays and HBM:
 *   host_register(ptr, bytes)       pin the caller's array IN PLACE (hipHostRegister): uploads and downloads of
 *                                   that memory are then direct DMA instead of a pageable staged copy.
 *                                   Registrations belong to the context and end with it (or with
 *                                   host_unregister). THE CALLER GUARANTEES that the memory stays allocated while
 *                                   it is registered: a range that is freed and handed out again by the allocator
 *                                   is not covered by the old pinning (registering the same address again renews
 *                                   it). RYUJIN_WARN (not an error): the pages could not be pinned; transfers
 *                                   still work, staged by the runtime.
 *   state_download_owned(h, U)      rows [0, n_owned) only: what step() is allowed to write
 *                                   (hyperbolic_module.h:207-213); U_aos has room for n_relevant rows, the
 *                                   ghost rows are left untouched.
 *   state_download_prepared(h, U)   the rows prepare_state_vector() can have changed in a vector the caller
 *                                   has just uploaded: the boundary_map rows (boundary conditions,
 *                                   hyperbolic_module.template.h:119-147) and the ghost range (:152-159).
 *                                   Everything else in U_aos is left untouched. */
int ryujin_hip_host_register(ryujin_hip_ctx *ctx, const void *ptr, size_t bytes);
int ryujin_hip_host_unregister(ryujin_hip_ctx *ctx, const void *ptr);
int ryujin_hip_state_download_owned(ryujin_hip_ctx *ctx, int handle, double *U_aos);
int ryujin_hip_state_download_prepared(ryujin_hip_ctx *ctx, int handle, double *U_aos);

/* ---- the hot path -------------------------------------------------------- */
/*
 * prepare_state_vector(state_vector, t) (hyperbolic_module.template.h:96-193).
 * dirichlet_aos: [n_bdry*k] the value of initial_values_->initial_state(
 * position, t) for every boundary_map entry (only read for dirichlet /
 * dynamic / dirichlet_momentum ids); may be NULL if none of those ids occur
 * or to re-use the data passed in the previous call.
 */
int ryujin_hip_prepare_state_vector(ryujin_hip_ctx *ctx, int handle, double t,
                                    const double *dirichlet_aos);

/*
 * step<stages>(old, stage_state_vectors, stage_weights, new, tau, tau_max)
 * (hyperbolic_module.template.h:234-1211). stages in [0,4]. Writes only
 * new.U on [0,n_owned). *tau_out = the tau actually used. Returns RYUJIN_OK,
 * RYUJIN_WARN, RYUJIN_RESTART or RYUJIN_ERR_TAU; counters as the reference.
 */
int ryujin_hip_step(ryujin_hip_ctx *ctx, int h_old, int stages, const int *h_stage,
                    const double *stage_weights, int h_new, double tau_in,
                    double tau_max_in, double *tau_out);

/* TimeIntegrator helpers that touch the device-resident vectors:
 * sadd(dst,s,b,src): dst.U = s*dst.U + b*src.U (time_integrator.template.h:18-25) */
int ryujin_hip_sadd(ryujin_hip_ctx *ctx, int h_dst, double s, double b, int h_src);

/* Device-resident TimeIntegrator::step (source/time_integrator.template.h:207-403) for the explicit
 * schemes built from prepare_state_vector + step<s> + sadd. One host synchronisation per RK step: the
 * tau of the first stage and the restart flags of all stages stay on the device. h_state names the
 * solution before and after the call (state_vector.swap(temp) is done on the handles); h_tmp are three
 * scratch state vectors (temp_[0..2]). dirichlet_aos as in prepare_state_vector (time independent;
 * ryujin_hip_time_step_fn below takes time-dependent Dirichlet data). tau_max = t_final - t. With
 * RYUJIN_CFL_RECOVERY_BANG_BANG the reference's bang-bang control (:250-274) is applied internally.
 * *tau_out = the time increment of the whole RK step (3 tau for ERK33). */
enum {
  RYUJIN_SCHEME_SSPRK_22 = 0,
  RYUJIN_SCHEME_SSPRK_33 = 1,
  RYUJIN_SCHEME_ERK_11 = 2,
  RYUJIN_SCHEME_ERK_22 = 3,
  RYUJIN_SCHEME_ERK_33 = 4,
  RYUJIN_SCHEME_ERK_43 = 5, /* step_erk_43, :405-440: 4 temporaries */
  RYUJIN_SCHEME_ERK_54 = 6  /* step_erk_54, :443-510: 5 temporaries */
};
enum { RYUJIN_CFL_RECOVERY_NONE = 0, RYUJIN_CFL_RECOVERY_BANG_BANG = 1 };
int ryujin_hip_time_step(ryujin_hip_ctx *ctx, int scheme, int h_state, const int h_tmp[3],
                         const double *dirichlet_aos, double tau_max, int cfl_recovery, double cfl_min,
                         double cfl_max, double *tau_out);
/* the same with n_tmp temporaries (ERK43 needs 4, ERK54 needs 5; TimeIntegrator::prepare :163-205) */
int ryujin_hip_time_step_n(ryujin_hip_ctx *ctx, int scheme, int h_state, int n_tmp, const int *h_tmp,
                           const double *dirichlet_aos, double tau_max, int cfl_recovery,
                           double cfl_min, double cfl_max, double *tau_out);

/* The same with TIME-DEPENDENT Dirichlet data: the reference evaluates initial_state(position, t + c_s tau) for
 * every stage (hyperbolic_module.template.h:137-139 called with the stage times of
 * time_integrator.template.h:279-510). `dirichlet_fn(user, time, values)` fills values[n_bdry * k] (the layout of
 * prepare_state_vector's dirichlet_aos; entries whose boundary id does not read Dirichlet data may be left
 * untouched) and is called once per stage from the calling thread: for the first stage at `t` before anything is
 * enqueued, for the later stages as soon as tau exists on the host -- behind step 3 of the first stage, while its
 * remaining sweeps run. NULL: no boundary id reads Dirichlet data, or the data of an earlier call is kept. */
typedef void (*ryujin_hip_dirichlet_fn)(void *user, double time, double *dirichlet_aos);
int ryujin_hip_time_step_fn(ryujin_hip_ctx *ctx, int scheme, int h_state, int n_tmp, const int *h_tmp, double t,
                            ryujin_hip_dirichlet_fn dirichlet_fn, void *user, double tau_max, int cfl_recovery,
                            double cfl_min, double cfl_max, double *tau_out);

/* Conservation monitor on the device (the interior integrals of ryujin::Quantities,
 * source/quantities.template.h): out[q] = sum over ALL ranks of
 * sum_{i < n_owned} m_i U_i[q], q < k. Fixed summation order: bitwise reproducible for a given
 * partition. Collective when the context has a communicator. */
int ryujin_hip_state_integrals(ryujin_hip_ctx *ctx, int handle, double *out /* [k] */);

/* ---- accessors of HyperbolicModule (hyperbolic_module.h:225-278) --------- */
int ryujin_hip_set_cfl(ryujin_hip_ctx *ctx, double cfl);
int ryujin_hip_get_cfl(ryujin_hip_ctx *ctx, double *cfl);
int ryujin_hip_set_id_violation_strategy(ryujin_hip_ctx *ctx, int strategy);
int ryujin_hip_get_alpha(ryujin_hip_ctx *ctx, double *alpha /* [n_relevant] */);
int ryujin_hip_get_counters(ryujin_hip_ctx *ctx, unsigned *n_restarts, unsigned *n_warnings);
/* What the data-dependent limiter sweeps saw between the two latest host synchronisations: the fraction of
 * (sampled) 64-row slices in which the first high-order sweep found a limited pair; how the latest step kept the
 * matrix P_ij (hyperbolic_module.template.h:795-846): 1 stored everywhere, 2 stored per slice -- only where steps
 * 6 and 7 read it --, 3 stored per (slice, column) tile (up to two dimensions); and the fraction
 * of slices (tiles) it was stored in. The results are the same bit for bit whatever is
 * stored -- FOR A GIVEN LAYOUT: which rows share a 64-row slice decides, through wave-uniform
 * masks, which of two roundings of the same sum a row's high-order update takes (V_i - sum (1 - l) lambda P over the
 * limited tiles, or the reference's U_low + sum l lambda P where every tile of the row is limited), so another
 * numbering, partition or rank count reproduces a run to round-off (1e-11 on U per update, the stated contract), not
 * to the bit. Diagnostics; any pointer may be NULL. */
int ryujin_hip_limiter_statistics(ryujin_hip_ctx *ctx, double *limited_slice_fraction, int *pij_stored,
                                  double *stored_slice_fraction);
/* The tile map of the context (ryujin_amd/csrc/host_layout.hpp, TileDesc): number of 64-entry tiles of the owned rows
 * and how many of them are served by a 16-byte descriptor instead of the explicit index arrays (0 where the sweeps
 * keep streaming the indices: 3-D, debug_tile_map < 0). bench.py takes the index bytes the sweeps no longer read out of their own
 * compulsory bytes with it. Any pointer may be NULL. */
int ryujin_hip_layout_info(ryujin_hip_ctx *ctx, unsigned long long *n_tiles, unsigned long long *n_regular_tiles);
/* Chained gathers (ryujin_amd/csrc/host_layout.hpp, TileDesc::chain): the number of tiles whose node data step 5 takes
 * from the previous column of the slice or from the slice's own rows, one lane over, instead of gathering it, and the
 * number of matrix entries (lanes of those tiles) that are served that way -- the others fetch their node as ever. On a
 * lattice-numbered mesh six of the eight off-diagonal columns of a 2-D Q1 row, 18 of 26 in 3-D; 0 on a mesh without
 * such runs and with debug_tile_map < 0. The results do not depend on it (the same values reach the same operations).
 * Diagnostics; any pointer may be NULL. */
int ryujin_hip_chain_info(ryujin_hip_ctx *ctx, unsigned long long *n_chained_tiles,
                          unsigned long long *n_chained_entries);
/* Where the latest step stored P_ij per tile (pij_stored == 3): of the (slice, column) tiles between the two latest
 * host synchronisations, the fractions step 5 stored, step 6 read, and step 6 had to form itself because step 5 had
 * not stored them (ryujin_amd/csrc/kernels_limiter_stage0.hpp). 1 / 1 / 0 otherwise. Any pointer may be NULL. */
int ryujin_hip_tile_statistics(ryujin_hip_ctx *ctx, double *stored_fraction, double *read_fraction,
                               double *formed_by_step6_fraction);
/* Where the tiles step 5 did not store are formed outside the sweep of step 6 (3-D; ryujin_amd/csrc/kernels_limiter.hpp,
 * kHoDefer): the number of 64-row slices the latest update that ended in a host synchronisation (ryujin_hip_step, or the
 * last stage of ryujin_hip_time_step) handed to the launch behind the sweep -- exact, not sampled. 0 where the repair
 * is part of the sweep. */
int ryujin_hip_deferred_slices(ryujin_hip_ctx *ctx, unsigned *n_slices);

/* ---- introspection for parity tests and profiling ------------------------ */
/* Module-owned intermediates of the LAST step() in the reference's logical
 * (row, col_idx) order as plain CSR over owned rows (nnz_owned entries):
 * what: 0 d_ij, 1 l_ij (after the last pass = lij_matrix_), 2 p_ij (k comps),
 *       3 bounds (n_bounds per row), 4 r_i (k per row), 5 lij_next;
 * over ALL locally relevant rows -- the ghost rows / ghost range a rank received included (multi-rank parity
 * tests): 6 d_ij, 7 l_ij, 8 lij_next (plain CSR over n_relevant rows; d_ij is never exchanged, its ghost rows
 * stay zero as in the reference), 9 r_i (k per row, n_relevant rows) */
int ryujin_hip_debug_fetch(ryujin_hip_ctx *ctx, int what, double *out, size_t n_doubles);
/* device time [ms] of the sweeps of the last step (hipEvent pairs; ms[n] = the reference's Scope timer
 * "time step [H] n", n = 2..7; ms[1] unused (step 1 is a separate call); ms[0] = the indicator kernel
 * alone when step 2 runs as two kernels, else 0); enable = nonzero switches the event recording on. */
int ryujin_hip_set_timers(ryujin_hip_ctx *ctx, int enable);
int ryujin_hip_get_timers(ryujin_hip_ctx *ctx, double ms[8]);
/* sums over all updates since the last reset (also filled by ryujin_hip_time_step) */
int ryujin_hip_get_timers_accum(ryujin_hip_ctx *ctx, double ms[8], unsigned *n_updates, int reset);
/* Block until all device work of this context has finished. */
int ryujin_hip_synchronize(ryujin_hip_ctx *ctx);
/* Record start/stop HIP events on the context's compute stream and read the elapsed ms. */
int ryujin_hip_event_record(ryujin_hip_ctx *ctx, int which /*0 start, 1 stop*/);
int ryujin_hip_event_elapsed_ms(ryujin_hip_ctx *ctx, double *ms);
/* Host-only check of the layout import (no GPU needed): converts `offline` into the device layout
 * (SELL-64 + ghost CSR) and back, and reports the logical (row, col_idx) view over ALL n_relevant
 * rows: ptr [n_relevant+1], col [nnz], transposed [nnz] = logical index of the (j,i) entry.
 * If data != NULL, the n_comp-component matrix `data` (reference layout) is scattered into the
 * device layout and gathered back into out [nnz*n_comp] (AoS per entry). Any pointer may be NULL. */
int ryujin_hip_debug_layout(const ryujin_hip_offline *offline, uint64_t *ptr, uint32_t *col,
                            uint64_t *transposed, const double *data, uint32_t n_comp, double *out);
/* Device addresses of the stencil streams of a context (cols, c_ij, m_ij, d_ij, l_ij, l'_ij, p_ij, idx_t): for
 * the placement study of scripts/placement_probe.py (how the kernel times depend on where the allocator put them). */
int ryujin_hip_debug_addresses(ryujin_hip_ctx *ctx, uint64_t out[8]);
/* Evaluate the device implementation of ryujin::pow (source/simd.template.h:196-272) on n pairs:
 * out[i] = pow(x[i], y[i]). Needs a GPU; used by the parity tests only. */
int ryujin_hip_debug_pow(int device, const double *x, const double *y, double *out, size_t n);
/* Evaluate a device function of the Euler / shallow-water Description on n independent items (needs a GPU;
 * parity tests only): the device code is pinned directly against the baselines of the reference's unit tests.
 *   RYUJIN_DEBUG_EULER_RIEMANN   in: rd_i[4], rd_j[4] = (rho, u, p, a)      out: lambda_max
 *                                (RiemannSolver::compute(rd_i, rd_j), riemann_solver.template.h:406-582;
 *                                 tests/euler/riemann_solver.cc:79-98)
 *   RYUJIN_DEBUG_EULER_LIMIT_1D  in: bounds[3], U[3], P[3] (dim = 1)        out: l, success, took the Newton tail
 *                                (Limiter::limit, limiter.template.h:15-327; tests/euler/limiter.cc:61-139)
 *   RYUJIN_DEBUG_EULER_LIMIT_CHECKED_1D  the same through the EXPENSIVE_BOUNDS_CHECK control flow (the build that
 *                                wrote tests/euler/limiter.output)           out: l, success, 0
 *   RYUJIN_DEBUG_SW_RIEMANN      in: rd_i[3], rd_j[3] = (h, u, a)           out: h_star, lambda_max
 *                                (shallow_water/riemann_solver.template.h:110-251;
 *                                 tests/shallow_water/riemann_solver.cc:75-77)
 *   RYUJIN_DEBUG_EULER_DIJ_2D/3D in: U_i[k], U_j[k], c_ij[dim]              out: d_ij = |c_ij| lambda_max
 *                                (hyperbolic_module.template.h:402-406) */
enum {
  RYUJIN_DEBUG_EULER_RIEMANN = 0,
  RYUJIN_DEBUG_EULER_LIMIT_1D = 1,
  RYUJIN_DEBUG_SW_RIEMANN = 2,
  RYUJIN_DEBUG_EULER_DIJ_2D = 3,
  RYUJIN_DEBUG_EULER_DIJ_3D = 4,
  RYUJIN_DEBUG_EULER_DIJ_RECORDS_2D = 5, /* the same through the per-node Riemann records the sweep uses */
  RYUJIN_DEBUG_EULER_DIJ_RECORDS_3D = 6,
  RYUJIN_DEBUG_SW_DIJ_2D = 7,          /* in: U_i[3], U_j[3], c_ij[2]   out: d_ij (shallow water, dim = 2) */
  RYUJIN_DEBUG_SW_DIJ_RECORDS_2D = 8,  /* the same through the per-node Riemann records the sweep uses */
  /* the production evaluation path of the sweeps -- per-node Riemann records, dij_from_records -- fed with the
   * reference's 1-D Riemann data: in as RYUJIN_DEBUG_EULER_RIEMANN / RYUJIN_DEBUG_SW_RIEMANN, out: lambda_max */
  RYUJIN_DEBUG_EULER_RIEMANN_RECORDS = 9,
  RYUJIN_DEBUG_SW_RIEMANN_RECORDS = 10,
  /* EulerAEOS (params->eos etc. select the equation of state):
   *   RYUJIN_DEBUG_AEOS_RIEMANN  in: rd_i[5], rd_j[5] = (rho, u, p, gamma, a)   out: lambda_max, evaluated through
   *                              the per-node records of the sweep (EulerAeos::dij_from_records with n = 1)
   *                              (euler_aeos/riemann_solver.template.h:443-560; tests/euler_aeos/riemann_solver*.cc)
   *   RYUJIN_DEBUG_AEOS_LIMIT_1D in: bounds[4], U[3], P[3] (dim = 1)            out: l, success, took the Newton tail
   *                              (euler_aeos/limiter.template.h:15-360; tests/euler_aeos/limiter*.cc) */
  RYUJIN_DEBUG_AEOS_RIEMANN = 11,
  RYUJIN_DEBUG_AEOS_LIMIT_1D = 12,
  /* EulerAEOS d_ij from two states (dim = 2; p from the equation of state): in U_i[4], U_j[4], c_ij[2], out d_ij --
   * in the reference's operation order and through the per-node Riemann records the sweep uses */
  RYUJIN_DEBUG_AEOS_DIJ_2D = 13,
  RYUJIN_DEBUG_AEOS_DIJ_RECORDS_2D = 14,
  RYUJIN_DEBUG_EULER_LIMIT_CHECKED_1D = 15,
  /* the limiter as the sweeps compose it, dim = 2: in bounds[3], U[4], P[4]; out l, success, took the Newton tail,
   * t_r behind the density clip, psi_r of the first Newton iteration (as the device evaluates it) */
  RYUJIN_DEBUG_EULER_LIMIT_2D = 16
};
int ryujin_hip_debug_function(int device, const ryujin_hip_params *params, int which, const double *in,
                              double *out, size_t n);
/* Host logic of ryujin_hip_time_step, exposed for the CPU test-suite: what the end-of-step flags mean.
 * restart_accum / tau_invalid_accum: 0 = never raised, else 100 - (index of the first RK stage that raised it).
 * Returns RYUJIN_ERR_TAU, RYUJIN_RESTART, RYUJIN_WARN or RYUJIN_OK. Under raise_exception the reference throws
 * Restart at the end of the offending stage (hyperbolic_module.template.h:1194-1207) and never evaluates tau_max
 * of a later stage (:573-576); here all stages are enqueued before the flags are read, so an invalid tau_max of a
 * stage LATER than the first Restart must not turn the recoverable Restart into the fatal error. */
int ryujin_hip_debug_rk_outcome(int restart_accum, int tau_invalid_accum, int id_violation_strategy);
const char *ryujin_hip_last_error(void);
const char *ryujin_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RYUJIN_HIP_H */

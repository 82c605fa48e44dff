// libryujin_hip.so -- C ABI (include/ryujin_hip.h) over the HIP kernels.
//
// One context per (rank, GPU). Everything the module owns lives in HBM for the lifetime of
// the context (hyperbolic_module.h:319-333: alpha, bounds, r, d_ij, l_ij, l_ij_next, p_ij);
// state vectors are device resident behind handles. All sweeps of one step() are enqueued on
// one HIP stream without host synchronisation; the only host<->device round trip per step is
// the 24-byte scalar read-back (tau, restart flag) at the end.
// Multi-GPU: ghost exchange as RCCL point-to-point (ncclSend/ncclRecv grouped per neighbour)
// over xGMI, tau_max / restart flag as 1-element all-reduces, all stream ordered.

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <limits>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <type_traits>
#include <vector>

#include "ryujin_hip.h"

#ifndef RYUJIN_TILE_PIJ
#define RYUJIN_TILE_PIJ 1 /* the plain kernels (most slices limited) store P_ij per (slice, column) tile */
#endif
#ifndef RYUJIN_BAND_DEFAULT
#ifndef RYUJIN_XCD_CHUNK_3D
#define RYUJIN_XCD_CHUNK_3D 8 /* XCD-local block ranges (ryujin_hip_params::debug_xcd_chunk == 0), blocks per XCD and chunk. Counted and timed
                                 (profiles/r06a_xcd_probe_sedov3d.md, r06g_xcd_probe_*.md): the L2-miss read traffic of the sweeps falls
                                 in every configuration -- C3 step 5 15.7 -> 10.1 GB, step 2 13.3 -> 9.6 GB per launch; C2 -20 ... -25 % in
                                 every sweep -- and the TIME follows only on C3 (200^3, planes of 4.5 MB: step 5 -5 %, the update
                                 -2.1 %); C4 +-1 %, C2 and C5 +-0.2 %. The sweeps are not bound by the bytes their gathers re-fetch
                                 across XCDs (HBM itself delivers 5.8 - 6.7 TB/s for their read/write mixes,
                                 profiles/r06h_hbm_mix.md): on in 3-D, where it is a gain or a wash, off below */
#endif
#define RYUJIN_BAND_DEFAULT 1 /* stacked blocks chosen from the mesh when ryujin_hip_params::debug_band_stride == 0, in 2-D:
                                 C2 (G = 47) -1.1 % per update, every sweep a little; in 3-D stacking lattice planes (G = 365
                                 on the C4 share) LOSES 1.7 % and stacking lattice rows (G = 2) is noise -- the re-fetches of
                                 neighbour data across XCDs that the counters show are not what limits the sweeps
                                 (profiles/r05p_ab_band_2d.log, r05p_ab_band_3d.log) */
#endif

#include "host_layout.hpp"
#include "kernels_euler.hpp"
#include "kernels_limiter.hpp"
#include "kernels_limiter_stage0.hpp"
#include "kernels_euler_aeos.hpp"
#include "kernels_shallow_water.hpp"
#include "scalar_conservation_device.hpp"

using namespace ryujin_hip;

namespace
{
  thread_local std::string g_error;

  struct HipError : std::runtime_error {
    int status;
    HipError(int status, const std::string &what)
        : std::runtime_error(what)
        , status(status)
    {
    }
  };

#define HIP_CHECK(expr)                                                                        \
  do {                                                                                         \
    const hipError_t err_ = (expr);                                                            \
    if (err_ != hipSuccess)                                                                    \
      throw HipError(RYUJIN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(err_));     \
  } while (0)

#define NCCL_CHECK(expr)                                                                       \
  do {                                                                                         \
    const ncclResult_t res_ = (expr);                                                          \
    if (res_ != ncclSuccess)                                                                   \
      throw HipError(RYUJIN_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(res_));   \
  } while (0)

  template <typename T>
  struct DeviceBuffer {
    T *ptr = nullptr;
    void *base = nullptr;
    size_t n = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { release(); }
    void release()
    {
      if (base)
        (void)hipFree(base);
      ptr = nullptr;
      base = nullptr;
      n = 0;
    }
    void alloc(size_t count, bool zero = true)
    {
      release();
      n = count;
      if (count == 0)
        count = 1;
      /* (the allocator hands out 2 MiB aligned blocks; shifting the streams of a context against each other by
       * 4.25 KiB or 260 KiB per stream changed no kernel time: profiles/r03i_placement_*.log) */
      HIP_CHECK(hipMalloc(&base, count * sizeof(T)));
      ptr = static_cast<T *>(base);
      if (zero) {
        /* hipMemset on device memory is asynchronous and ordered on the NULL stream, which does not
         * synchronise with the non-blocking streams of the context: a kernel launched right after the
         * allocation could otherwise be overwritten by the late memset (seen as a flaky zero result of
         * the first ryujin_hip_state_integrals call) */
        HIP_CHECK(hipMemset(ptr, 0, count * sizeof(T)));
        HIP_CHECK(hipDeviceSynchronize());
      }
    }
    void upload(const std::vector<T> &host)
    {
      alloc(host.size(), false);
      if (!host.empty())
        HIP_CHECK(hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    }
    void upload(const T *host, size_t count)
    {
      alloc(count, false);
      if (count)
        HIP_CHECK(hipMemcpy(ptr, host, count * sizeof(T), hipMemcpyHostToDevice));
    }
  };

  EulerParams make_euler_params(const ryujin_hip_params &p)
  {
    EulerParams q{};
    q.gamma = p.gamma;
    q.gamma_inverse = 1. / p.gamma;
    q.gamma_plus_one_inverse = 1. / (p.gamma + 1.);
    q.gamma_minus_one_inverse = 1. / (p.gamma - 1.);
    q.reference_density = p.reference_density;
    q.vacuum_small = p.vacuum_state_relaxation_small;
    q.vacuum_large = p.vacuum_state_relaxation_large;
    q.evc_factor = p.indicator_evc_factor;
    q.lim_newton_tolerance = p.limiter_newton_tolerance;
    q.lim_relaxation_factor = p.limiter_relaxation_factor;
    q.lim_newton_max_iterations = p.limiter_newton_max_iterations;
    q.riemann_newton_max_iterations = p.riemann_newton_max_iterations;
    q.riemann_newton_tolerance = p.riemann_newton_tolerance;
    {
      /* gamma = 7/5 in binary gives 2 gamma / (gamma - 1) = 7 (1 + 3e-16): integral to round-off counts
       * (x^7 against x^(7 + 2e-15) differs by 2e-15 |ln x|, far inside the 1e-12 contract on d_ij) */
      const double e = 2. * p.gamma / (p.gamma - 1.);
      q.rarefaction_power =
          (std::fabs(e - std::rint(e)) <= 8. * std::numeric_limits<double>::epsilon() * e && e >= 1. && e <= 16.)
              ? (int)std::rint(e)
              : 0;
    }
  
    return q;
  }

  EulerAeosParams make_aeos_params(const ryujin_hip_params &p)
  {
    EulerAeosParams aeosparams{};
    aeosparams.eos = p.eos;
    aeosparams.strict = p.compute_strict_bounds != 0;
    aeosparams.gamma = p.gamma;
    aeosparams.eos_b = p.eos_covolume_b;
    aeosparams.eos_q = p.eos_q;
    aeosparams.eos_pinf = p.eos_pinf;
    aeosparams.vdw_a = p.eos_vdw_a;
    aeosparams.jwl_A = p.jwl_A;
    aeosparams.jwl_B = p.jwl_B;
    aeosparams.jwl_R1 = p.jwl_R1;
    aeosparams.jwl_R2 = p.jwl_R2;
    aeosparams.jwl_omega = p.jwl_omega;
    aeosparams.jwl_rho_0 = p.jwl_rho_0;
    aeosparams.jwl_q_0 = p.jwl_q_0;
    /* interpolation parameters of the surrogate (equation_of_state_noble_abel_stiffened_gas.h:52-56,
     * equation_of_state_van_der_waals.h:46-52; zero for the other equations of state) */
    aeosparams.b = aeosparams.pinf = aeosparams.q = 0.;
    if (p.eos == RYUJIN_EOS_NOBLE_ABEL_STIFFENED_GAS) {
      aeosparams.b = p.eos_covolume_b;
      aeosparams.pinf = p.eos_pinf;
      aeosparams.q = p.eos_q;
    } else if (p.eos == RYUJIN_EOS_VAN_DER_WAALS) {
      aeosparams.b = p.eos_covolume_b;
      if (p.eos_covolume_b > 0.)
        aeosparams.pinf = p.eos_vdw_a / (p.eos_covolume_b * p.eos_covolume_b);
    }
    aeosparams.reference_density = p.reference_density;
    aeosparams.vacuum_small = p.vacuum_state_relaxation_small;
    aeosparams.vacuum_large = p.vacuum_state_relaxation_large;
    aeosparams.evc_factor = p.indicator_evc_factor;
    aeosparams.lim_newton_tolerance = p.limiter_newton_tolerance;
    aeosparams.lim_relaxation_factor = p.limiter_relaxation_factor;
    aeosparams.lim_newton_max_iterations = p.limiter_newton_max_iterations;
    return aeosparams;
  }

  ShallowWaterParams make_sw_params(const ryujin_hip_params &p)
  {
    ShallowWaterParams q{};
    q.gravity = p.gravity;
    q.manning = p.manning_friction_coefficient;
    q.reference_water_depth = p.reference_water_depth;
    q.dry_state_relaxation_factor = p.dry_state_relaxation_factor;
    q.dry_small = p.dry_state_relaxation_small;
    q.dry_large = p.dry_state_relaxation_large;
    q.evc_factor = p.indicator_evc_factor;
    q.lim_newton_tolerance = p.limiter_newton_tolerance;
    q.lim_relaxation_factor = p.limiter_relaxation_factor;
    q.limit_on_kinetic_energy = p.limiter_limit_on_kinetic_energy;
    q.limit_on_square_velocity = p.limiter_limit_on_square_velocity;
    return q;
  }

  /* Events that order two streams of ONE device: no timing, and no system-scope fence when they complete (the
   * kernels on either side carry their own device-scope release/acquire; a system-scope fence per event writes
   * back and invalidates the L2s under the interior launch that runs next to it). */
  constexpr unsigned kDeviceEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;

  /* (a runtime that does not know the flag -- the process may run on the ROCm runtime its host application
   * bundles -- gets the plain synchronisation event; so does a context created with
   * ryujin_hip_params::system_scope_events != 0) */
  void create_device_event(hipEvent_t *e, const bool system_scope = false)
  {
    if (system_scope || hipEventCreateWithFlags(e, kDeviceEventFlags) != hipSuccess) {
      (void)hipGetLastError();
      HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
  }

  /* waves of k_pij_lij_recompute an MI355X holds at once (256 CUs x 4 SIMDs x 2): below that step 5 splits the
   * columns of a slice over several waves */
  constexpr uint32_t kResidentWavesStep5 = 2048;
  /* the same for the first of the two high-order sweeps (4 waves per SIMD) */
  constexpr uint32_t kResidentWavesStep6 = 4096;
  /* boundary conditions ride on the pre-pass kernel up to this many slices (262 k gridpoints), see BcFold */
  constexpr uint32_t kBcFoldMaxSlices = 4096;

  int grid_for(size_t n, int block = kBlock) { return (int)std::max<size_t>(1, (n + block - 1) / block); }
} // namespace

/* In-process transport (test facility): several contexts of ONE process, each driven by its own host
 * thread, exchange ghost data on a single GPU (RCCL refuses two ranks on one device). It shares the pack
 * kernels, send/receive offsets, ghost-row layout, streams and events with the RCCL path and -- like RCCL --
 * is purely STREAM ORDERED: no hipStreamSynchronize, no device-side idle point. An ncclSend/ncclRecv pair
 * becomes "neighbour records an event behind its pack kernel; my comm_stream waits for that event and pulls
 * the segment with a device-to-device copy"; because the pull runs on the RECEIVER's stream, a second event
 * in the opposite direction keeps the sender from re-packing its send buffer before every neighbour has
 * pulled (with RCCL the send kernel itself sits on the sender's stream). The host threads only rendezvous
 * on generation counters so that an event is always recorded before a wait on it is enqueued (the one
 * ordering HIP requires; it also makes the cross-stream dependency graph acyclic). A missing
 * hipStreamWaitEvent in the shared choreography is therefore a real data race here, as it would be with
 * RCCL. All-reduces: every rank writes its value into a device slot (double buffered by parity), records
 * an event, waits for the events of all other ranks and reduces the slots with a one-thread kernel. */
struct LocalGroup {
  static constexpr int kRing = 2;
  int n_ranks;
  std::mutex mtx;
  std::condition_variable cv;
  int arrived = 0;
  unsigned long generation = 0;
  /* exchange number g: packed[r] / pulled[r] = number of exchanges whose event rank r has recorded */
  std::vector<unsigned long> packed, pulled, reduced;
  std::vector<hipEvent_t> ev_packed, ev_pulled, ev_reduced; /* [rank * kRing + g % kRing] */
  std::vector<std::vector<const double *>> mail;           /* [src][dst] -> segment in src's send buffer */
  unsigned long long *slots = nullptr;                      /* device: [2][n_ranks][2] 8-byte slots */
  std::vector<double> scratch_vec;                          /* [n_ranks][8] vector sums (host) */
  int refs = 0;
  int device = 0;
  bool aborted = false; /* a rank failed: the others must not wait for it (guarded_ctx sets it) */
  /* Measurement facility: ONE rank of an n-rank slab partition run alone, every neighbour replaced by the rank
   * itself (the segment packed for the opposite neighbour is pulled into the ghost range: a periodic channel).
   * Same launches, pack kernels, copies, events and reductions as a middle rank of a real run, on an unshared
   * GPU -- what the choreography costs per rank, without the network (scripts/overhead_loopback.py). */
  bool loopback = false;

  LocalGroup(int n, int dev)
      : n_ranks(n)
      , packed(n, 0)
      , pulled(n, 0)
      , reduced(n, 0)
      , ev_packed((size_t)n * kRing, nullptr)
      , ev_pulled((size_t)n * kRing, nullptr)
      , ev_reduced((size_t)n * kRing, nullptr)
      , mail(n, std::vector<const double *>(n, nullptr))
      , scratch_vec((size_t)n * 8, 0.)
      , device(dev)
  {
    HIP_CHECK(hipSetDevice(dev));
    for (auto *set : {&ev_packed, &ev_pulled, &ev_reduced})
      for (auto &e : *set)
        create_device_event(&e); /* one GPU by construction */
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&slots), sizeof(unsigned long long) * 4 * n));
    HIP_CHECK(hipMemset(slots, 0, sizeof(unsigned long long) * 4 * n));
    HIP_CHECK(hipDeviceSynchronize());
  }

  ~LocalGroup()
  {
    /* contexts of the group may still have waits on these events enqueued */
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (auto *set : {&ev_packed, &ev_pulled, &ev_reduced})
      for (auto &e : *set)
        if (e)
          (void)hipEventDestroy(e);
    if (slots)
      (void)hipFree(slots);
  }

  /* host-side: publish / await a generation counter (never touches the device) */
  void publish(std::vector<unsigned long> &counter, int rank, unsigned long value)
  {
    {
      std::lock_guard<std::mutex> lock(mtx);
      counter[rank] = value;
    }
    cv.notify_all();
  }
  void await(const std::vector<unsigned long> &counter, int rank, unsigned long value)
  {
    std::unique_lock<std::mutex> lock(mtx);
    cv.wait(lock, [&] { return aborted || counter[rank] >= value; });
    if (aborted && counter[rank] < value)
      throw HipError(RYUJIN_ERR_COMM, "in-process transport: another rank of the group failed");
  }
  void abort()
  {
    {
      std::lock_guard<std::mutex> lock(mtx);
      aborted = true;
    }
    cv.notify_all();
  }

  void barrier() /* host rendezvous for the host-valued reductions (state_integrals) */
  {
    std::unique_lock<std::mutex> lock(mtx);
    const unsigned long gen = generation;
    if (++arrived == n_ranks) {
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return aborted || generation != gen; });
      if (aborted && generation == gen)
        throw HipError(RYUJIN_ERR_COMM, "in-process transport: another rank of the group failed");
    }
  }
};

struct ryujin_hip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1, device = 0;
  LocalGroup *local = nullptr;
};

struct ryujin_hip_ctx {
  ryujin_hip_params params{};
  int dim = 0, K = 0, KP = 0, NB = 3, NPREC = 2;
  int device = 0;
  ryujin_hip_comm *comm = nullptr;
  hipStream_t stream = nullptr;      /* compute */
  hipStream_t comm_stream = nullptr; /* ghost exchange, overlapped with the interior rows */
  hipEvent_t ev_comm = nullptr;
  hipStream_t launch_stream = nullptr; /* the stream the sweep lambdas launch on (stream or comm_stream) */
  FusedSadd pending_sadd{0., 0., nullptr}; /* set by time_step for the SSPRK stages, consumed by step */
  bool pending_precompute = false; /* set by time_step: the next call on the new vector is prepare_state_vector */
  hipEvent_t ev_prev = nullptr; /* compute stream -> comm_stream: everything enqueued so far */
  hipEvent_t ev_exp = nullptr;  /* comm_stream -> compute stream: the latest export part (not its exchange) */
  bool comm_pending = false;    /* comm_stream holds work the compute stream has not joined */
  bool exp_pending = false;     /* ... of which an export part later kernels on the compute stream depend on */
  bool exchange_after_exp = false; /* an exchange was enqueued behind the latest export part (ev_exp misses it) */
  StepBegin pending_begin{}; /* set by step(), carried by its first sweep (step_begin) */
  uint32_t bc_fold_max_slices = kBcFoldMaxSlices;
  uint32_t resident_waves_step5 = kResidentWavesStep5, resident_waves_step6 = kResidentWavesStep6;
  bool interior_reads_ghosts = false; /* asymmetric stencil: every sweep joins the exchanges (no overlap) */
  uint32_t n_export_slices = 0;
  uint32_t bounds_stride = 0; /* SoA stride of the limiter bounds: covers the ghost range (dG reads bounds_j) */

  SellLayout L;
  DeviceMesh mesh{};
  EulerParams eparams{};
  ShallowWaterParams swparams{};
  EulerAeosParams aeosparams{};
  ScalarParams scparams{};
  DeviceBuffer<double> d_Z; /* initial_precomputed (bathymetry), shallow water only */

  template <typename E>
  const typename E::Params &eq_params() const
  {
    if constexpr (std::is_same<typename E::Params, EulerParams>::value)
      return eparams;
    else if constexpr (std::is_same<typename E::Params, EulerAeosParams>::value)
      return aeosparams;
    else if constexpr (std::is_same<typename E::Params, ScalarParams>::value)
      return scparams;
    else
      return swparams;
  }

  /* mesh arrays */
  DeviceBuffer<uint32_t> d_slice_off, d_cols, d_idx_t, d_lower_mask;
  DeviceBuffer<TileDesc> d_tiles; /* the tile map (host_layout.hpp); empty with debug_tile_map < 0 */
  DeviceBuffer<uint64_t> d_chain_loads; /* ... and the lanes of its chained tiles that fetch their node themselves */
  DeviceBuffer<uint16_t> d_row_len;
  DeviceBuffer<double> d_cij, d_mij, d_mi, d_mi_inv;
  bool dg = false; /* discontinuous ansatz */
  DeviceBuffer<double> d_incidence, d_minv, d_bounds_combined;

  /* boundary data */
  uint32_t n_bdry = 0, n_groups = 0;
  std::vector<uint32_t> bdry_perm; /* sorted entry -> original entry */
  DeviceBuffer<uint32_t> d_grp_start, d_b_i, d_bc_first;
  DeviceBuffer<unsigned long long> d_bc_mask; /* BcFold: boundary rows of every slice */
  DeviceBuffer<double> d_b_normal, d_dirichlet;
  DeviceBuffer<uint8_t> d_b_id;
  bool have_dirichlet = false, needs_dirichlet = false;
  /* ryujin_hip_state_download_prepared: the distinct boundary_map rows, packed on the device, scattered on the host */
  std::vector<uint32_t> h_bc_rows;
  DeviceBuffer<uint32_t> d_bc_rows;
  DeviceBuffer<double> d_bc_pack;
  double *h_bc_pack = nullptr; /* pinned */
  /* ryujin_hip_host_register: caller-owned arrays pinned in place */
  std::vector<const void *> registered_host;

  /* coupling boundary pairs */
  uint32_t n_pairs = 0;
  DeviceBuffer<uint32_t> d_p_i, d_p_j, d_p_pos;
  DeviceBuffer<double> d_p_cji;

  /* module-owned vectors and matrices */
  DeviceBuffer<double> d_alpha, d_bounds, d_r, d_dij, d_lij, d_lij_next, d_pij;
  DeviceBuffer<double> d_gamma; /* EulerAEOS: surrogate gamma_i of cycle 0, one double per node, for the stencil minimum of cycle 1 */
  DeviceBuffer<DeviceScalars> d_scalars;
  DeviceBuffer<double> d_integrals; /* ryujin_hip_state_integrals: block partials + result */
  DeviceScalars *h_scalars = nullptr; /* pinned */
  /* time-dependent Dirichlet data inside a device-resident RK step (ryujin_hip_time_step_fn): the tau of the
   * first stage is copied to the host as soon as it exists (behind step 3 of the first stage), the later stages'
   * boundary data is evaluated at t + c_s tau while the rest of the first stage runs */
  static constexpr int kMaxRkStages = 5;
  unsigned long long *h_tau_early = nullptr; /* pinned */
  hipEvent_t ev_tau = nullptr;
  bool want_tau_early = false;
  double *h_dirichlet_stage[kMaxRkStages] = {}; /* pinned staging, one per RK stage */

  struct State {
    DeviceBuffer<double> U, prec;
    DeviceBuffer<double> rrec; /* Euler, EulerAEOS, shallow water: per-node Riemann records (E::riemann_record) */
    bool used = false;
    /* prec and rrec of the owned rows belong to the U stored here, before boundary conditions (left behind by the
     * last sweep of the step that wrote U: FusedPrecompute); cleared by whatever else writes U */
    bool precomputed = false;
  };
  std::vector<std::unique_ptr<State>> states;

  /* exchange pattern */
  int n_nbr = 0;
  std::vector<int> nbr_rank;
  std::vector<uint32_t> send_off, recv_off, row_send_off;
  std::vector<uint64_t> row_recv_off; /* offsets into the ghost CSR region */
  DeviceBuffer<uint32_t> d_send_idx, d_row_send_pos;
  DeviceBuffer<double> d_send_buf;

  /* step 5 of the running step left V_i = U_i^low + sum_j lambda P_ij (k_lij_stage0, k_pij_lij): step 6 may take it */
  DeviceBuffer<double> d_V;
  bool stage0_V = false;
  /* the last step stored P_ij per slice (per_slice below): ryujin_hip_debug_fetch forms it from these operands for
   * the slices the sweeps left out */
  bool last_per_slice = false;
  bool last_tile_store = false; /* the last step stored P_ij per tile (plain kernels): debug_fetch forms the rest */
  Stage0Src last_s0{};
  /* SliceFlags (kernels_limiter.hpp), [n_slices] each; `unlimited` starts at 0 = "limited": the first update of a
   * context stores P_ij everywhere */
  DeviceBuffer<uint8_t> d_slice_unlimited, d_slice_first_stored, d_slice_todo;
  DeviceBuffer<uint32_t> d_slice_needed; /* SliceFlags::needed_tiles; starts as all ones: the first update stores every tile */
  DeviceBuffer<uint32_t> d_slice_deferred; /* SliceFlags::deferred */
  /* fractions of the (sampled) slices in which the first high-order sweep found a limited pair / whose P_ij step 5
   * stored, from the device counters at the latest host synchronisation (DeviceScalars::n_sampled_*); 1 until the
   * first measurement. Diagnostics only: nothing is decided from them. */
  double limited_fraction = 1., stored_fraction = 1.;
  unsigned int seen_sampled_slices = 0, seen_sampled_limited = 0, seen_sampled_stored = 0;
  unsigned int seen_sampled_tiles = 0, seen_sampled_tiles_stored = 0;
  unsigned int seen_sampled_tiles_needed = 0, seen_sampled_tiles_formed = 0;
  double tiles_needed_fraction = 1., tiles_formed_fraction = 0.; /* of the tiles: read by step 6; formed there */
  void update_limited_fraction()
  {
    const unsigned int d_tiles = h_scalars->n_sampled_tiles - seen_sampled_tiles;
    const unsigned int d_tiles_stored = h_scalars->n_sampled_tiles_stored - seen_sampled_tiles_stored;
    seen_sampled_tiles = h_scalars->n_sampled_tiles;
    seen_sampled_tiles_stored = h_scalars->n_sampled_tiles_stored;
    const unsigned int d_slices = h_scalars->n_sampled_slices - seen_sampled_slices;
    const unsigned int d_limited = h_scalars->n_sampled_limited - seen_sampled_limited;
    const unsigned int d_stored = h_scalars->n_sampled_stored - seen_sampled_stored;
    seen_sampled_slices = h_scalars->n_sampled_slices;
    seen_sampled_limited = h_scalars->n_sampled_limited;
    seen_sampled_stored = h_scalars->n_sampled_stored;
    if (d_slices != 0) {
      limited_fraction = (double)d_limited / (double)d_slices;
      stored_fraction = last_per_slice ? (double)d_stored / (double)d_slices : 1.;
    }
    const unsigned int d_needed = h_scalars->n_sampled_tiles_needed - seen_sampled_tiles_needed;
    const unsigned int d_formed = h_scalars->n_sampled_tiles_formed - seen_sampled_tiles_formed;
    seen_sampled_tiles_needed = h_scalars->n_sampled_tiles_needed;
    seen_sampled_tiles_formed = h_scalars->n_sampled_tiles_formed;
    if (last_tile_store && d_tiles != 0) { /* (of the tiles, by step 5; step 6 adds the few it has to form itself) */
      stored_fraction = (double)d_tiles_stored / (double)d_tiles;
      tiles_needed_fraction = (double)d_needed / (double)d_tiles;
      tiles_formed_fraction = (double)d_formed / (double)d_tiles;
    }
  }
  void ensure_pij()
  {
    if (d_pij.n == 0)
      d_pij.alloc(L.nnz_total * (size_t)K);
  }
  template <typename E>
  void store_pij_for_debug();

  unsigned n_restarts = 0, n_warnings = 0;
  unsigned long long n_exchanges = 0, n_allreduces = 0; /* ryujin_hip_exchange_info */

  /* profiling */
  bool timers_enabled = false;
  bool step2_split = false; /* step 2 ran as indicator kernel + Riemann kernel (event 8 recorded in between) */
  /* device-resident RK driver: per-step host synchronisation is deferred to the end of the RK step */
  bool deferred = false;
  int rk_stage = 0; /* selects the event set while deferred */
  double sweep_ms_accum[8] = {};
  unsigned sweep_updates_accum = 0;
  hipEvent_t ev_rk[5][9] = {};
  hipEvent_t ev[9] = {};
  hipEvent_t ev_user[2] = {};
  double sweep_ms[8] = {};

  ~ryujin_hip_ctx()
  {
    if (stream)
      (void)hipStreamSynchronize(stream);
    if (comm_stream) {
      (void)hipStreamSynchronize(comm_stream);
      (void)hipStreamDestroy(comm_stream);
    }
    if (ev_prev)
      (void)hipEventDestroy(ev_prev);
    if (ev_exp)
      (void)hipEventDestroy(ev_exp);
    if (ev_comm)
      (void)hipEventDestroy(ev_comm);
    for (auto &e : ev)
      if (e)
        (void)hipEventDestroy(e);
    for (auto &set : ev_rk)
      for (auto &e : set)
        if (e)
          (void)hipEventDestroy(e);
    for (auto &e : ev_user)
      if (e)
        (void)hipEventDestroy(e);
    if (h_scalars)
      (void)hipHostFree(h_scalars);
    if (h_bc_pack)
      (void)hipHostFree(h_bc_pack);
    for (const void *ptr : registered_host)
      (void)hipHostUnregister(const_cast<void *>(ptr));
    if (h_tau_early)
      (void)hipHostFree(h_tau_early);
    if (ev_tau)
      (void)hipEventDestroy(ev_tau);
    for (auto *b : h_dirichlet_stage)
      if (b)
        (void)hipHostFree(b);
    if (stream)
      (void)hipStreamDestroy(stream);
  }

  State &state(int h)
  {
    if (h < 0 || h >= (int)states.size() || !states[h] || !states[h]->used)
      throw HipError(RYUJIN_ERR_ARG, "invalid state handle " + std::to_string(h));
    return *states[h];
  }

  void create(const ryujin_hip_offline &o, const ryujin_hip_params &p, ryujin_hip_comm *c, int dev);
  void exchange_vector(double *v, int stride, bool after_split_sweep);
  void exchange_matrix(double *m, bool after_split_sweep);
  unsigned long local_exchanges = 0, local_reduces = 0; /* in-process transport: generation counters */
  void local_before_pack();
  void local_exchange(double *base, const std::vector<size_t> &send_offset,
                      const std::vector<size_t> &recv_offset, const std::vector<size_t> &recv_count);
  void allreduce_scalar(void *dev_ptr, int op, int count = 1);
  void wait_comm();
  void join_export();
  void finish();
  void begin_exchange(bool after_split_sweep);
  void end_exchange();
  template <typename F>
  void sweep(F &&launch);
  template <typename E>
  void prepare_state_vector(int h, const double *dirichlet);
  template <typename E>
  int step(int h_old, int stages, const int *h_stage, const double *w, int h_new, double tau_in,
           double tau_max_in, double *tau_out);
  template <typename E>
  int time_step(int scheme, int h_state, int n_tmp, const int *h_tmp, const double *dirichlet,
                double tau_max, int cfl_recovery, double cfl_min, double cfl_max, double *tau_out,
                double t = 0., ryujin_hip_dirichlet_fn dirichlet_fn = nullptr, void *dirichlet_user = nullptr);
  void mark(int k)
  {
    if (timers_enabled)
      HIP_CHECK(hipEventRecord(deferred ? ev_rk[rk_stage][k] : ev[k], stream));
  }
};

void ryujin_hip_ctx::create(const ryujin_hip_offline &o, const ryujin_hip_params &p,
                            ryujin_hip_comm *c, int dev)
{
  params = p;
  comm = c;
  device = dev;
  dim = p.dim;
  if (p.equation != RYUJIN_EQ_EULER && p.equation != RYUJIN_EQ_SHALLOW_WATER &&
      p.equation != RYUJIN_EQ_EULER_AEOS && p.equation != RYUJIN_EQ_SCALAR_CONSERVATION)
    throw HipError(RYUJIN_ERR_UNSUPPORTED, "unknown equation");
  if (p.equation == RYUJIN_EQ_SCALAR_CONSERVATION) {
    if (p.sc_flux < RYUJIN_FLUX_BURGERS || p.sc_flux > RYUJIN_FLUX_POLYNOMIAL)
      throw HipError(RYUJIN_ERR_UNSUPPORTED, "unknown flux");
    if (p.sc_flux == RYUJIN_FLUX_KPP && p.dim == 3)
      throw HipError(RYUJIN_ERR_UNSUPPORTED, "KPP is only defined in (1 or) 2 space dimensions");
    if (p.sc_random_entropies != 0) /* std::random_device in riemann_solver.template.h:93-107 */
      throw HipError(RYUJIN_ERR_UNSUPPORTED,
                     "scalar conservation: random entropies are not reproducible in the reference");
    for (uint32_t b = 0; b < o.n_bdry; ++b)
      if (o.b_id[b] == RYUJIN_BC_SLIP || o.b_id[b] == RYUJIN_BC_NO_SLIP || o.b_id[b] == RYUJIN_BC_DYNAMIC)
        throw HipError(RYUJIN_ERR_UNSUPPORTED, "slip, no-slip and dynamic boundary conditions are "
                                               "unavailable for scalar conservation equations");
  }
  if (p.equation == RYUJIN_EQ_EULER_AEOS) {
    if (p.eos < RYUJIN_EOS_POLYTROPIC_GAS || p.eos > RYUJIN_EOS_JONES_WILKINS_LEE)
      throw HipError(RYUJIN_ERR_UNSUPPORTED, "unknown equation of state");
    for (uint32_t b = 0; b < o.n_bdry; ++b)
      if (o.b_id[b] == RYUJIN_BC_DYNAMIC) /* __builtin_trap() in euler_aeos/hyperbolic_system.h:1337 */
        throw HipError(RYUJIN_ERR_UNSUPPORTED,
                       "euler aeos: dynamic boundary conditions are not implemented in the reference");
  }
  if (p.equation == RYUJIN_EQ_SHALLOW_WATER && p.dim == 3)
    throw HipError(RYUJIN_ERR_UNSUPPORTED, "the shallow water equations are defined for dim 1 and 2");
  if (dim < 1 || dim > 3)
    throw HipError(RYUJIN_ERR_ARG, "dim must be 1, 2 or 3");
  if (p.limiter_iterations < 0 || p.limiter_iterations > 2)
    throw HipError(RYUJIN_ERR_ARG, "The number of limiter iterations must be between [0,2]");
  K = p.equation == RYUJIN_EQ_SHALLOW_WATER ? dim + 1 : dim + 2;
  NB = p.equation == RYUJIN_EQ_EULER ? 3 : (p.equation == RYUJIN_EQ_EULER_AEOS ? 4 : 5);
  NPREC = p.equation == RYUJIN_EQ_EULER_AEOS ? 4 : 2;
  if (p.equation == RYUJIN_EQ_SCALAR_CONSERVATION) {
    K = 1;
    NB = 2;
    NPREC = 2 * dim;
  }
  KP = (K + 1) / 2 * 2;

  HIP_CHECK(hipSetDevice(device));
  HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
  launch_stream = stream;
  create_device_event(&ev_prev, p.system_scope_events != 0);
  create_device_event(&ev_exp, p.system_scope_events != 0);
  create_device_event(&ev_comm, p.system_scope_events != 0);
  for (auto &e : ev)
    HIP_CHECK(hipEventCreate(&e));
  for (auto &set : ev_rk)
    for (auto &e : set)
      HIP_CHECK(hipEventCreate(&e));
  for (auto &e : ev_user)
    HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h_scalars), sizeof(DeviceScalars)));

  eparams = make_euler_params(p);
  aeosparams = make_aeos_params(p);

  scparams.flux = p.sc_flux;
  scparams.use_greedy_wavespeed = p.sc_use_greedy_wavespeed != 0;
  scparams.use_averaged_entropy = p.sc_use_averaged_entropy != 0;
  for (int d = 0; d < 3; ++d)
    for (int n = 0; n < 4; ++n)
      scparams.poly[d][n] = p.sc_flux_polynomial[d][n];
  /* flux.h:33-34; the "function" flux reads its own parameter (flux_function.h:40-44) */
  scparams.delta = p.sc_flux == RYUJIN_FLUX_POLYNOMIAL ? p.sc_derivative_approximation_delta
                                                       : 1.e4 * std::numeric_limits<double>::epsilon();
  scparams.evc_factor = p.indicator_evc_factor;
  scparams.lim_relaxation_factor = p.limiter_relaxation_factor;

  swparams = make_sw_params(p);
  if (p.equation == RYUJIN_EQ_SHALLOW_WATER) {
    if (o.initial_precomputed)
      d_Z.upload(o.initial_precomputed, o.n_relevant);
    else
      d_Z.alloc(o.n_relevant); /* flat bed */
  }

  /* ---- stencil ------------------------------------------------------------- */
  L.build(o);
  d_slice_off.upload(L.slice_off);
  d_row_len.upload(L.row_len);
  d_cols.upload(L.cols);
  d_idx_t.upload(L.idx_t);
  if (p.debug_tile_map >= 0) { /* (3-D sweeps keep streaming the explicit indices, tile_map_pays(); they use the chain codes) */
    L.build_tiles();
    d_tiles.upload(L.tiles);
    d_chain_loads.upload(L.chain_loads);
  }
  {
    /* bit c of lower_mask[row] <=> column c of the row lies below the diagonal (cols < row) */
    std::vector<uint32_t> lower_mask(L.rows_padded, 0u);
    if (L.max_row_len <= 32)
      parallel_chunks(L.n_owned, [&](const uint64_t i0, const uint64_t i1) {
        for (uint32_t i = (uint32_t)i0; i < (uint32_t)i1; ++i)
          for (uint32_t c = 1; c < L.row_len[i]; ++c)
            if (L.cols[L.pos(i, c)] < i)
              lower_mask[i] |= 1u << c;
      });
    d_lower_mask.upload(lower_mask);
  }
  {
    const auto cij = L.scatter(o, o.cij, (uint32_t)dim);
    d_cij.upload(cij);
    const auto mij = L.scatter(o, o.mij, 1);
    d_mij.upload(mij);
    dg = o.discontinuous_ansatz != 0;
    if (dg) {
      /* Scalar conservation: refused. The branch is written (k_low_order_sc<DIM, HAS_STAGES, true>,
       * k_bounds_combine_minmax<2, 0b10>, k_pij_lij<E, true>), but the reference's own scalar Riemann solver cannot
       * run on a dG stencil: the entries between DoFs of face-neighbour cells that do not lie on the shared face have
       * c_ij = 0 exactly, n_ij = c_ij / |c_ij| is 0/0, and |f_i.n - f_j.n| / max(|u_i - u_j|, 2 delta) -- taken
       * through std::max, which returns its first argument when the comparison with a NaN fails
       * (scalar_conservation/riemann_solver.template.h:63,95-96) -- stays NaN: d_ij = 0 * NaN
       * (tests/test_dg_q1.py::test_scalar_conservation_on_a_dg_stencil_is_nan_in_the_reference_formulas). */
      if (p.equation == RYUJIN_EQ_SCALAR_CONSERVATION)
        throw HipError(RYUJIN_ERR_UNSUPPORTED,
                       "discontinuous ansatz with scalar conservation equations: the reference's Riemann solver is 0/0 "
                       "on the structural zeros of a dG stencil");
      if (!o.incidence || !o.mass_matrix_inverse)
        throw HipError(RYUJIN_ERR_ARG, "discontinuous ansatz without incidence / inverse mass matrix");
      d_incidence.upload(L.scatter(o, o.incidence, 1));
      d_minv.upload(L.scatter(o, o.mass_matrix_inverse, 1));
    }

    /* coupling boundary pairs: position of (i,col_idx) and the value of c_ji */
    n_pairs = o.n_pairs;
    std::vector<uint32_t> p_pos(n_pairs);
    std::vector<double> p_cji((size_t)n_pairs * dim);
    for (uint32_t q = 0; q < n_pairs; ++q) {
      const uint64_t pos = L.pos(o.p_i[q], o.p_col[q]);
      if (L.cols[pos] != o.p_j[q])
        throw HipError(RYUJIN_ERR_ARG, "coupling_boundary_pairs do not match the sparsity pattern");
      p_pos[q] = (uint32_t)pos;
      const uint64_t pos_t = L.idx_t[pos];
      for (int d = 0; d < dim; ++d)
        p_cji[(size_t)q * dim + d] = cij[L.comp_pos(pos_t, (uint32_t)dim, (uint32_t)d)];
    }
    d_p_i.upload(o.p_i, n_pairs);
    d_p_j.upload(o.p_j, n_pairs);
    d_p_pos.upload(p_pos);
    d_p_cji.upload(p_cji);
  }
  d_mi.upload(o.mi, o.n_relevant);
  d_mi_inv.upload(o.mi_inv, o.n_relevant);

  mesh.n_owned = L.n_owned;
  mesh.n_relevant = L.n_relevant;
  mesh.n_slices = L.n_slices;
  bounds_stride = (std::max<uint32_t>(L.rows_padded, L.n_relevant) + 63u) / 64u * 64u;
  mesh.bounds_stride = bounds_stride;
  mesh.slice_begin = 0;
  mesh.slice_end = L.n_slices;
  /* rows [0, n_export) are the ones other ranks hold as ghosts (offline_data.template.h:213-249) */
  n_export_slices = std::min<uint32_t>(L.n_slices, (o.n_export + kWave - 1) / kWave);
  /* The overlap of the exchanges with the interior rows rests on: only export rows couple to ghost columns
   * (true for a symmetric stencil: if i sees the ghost j, the owner of j sees i). Checked, not assumed. */
  if (L.n_owned > n_export_slices * kWave) {
    const uint32_t first = n_export_slices * kWave;
    std::atomic<bool> found{false};
    parallel_chunks(L.n_owned - first, [&](const uint64_t i0, const uint64_t i1) {
      for (uint32_t i = first + (uint32_t)i0; i < first + (uint32_t)i1 && !found.load(std::memory_order_relaxed); ++i)
        for (uint32_t c = 0; c < L.row_len[i]; ++c)
          if (L.cols[L.pos(i, c)] >= L.n_owned) {
            found.store(true, std::memory_order_relaxed);
            break;
          }
    });
    interior_reads_ghosts = found.load();
  }
  /* test hooks (ryujin_hip_params::debug_*; tests/test_gpu_parity.py runs the partitioned cases through both
   * branches of each): force the fallback choreography / move the mesh size below which boundary conditions
   * ride on the pre-pass / small meshes run the kernels of the large ones */
  interior_reads_ghosts = interior_reads_ghosts || p.debug_join_exchanges != 0;
  if (p.debug_bc_fold_max_slices != 0)
    bc_fold_max_slices = p.debug_bc_fold_max_slices < 0 ? 0u : (uint32_t)p.debug_bc_fold_max_slices;
  if (p.debug_no_small_mesh_split != 0)
    resident_waves_step5 = resident_waves_step6 = 0;
  mesh.slice_off = d_slice_off.ptr;
  mesh.row_len = d_row_len.ptr;
  mesh.cols = d_cols.ptr;
  mesh.idx_t = d_idx_t.ptr;
  mesh.tiles = d_tiles.n != 0 ? d_tiles.ptr : nullptr;
  mesh.chain_loads = d_chain_loads.n != 0 ? d_chain_loads.ptr : nullptr;
  mesh.tail_queue_columns = std::max<uint32_t>(1u, std::min<uint32_t>(63u, L.max_row_len - 1u));
  /* stacked blocks (row_context(), kernels_euler.hpp): debug_band_stride < 0 off, > 0 that many slices, 0 from the mesh */
  mesh.band_stride = 1;
  if (p.debug_band_stride > 0)
    mesh.band_stride = (uint32_t)p.debug_band_stride;
  else if (p.debug_band_stride == 0 && RYUJIN_BAND_DEFAULT && dim == 2) {
    const uint32_t stride = L.lattice_stride(dim);
    const uint32_t G = (stride + kWave / 2) / kWave;
    if (G >= 2 && (uint64_t)G * kWavesPerBlock * 4 <= L.n_slices)
      mesh.band_stride = G;
  }
  /* XCD-local block ranges (row_context()): debug_xcd_chunk < 0 off, > 0 that many blocks per XCD and chunk */
  mesh.xcd_chunk = 0;
  if (p.debug_xcd_chunk > 0)
    mesh.xcd_chunk = (uint32_t)p.debug_xcd_chunk;
  else if (p.debug_xcd_chunk == 0 && dim == 3)
    mesh.xcd_chunk = RYUJIN_XCD_CHUNK_3D;
  mesh.cij = d_cij.ptr;
  mesh.mij = d_mij.ptr;
  mesh.incidence = dg ? d_incidence.ptr : nullptr;
  mesh.mass_matrix_inverse = dg ? d_minv.ptr : nullptr;
  mesh.mi = d_mi.ptr;
  mesh.mi_inv = d_mi_inv.ptr;
  mesh.measure_of_omega_inverse = 1. / o.measure_of_omega;

  /* ---- boundary map: group entries by DoF, keep the original order inside a group ---- */
  n_bdry = o.n_bdry;
  bdry_perm.resize(n_bdry);
  std::iota(bdry_perm.begin(), bdry_perm.end(), 0u);
  std::stable_sort(bdry_perm.begin(), bdry_perm.end(),
                   [&](uint32_t a, uint32_t b) { return o.b_i[a] < o.b_i[b]; });
  {
    std::vector<uint32_t> b_i(n_bdry), grp_start;
    std::vector<double> b_normal((size_t)n_bdry * dim);
    std::vector<uint8_t> b_id(n_bdry);
    for (uint32_t e = 0; e < n_bdry; ++e) {
      const uint32_t src = bdry_perm[e];
      b_i[e] = o.b_i[src];
      if (b_i[e] >= o.n_owned)
        throw HipError(RYUJIN_ERR_ARG, "boundary_map entry refers to a non-owned DoF");
      b_id[e] = o.b_id[src];
      /* ids whose boundary condition reads the Dirichlet state (hyperbolic_system.h:1099-1159) */
      if (b_id[e] == RYUJIN_BC_DIRICHLET || b_id[e] == RYUJIN_BC_DYNAMIC ||
          b_id[e] == RYUJIN_BC_DIRICHLET_MOMENTUM)
        needs_dirichlet = true;
      for (int d = 0; d < dim; ++d)
        b_normal[(size_t)e * dim + d] = o.b_normal[(size_t)src * dim + d];
      if (e == 0 || b_i[e] != b_i[e - 1])
        grp_start.push_back(e);
    }
    n_groups = (uint32_t)grp_start.size();
    h_bc_rows.clear();
    for (const uint32_t e : grp_start)
      h_bc_rows.push_back(b_i[e]);
    /* which rows of a slice are boundary DoFs, and the group of the first one (BcFold) */
    std::vector<unsigned long long> bc_mask(L.n_slices, 0ull);
    std::vector<uint32_t> bc_first(L.n_slices, 0u);
    for (uint32_t g = n_groups; g-- > 0;) {
      const uint32_t row = b_i[grp_start[g]];
      bc_mask[row / kWave] |= 1ull << (row % kWave);
      bc_first[row / kWave] = g; /* descending loop: the smallest group of the slice wins */
    }
    d_bc_mask.upload(bc_mask);
    d_bc_first.upload(bc_first);
    grp_start.push_back(n_bdry);
    d_grp_start.upload(grp_start);
    d_b_i.upload(b_i);
    d_b_normal.upload(b_normal);
    d_b_id.upload(b_id);
    d_dirichlet.alloc((size_t)n_bdry * K);
  }

  /* ---- module-owned storage (prepare(): hyperbolic_module.template.h:52-86) ---- */
  d_alpha.alloc(L.n_relevant);
  if (params.equation == RYUJIN_EQ_EULER_AEOS)
    d_gamma.alloc(L.n_relevant);
  d_bounds.alloc((size_t)NB * bounds_stride);
  if (dg)
    d_bounds_combined.alloc((size_t)NB * bounds_stride);
  d_r.alloc((size_t)L.n_relevant * KP);
  d_dij.alloc(L.nnz_total);
  d_lij.alloc(L.nnz_total);
  d_lij_next.alloc(L.nnz_total);
  /* (p_ij is allocated by the first step: ensure_pij) */
  d_scalars.alloc(1);

  /* ---- exchange pattern ---- */
  n_nbr = o.n_nbr;
  if (n_nbr > 0 && !comm)
    throw HipError(RYUJIN_ERR_COMM, "offline data has neighbour ranks but no communicator was given");
  if (n_nbr > 0) {
    nbr_rank.assign(o.nbr_rank, o.nbr_rank + n_nbr);
    send_off.assign(o.send_off, o.send_off + n_nbr + 1);
    recv_off.assign(o.recv_off, o.recv_off + n_nbr + 1);
    row_send_off.assign(o.row_send_off, o.row_send_off + n_nbr + 1);
    d_send_idx.upload(o.send_idx, send_off[n_nbr]);
    std::vector<uint32_t> row_send_pos(row_send_off[n_nbr]);
    for (uint32_t e = 0; e < row_send_off[n_nbr]; ++e)
      row_send_pos[e] = (uint32_t)L.pos(o.row_send_row[e], o.row_send_col[e]);
    d_row_send_pos.upload(row_send_pos);
    row_recv_off.resize((size_t)n_nbr + 1);
    for (int q = 0; q <= n_nbr; ++q)
      row_recv_off[q] = L.ghost_ptr[recv_off[q] - L.n_owned];
    const size_t buf =
        std::max<size_t>((size_t)send_off[n_nbr] * std::max(KP, NPREC), row_send_off[n_nbr]);
    d_send_buf.alloc(buf);
  }
  HIP_CHECK(hipStreamSynchronize(stream));
}

/* ---- stream choreography of the ghost exchange ------------------------------------------------
 * Two streams. Every sweep runs as two launches: the few export slices (the rows other ranks hold as
 * ghosts = the only rows that couple to ghost columns) on comm_stream, the interior slices on the compute
 * stream. Exchanges (pack + RCCL send/recv, or the in-process copies) follow the export launch that
 * produced their data in stream order on comm_stream -- the reference's SynchronizationDispatch idea
 * (source/openmp.h:141-183), taken one step further: no SWEEP on the compute stream ever waits for an
 * exchange (the scalar all-reduces do, see allreduce_scalar). Dependencies:
 *   export part N      needs interior parts <= N-1 (ev_prev: compute -> comm), export parts and
 *                      exchanges <= N-1 (stream order on comm_stream)
 *   interior part N    needs export parts <= N-1 (ev_exp: comm -> compute, recorded BEFORE exchange N-1),
 *                      interior parts <= N-1 (stream order); never ghost data
 *   exchange N         needs export part N only (stream order)
 * so an exchange has the whole interior launch of the NEXT sweep as well to hide behind, and a short sweep
 * (the pre-pass, 30 us in 2-D) no longer exposes the latency of the exchange in front of it. Kernels outside
 * sweep() that touch export rows (boundary conditions, boundary d_ij) call join_export(); whoever reads the
 * ghost range on the compute stream or on the host, and the scalar all-reduces (one communicator: RCCL orders
 * them behind the exchanges anyway), call wait_comm().
 * (Round 2 started with a third stream for the export part and a join of every exchange in front of the next
 * sweep: +11 % per update on a middle rank in 2-D without any network in the loop, profiles/r02k_*.) */
void ryujin_hip_ctx::wait_comm()
{
  if (comm_pending) {
    HIP_CHECK(hipEventRecord(ev_comm, comm_stream)); /* the tail of comm_stream as of now */
    HIP_CHECK(hipStreamWaitEvent(stream, ev_comm, 0));
    comm_pending = false;
    exp_pending = false;
    exchange_after_exp = false;
  }
}

void ryujin_hip_ctx::join_export()
{
  if (interior_reads_ghosts) {
    wait_comm();
    return;
  }
  if (exp_pending) {
    HIP_CHECK(hipStreamWaitEvent(stream, ev_exp, 0));
    exp_pending = false;
  }
}

/* dynamic LDS of k_pij_lij: the queue of undecided pairs, (widest row - 1) columns of 64 two-byte entries per wave */
static inline size_t tail_queue_bytes(const DeviceMesh &mm)
{
  return (size_t)kWavesPerBlock * mm.tail_queue_columns * 64u * sizeof(uint16_t);
}

void ryujin_hip_ctx::finish()
{
  wait_comm();
  HIP_CHECK(hipStreamSynchronize(stream));
}

/* exchange of data the compute stream produced outside a sweep (the state vector behind the boundary
 * conditions); after a sweep the export part already sits on comm_stream */
void ryujin_hip_ctx::begin_exchange(bool after_split_sweep)
{
  if (after_split_sweep)
    return;
  HIP_CHECK(hipEventRecord(ev_prev, stream)); /* everything enqueued so far */
  HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_prev, 0));
}

void ryujin_hip_ctx::end_exchange()
{
  ++n_exchanges;
  comm_pending = true;
  exchange_after_exp = true;
}

template <typename F>
void ryujin_hip_ctx::sweep(F &&launch)
{
  auto run = [&](uint32_t s0, uint32_t s1, hipStream_t on) {
    if (s1 <= s0)
      return;
    DeviceMesh mm = mesh;
    mm.begin = pending_begin;
    mm.slice_begin = s0;
    mm.slice_end = s1;
    /* one wave per slice, 4 slices per block (waves beyond slice_end return at once) */
    const dim3 grid((s1 - s0 + kWavesPerBlock - 1) / kWavesPerBlock);
    launch_stream = on;
    launch(mm, grid);
    launch_stream = stream;
  };
  struct Consume { /* the start of a step rides on one sweep only */
    StepBegin &b;
    ~Consume() { b = StepBegin{}; }
  } consume{pending_begin};
  if (n_nbr == 0) {
    run(0, L.n_slices, stream);
    return;
  }
  join_export();
  HIP_CHECK(hipEventRecord(ev_prev, stream));
  HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_prev, 0));
  run(0, n_export_slices, comm_stream);
  HIP_CHECK(hipEventRecord(ev_exp, comm_stream));
  exp_pending = true;
  comm_pending = true;
  exchange_after_exp = false; /* stream order: ev_exp covers every exchange enqueued so far */
  run(n_export_slices, L.n_slices, stream);
}

/* in-process transport, part 1 (before the pack kernel): the send buffer may only be overwritten once
 * every neighbour has pulled the segments of the previous exchange */
void ryujin_hip_ctx::local_before_pack()
{
  LocalGroup &g = *comm->local;
  const unsigned long x = local_exchanges;
  if (x == 0)
    return;
  for (int q = 0; q < n_nbr; ++q) {
    const int peer = g.loopback ? comm->rank : nbr_rank[q];
    g.await(g.pulled, peer, x);
    HIP_CHECK(hipStreamWaitEvent(comm_stream,
                                 g.ev_pulled[(size_t)peer * LocalGroup::kRing + (x - 1) % LocalGroup::kRing], 0));
  }
}

/* part 2 (behind the pack kernel): publish the per-neighbour segments of the send buffer behind an event,
 * pull the neighbours' segments behind theirs. Stream ordered throughout, see LocalGroup. */
void ryujin_hip_ctx::local_exchange(double *base, const std::vector<size_t> &send_offset,
                                    const std::vector<size_t> &recv_offset,
                                    const std::vector<size_t> &recv_count)
{
  LocalGroup &g = *comm->local;
  const unsigned long x = local_exchanges;
  const size_t slot = x % LocalGroup::kRing;
  const int me = comm->rank;
  for (int q = 0; q < n_nbr; ++q)
    g.mail[me][nbr_rank[q]] = d_send_buf.ptr + send_offset[q];
  HIP_CHECK(hipEventRecord(g.ev_packed[(size_t)me * LocalGroup::kRing + slot], comm_stream));
  g.publish(g.packed, me, x + 1);
  for (int q = 0; q < n_nbr; ++q) {
    const int peer = g.loopback ? me : nbr_rank[q];
    g.await(g.packed, peer, x + 1);
    HIP_CHECK(hipStreamWaitEvent(comm_stream, g.ev_packed[(size_t)peer * LocalGroup::kRing + slot], 0));
    /* loopback: what the neighbour on the other side would have received from me */
    const double *src = g.loopback ? g.mail[me][nbr_rank[n_nbr - 1 - q]] : g.mail[nbr_rank[q]][me];
    if (recv_count[q])
      HIP_CHECK(hipMemcpyAsync(base + recv_offset[q], src, recv_count[q] * sizeof(double),
                               hipMemcpyDeviceToDevice, comm_stream));
  }
  HIP_CHECK(hipEventRecord(g.ev_pulled[(size_t)me * LocalGroup::kRing + slot], comm_stream));
  g.publish(g.pulled, me, x + 1);
  ++local_exchanges;
}

/* 1-element all-reduce on a device scalar; op: 0 = min (double), 1 = max (int) */
void ryujin_hip_ctx::allreduce_scalar(void *dev_ptr, int op, int count)
{
  if (!comm || comm->n_ranks <= 1)
    return;
  /* The scalar is written by export parts as well, and ONE communicator serves both streams: RCCL orders the
   * operations of a communicator in issue order whatever stream they are enqueued on, so an all-reduce on the
   * compute stream implicitly waits for the point-to-point exchanges issued before it. The join is therefore
   * made explicit -- for BOTH transports, so that the in-process transport the partitioned parity tests (and
   * scripts/overhead_loopback.py) run has exactly the dependency graph of the RCCL leg. This is the one place
   * where the compute stream waits for an exchange: the exchange of alpha in front of the tau_max reduction
   * of the first stage, enqueued behind the export part of step 2 and long finished when step 3 retires (two
   * interior launches later), and whatever is in flight at the end of a step / of a Runge-Kutta step. */
  wait_comm();
  ++n_allreduces;
  if (!comm->local) {
    if (op == 0)
      NCCL_CHECK(ncclAllReduce(dev_ptr, dev_ptr, 1, ncclDouble, ncclMin, comm->comm, stream));
    else
      NCCL_CHECK(ncclAllReduce(dev_ptr, dev_ptr, count, ncclInt, ncclMax, comm->comm, stream));
    return;
  }
  /* in-process transport: slot write -> event -> wait for everybody's event -> slot reduce, all on the
   * compute stream (see LocalGroup) */
  LocalGroup &g = *comm->local;
  const unsigned long a = local_reduces;
  const size_t par = a & 1;
  const int me = comm->rank;
  unsigned long long *slots = g.slots + par * (size_t)g.n_ranks * 2;
  hipLaunchKernelGGL(k_slot_write, dim3(1), dim3(1), 0, stream, dev_ptr, op, count, slots + (size_t)me * 2);
  HIP_CHECK(hipEventRecord(g.ev_reduced[(size_t)me * LocalGroup::kRing + par], stream));
  g.publish(g.reduced, me, a + 1);
  for (int q = 0; q < g.n_ranks && !g.loopback; ++q) {
    if (q == me)
      continue;
    g.await(g.reduced, q, a + 1);
    HIP_CHECK(hipStreamWaitEvent(stream, g.ev_reduced[(size_t)q * LocalGroup::kRing + par], 0));
  }
  if (g.loopback) /* reduce over my own slot only */
    hipLaunchKernelGGL(k_slot_reduce, dim3(1), dim3(1), 0, stream, slots + (size_t)me * 2, 1, op, count, dev_ptr);
  else
    hipLaunchKernelGGL(k_slot_reduce, dim3(1), dim3(1), 0, stream, slots, g.n_ranks, op, count, dev_ptr);
  ++local_reduces;
}

void ryujin_hip_ctx::exchange_vector(double *v, int stride, bool after_split_sweep)
{
  if (n_nbr == 0)
    return;
  begin_exchange(after_split_sweep);
  if (comm->local)
    local_before_pack();
  const uint32_t n_send = send_off[n_nbr];
  hipLaunchKernelGGL(k_pack_vector, dim3(grid_for((size_t)n_send * stride)), dim3(kBlock), 0, comm_stream,
                     n_send, d_send_idx.ptr, stride, v, d_send_buf.ptr);
  if (comm->local) {
    std::vector<size_t> off(n_nbr), dst(n_nbr), rcnt(n_nbr);
    for (int q = 0; q < n_nbr; ++q) {
      off[q] = (size_t)send_off[q] * stride;
      dst[q] = (size_t)recv_off[q] * stride;
      rcnt[q] = (size_t)(recv_off[q + 1] - recv_off[q]) * stride;
    }
    local_exchange(v, off, dst, rcnt);
    end_exchange();
    return;
  }
  NCCL_CHECK(ncclGroupStart());
  for (int q = 0; q < n_nbr; ++q) {
    NCCL_CHECK(ncclSend(d_send_buf.ptr + (size_t)send_off[q] * stride,
                        (size_t)(send_off[q + 1] - send_off[q]) * stride, ncclDouble, nbr_rank[q],
                        comm->comm, comm_stream));
    NCCL_CHECK(ncclRecv(v + (size_t)recv_off[q] * stride,
                        (size_t)(recv_off[q + 1] - recv_off[q]) * stride, ncclDouble, nbr_rank[q],
                        comm->comm, comm_stream));
  }
  NCCL_CHECK(ncclGroupEnd());
  end_exchange();
}

void ryujin_hip_ctx::exchange_matrix(double *m, bool after_split_sweep)
{
  if (n_nbr == 0)
    return;
  begin_exchange(after_split_sweep);
  if (comm->local)
    local_before_pack();
  const uint32_t n_send = row_send_off[n_nbr];
  hipLaunchKernelGGL(k_pack_matrix, dim3(grid_for(n_send)), dim3(kBlock), 0, comm_stream, n_send,
                     d_row_send_pos.ptr, m, d_send_buf.ptr);
  if (comm->local) {
    std::vector<size_t> off(n_nbr), dst(n_nbr), rcnt(n_nbr);
    for (int q = 0; q < n_nbr; ++q) {
      off[q] = row_send_off[q];
      dst[q] = L.nnz_sell + row_recv_off[q];
      rcnt[q] = row_recv_off[q + 1] - row_recv_off[q];
    }
    local_exchange(m, off, dst, rcnt);
    end_exchange();
    return;
  }
  NCCL_CHECK(ncclGroupStart());
  for (int q = 0; q < n_nbr; ++q) {
    NCCL_CHECK(ncclSend(d_send_buf.ptr + row_send_off[q], row_send_off[q + 1] - row_send_off[q],
                        ncclDouble, nbr_rank[q], comm->comm, comm_stream));
    NCCL_CHECK(ncclRecv(m + L.nnz_sell + row_recv_off[q], row_recv_off[q + 1] - row_recv_off[q],
                        ncclDouble, nbr_rank[q], comm->comm, comm_stream));
  }
  NCCL_CHECK(ncclGroupEnd());
  end_exchange();
}

template <typename E>
void ryujin_hip_ctx::store_pij_for_debug()
{
  finish();
  DeviceMesh mm = mesh;
  mm.begin = StepBegin{};
  mm.slice_begin = 0;
  mm.slice_end = L.n_slices;
  const dim3 grid((L.n_slices + kWavesPerBlock - 1) / kWavesPerBlock), block(kBlock);
  hipLaunchKernelGGL(k_pij_stage0_store<E>, grid, block, 0, stream, mm, last_s0, d_pij.ptr,
                     last_per_slice ? (const uint8_t *)d_slice_first_stored.ptr : (const uint8_t *)nullptr);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(stream));
}

template <typename E>
void ryujin_hip_ctx::prepare_state_vector(int h, const double *dirichlet)
{
  const auto &eparams = eq_params<E>(); /* shadows the member: the equation's parameter block */
  State &s = state(h);
  const dim3 block(kBlock);
  if (dirichlet && n_bdry) {
    /* permute into the grouped order, then upload */
    if (deferred) {
      /* inside a device-resident RK step: a pinned staging buffer per stage, no host synchronisation (a buffer
       * is rewritten in the next RK step at the earliest, i.e. behind the end-of-step synchronisation). The
       * copy is ordered on the compute stream behind every kernel of the previous stage that read the data. */
      double *&buf = h_dirichlet_stage[rk_stage];
      if (!buf)
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&buf), (size_t)n_bdry * K * sizeof(double)));
      for (uint32_t e = 0; e < n_bdry; ++e)
        std::memcpy(&buf[(size_t)e * K], &dirichlet[(size_t)bdry_perm[e] * K], sizeof(double) * K);
      join_export(); /* the export part of the previous pre-pass may have applied boundary conditions */
      HIP_CHECK(hipMemcpyAsync(d_dirichlet.ptr, buf, (size_t)n_bdry * K * sizeof(double), hipMemcpyHostToDevice,
                               stream));
    } else {
      std::vector<double> tmp((size_t)n_bdry * K);
      for (uint32_t e = 0; e < n_bdry; ++e)
        std::memcpy(&tmp[(size_t)e * K], &dirichlet[(size_t)bdry_perm[e] * K], sizeof(double) * K);
      HIP_CHECK(hipMemcpyAsync(d_dirichlet.ptr, tmp.data(), tmp.size() * sizeof(double),
                               hipMemcpyHostToDevice, stream));
      HIP_CHECK(hipStreamSynchronize(stream)); /* tmp goes out of scope */
    }
    have_dirichlet = true;
  }
  if (needs_dirichlet && !have_dirichlet)
    throw HipError(RYUJIN_ERR_ARG, "prepare_state_vector: the boundary map holds dirichlet / dynamic / "
                                   "dirichlet_momentum ids but no Dirichlet data was ever passed");
  /* Boundary conditions (:102-146). Small meshes: applied by the first pre-pass sweep itself (apply_bc_row; a
   * launch less per update). Boundary rows may be export rows -- read by the pack kernel of an exchange of this
   * vector that is still in flight (two calls in a row): the export part runs behind it in stream order on
   * comm_stream, the interior part never writes them. Large meshes: a launch of its own in front of the sweep
   * (the streaming pre-pass kernel keeps its occupancy); it writes export rows from the compute stream and
   * therefore joins whatever comm_stream still holds for them. */
  const BcFold bc{n_groups ? d_bc_mask.ptr : nullptr, d_bc_first.ptr, d_grp_start.ptr, d_b_normal.ptr,
                  d_b_id.ptr,   d_dirichlet.ptr};
  const bool fold_bc = L.n_slices <= bc_fold_max_slices;
  /* the step that wrote this vector left its precomputed values and Riemann records behind (FusedPrecompute):
   * no pre-pass sweep, the boundary rows are redone behind the boundary conditions */
  bool have_records = false;
  if constexpr (E::kFusablePrecompute)
    have_records = s.precomputed && !fold_bc;
  s.precomputed = false;
  if (!fold_bc && n_groups) {
    if (exchange_after_exp)
      wait_comm();
    else
      join_export();
    if constexpr (E::kFusablePrecompute) {
      if (have_records)
        hipLaunchKernelGGL(k_apply_bc_records<E>, dim3(grid_for(n_groups)), block, 0, stream, eparams, n_groups,
                           d_b_i.ptr, d_row_len.ptr, bc, s.U.ptr, s.prec.ptr, s.rrec.ptr);
    }
    if (!have_records)
      hipLaunchKernelGGL(k_apply_bc<E>, dim3(grid_for(n_groups)), block, 0, stream, eparams, n_groups, d_b_i.ptr,
                         bc, s.U.ptr);
  }
  /* U.update_ghost_values(), :148, is enqueued BEHIND the export part of the first pre-pass sweep, which
   * reads owned states only: the interior part of step 2 then waits for that export part alone, and the
   * exchange of U hides behind the interior parts of the pre-pass and of step 2 */
  if constexpr (std::is_same<typename E::Params, EulerAeosParams>::value) {
    /* n_precomputation_cycles = 2 (euler_aeos/hyperbolic_system.h:433), ghost update after each */
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      if (fold_bc)
        hipLaunchKernelGGL((k_precompute_aeos0<E::DIMENSION, true>), grid, block, 0, launch_stream, eparams, mm, bc, s.U.ptr,
                         s.prec.ptr, s.rrec.ptr, d_gamma.ptr);
      else
        hipLaunchKernelGGL((k_precompute_aeos0<E::DIMENSION, false>), grid, block, 0, launch_stream, eparams, mm, bc, s.U.ptr,
                         s.prec.ptr, s.rrec.ptr, d_gamma.ptr);
    });
    exchange_vector(s.U.ptr, KP, true);
    exchange_vector(s.prec.ptr, 4, true);
    if (L.n_relevant > L.n_owned) {
      /* Riemann records of the ghost rows from the exchanged (U_j, p_j): on comm_stream behind the two exchanges */
      hipLaunchKernelGGL(k_ghost_records_aeos<E::DIMENSION>, dim3(grid_for(L.n_relevant - L.n_owned)), block, 0,
                         n_nbr ? comm_stream : stream, eparams, L.n_owned, L.n_relevant, s.U.ptr, s.prec.ptr,
                         s.rrec.ptr, d_gamma.ptr);
      if (n_nbr) {
        comm_pending = true;
        exchange_after_exp = true;
      }
    }
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      hipLaunchKernelGGL(k_precompute_aeos1<E::DIMENSION>, grid, block, 0, launch_stream, eparams, mm, s.U.ptr,
                         d_gamma.ptr, s.prec.ptr, s.prec.ptr);
    });
    exchange_vector(s.prec.ptr, 4, true);
  } else if constexpr (std::is_same<typename E::Params, ScalarParams>::value) {
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      if (fold_bc)
        hipLaunchKernelGGL((k_precompute_sc<E::DIMENSION, true>), grid, block, 0, launch_stream, eparams, mm, bc, s.U.ptr,
                         s.prec.ptr);
      else
        hipLaunchKernelGGL((k_precompute_sc<E::DIMENSION, false>), grid, block, 0, launch_stream, eparams, mm, bc, s.U.ptr,
                         s.prec.ptr);
    });
    exchange_vector(s.U.ptr, KP, true);
    exchange_vector(s.prec.ptr, E::NPREC, true);
  } else {
    static_assert(std::is_same<typename E::Params, EulerParams>::value ||
                      std::is_same<typename E::Params, ShallowWaterParams>::value,
                  "Descriptions with node records");
    /* The pre-pass reads owned U only. Precomputed values and records of the ghost rows are computed locally
     * from the exchanged ghost states (functions of U_j alone: nothing to exchange, :157-160 moves the same
     * numbers), behind the exchange of U on comm_stream. */
    if (have_records) {
      /* the owned rows are complete. Their producers: the last sweep of the step (its export part sits on
       * comm_stream, in front of this exchange in stream order) and, for boundary rows, the kernel above on the
       * compute stream -- with boundary DoFs the exchange is ordered behind the compute stream as well */
      exchange_vector(s.U.ptr, KP, n_groups == 0);
    } else {
      sweep([&](const DeviceMesh &mm, dim3 grid) {
        if (fold_bc)
          hipLaunchKernelGGL((k_precompute_records<E, true>), grid, block, 0, launch_stream, eparams, mm, bc,
                             s.U.ptr, s.prec.ptr, s.rrec.ptr);
        else
          hipLaunchKernelGGL((k_precompute_records<E, false>), grid, block, 0, launch_stream, eparams, mm, bc,
                             s.U.ptr, s.prec.ptr, s.rrec.ptr);
      });
      exchange_vector(s.U.ptr, KP, true);
    }
    if (L.n_relevant > L.n_owned) {
      hipLaunchKernelGGL(k_ghost_precompute_records<E>, dim3(grid_for(L.n_relevant - L.n_owned)), block, 0,
                         n_nbr ? comm_stream : stream, eparams, L.n_owned, L.n_relevant, s.U.ptr, s.prec.ptr,
                         s.rrec.ptr);
      if (n_nbr) { /* work on comm_stream behind the latest export part, as an exchange (not counted as one) */
        comm_pending = true;
        exchange_after_exp = true;
      }
    }
  }
  HIP_CHECK(hipGetLastError());
}

template <typename E>
int ryujin_hip_ctx::step(int h_old, int stages, const int *h_stage, const double *w, int h_new,
                         double tau_in, double tau_max_in, double *tau_out)
{
  constexpr int DIM = E::DIMENSION;
  constexpr bool is_euler = std::is_same<typename E::Params, EulerParams>::value;
  constexpr bool is_aeos = std::is_same<typename E::Params, EulerAeosParams>::value;
  constexpr bool is_scalar = std::is_same<typename E::Params, ScalarParams>::value;
  constexpr bool is_sw = std::is_same<typename E::Params, ShallowWaterParams>::value;
  const auto &eparams = eq_params<E>(); /* shadows the member: the equation's parameter block */
  State &old = state(h_old);
  State &nw = state(h_new);
  if (h_old == h_new)
    throw HipError(RYUJIN_ERR_ARG, "old and new state vector must differ");
  nw.precomputed = false; /* rewritten below */

  const dim3 block(kBlock);

  /* scalars: tau_max := tau_max_in, flags := 0 -- carried by the first sweep of step 2 (step_begin) */
  const bool use_device_tau = deferred && rk_stage > 0;
  struct ClearBegin { /* a step that throws before its first sweep must not leave it armed */
    StepBegin &b;
    ~ClearBegin() { b = StepBegin{}; }
  } clear_begin{pending_begin};
  pending_begin = StepBegin{d_scalars.ptr,
                            tau_max_in,
                            tau_in,
                            (!deferred || rk_stage == 0) ? 1 : 0,
                            (deferred && rk_stage > 0) ? rk_stage - 1 : -1,
                            use_device_tau ? 1 : 0,
                            deferred ? rk_stage : 0};

  bool euler_fast_riemann = false;
  if constexpr (is_euler)
    euler_fast_riemann = eparams.riemann_newton_max_iterations == 0 && eparams.rarefaction_power > 0;
  mark(0);
  step2_split = false;
  /* Step 2: d_ij (upper triangle), alpha_i; ghost alpha (:341-424) */
  if constexpr (is_aeos) {
    if (L.max_row_len > 32)
      throw HipError(RYUJIN_ERR_UNSUPPORTED, "euler aeos: stencils of more than 32 entries");
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      hipLaunchKernelGGL(k_alpha_aeos<DIM>, grid, block, 0, launch_stream, eparams, mm, old.U.ptr, old.prec.ptr,
                         d_alpha.ptr);
    });
    mark(8);
    step2_split = true;
    exchange_vector(d_alpha.ptr, 1, true);
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      hipLaunchKernelGGL(k_dij_aeos<DIM>, grid, block, 0, launch_stream, eparams, mm, d_lower_mask.ptr,
                         old.rrec.ptr, d_dij.ptr);
    });
  } else if constexpr (is_scalar) {
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      hipLaunchKernelGGL(k_dij_alpha_sc<DIM>, grid, block, 0, launch_stream, eparams, mm, old.U.ptr,
                         old.prec.ptr, d_dij.ptr, d_alpha.ptr);
    });
    mark(8);
    step2_split = true;
    exchange_vector(d_alpha.ptr, 1, true);
  } else if (((is_euler && euler_fast_riemann) || is_sw) && L.max_row_len <= 32) {
    /* (Euler's general Riemann path -- Newton iterations or a non-integral exponent -- holds twice the
     * registers and keeps the two-kernel form below) */
    if constexpr (is_euler || is_sw) {
      sweep([&](const DeviceMesh &mm, dim3 grid) {
        hipLaunchKernelGGL((k_dij_alpha_records<E, false>), grid, block, 0, launch_stream, eparams, mm,
                           old.U.ptr, old.prec.ptr, old.rrec.ptr, d_dij.ptr, d_alpha.ptr);
      });
      exchange_vector(d_alpha.ptr, 1, true);
    }
  } else if (is_euler && L.max_row_len <= 32) {
    /* Euler with the general Riemann path: the indicator sweep and the Riemann sweep as two kernels */
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      hipLaunchKernelGGL(k_alpha<E>, grid, block, 0, launch_stream, eparams, mm, old.U.ptr, old.prec.ptr,
                         d_alpha.ptr);
    });
    mark(8); /* end of the indicator kernel: sweep_ms[0] = k_alpha alone */
    step2_split = true;
    exchange_vector(d_alpha.ptr, 1, true); /* overlaps with the Riemann sweep as well */
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      if constexpr (is_euler) {
        if (eparams.riemann_newton_max_iterations == 0 && eparams.rarefaction_power > 0)
          hipLaunchKernelGGL((k_dij_records<E, false>), grid, block, 0, launch_stream, eparams, mm,
                             d_lower_mask.ptr, old.rrec.ptr, d_dij.ptr);
        else
          hipLaunchKernelGGL((k_dij_records<E, true>), grid, block, 0, launch_stream, eparams, mm,
                             d_lower_mask.ptr, old.rrec.ptr, d_dij.ptr);
      }
    });
  } else {
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      hipLaunchKernelGGL(k_dij_alpha<E>, grid, block, 0, launch_stream, eparams, mm, old.U.ptr,
                         old.prec.ptr, d_dij.ptr, d_alpha.ptr);
    });
    exchange_vector(d_alpha.ptr, 1, true);
  }
  mark(1);

  /* Step 3: boundary d_ij, symmetrise, diagonal, tau_max (:432-578) */
  if (n_pairs) {
    join_export(); /* boundary pairs may sit in export rows, whose d_ij the export part wrote on comm_stream
                    * (and may point to ghost columns: the U exchange precedes that export part in stream order) */
    if constexpr (is_aeos)
      hipLaunchKernelGGL(k_dij_boundary_aeos<DIM>, dim3(grid_for(n_pairs)), block, 0, stream, eparams,
                         n_pairs, d_p_i.ptr, d_p_j.ptr, d_p_pos.ptr, d_p_cji.ptr, old.rrec.ptr, d_dij.ptr);
    else if constexpr (is_scalar)
      hipLaunchKernelGGL(k_dij_boundary_sc<DIM>, dim3(grid_for(n_pairs)), block, 0, stream, eparams,
                         n_pairs, d_p_i.ptr, d_p_j.ptr, d_p_pos.ptr, d_p_cji.ptr, old.U.ptr,
                         old.prec.ptr, d_dij.ptr);
    else
      hipLaunchKernelGGL(k_dij_boundary<E>, dim3(grid_for(n_pairs)), block, 0, stream, eparams,
                         n_pairs, d_p_i.ptr, d_p_j.ptr, d_p_pos.ptr, d_p_cji.ptr, old.U.ptr, d_dij.ptr);
  }
  sweep([&](const DeviceMesh &mm, dim3 grid) {
    if (L.max_row_len <= 3)
      hipLaunchKernelGGL(k_dij_diag_unrolled<3>, grid, block, 0, launch_stream, mm, d_lower_mask.ptr,
                         params.cfl, d_dij.ptr, d_scalars.ptr);
    else if (L.max_row_len <= 9)
      hipLaunchKernelGGL(k_dij_diag_unrolled<9>, grid, block, 0, launch_stream, mm, d_lower_mask.ptr,
                         params.cfl, d_dij.ptr, d_scalars.ptr);
    else if (L.max_row_len <= 27)
      hipLaunchKernelGGL(k_dij_diag_unrolled<27>, grid, block, 0, launch_stream, mm, d_lower_mask.ptr,
                         params.cfl, d_dij.ptr, d_scalars.ptr);
    else
      hipLaunchKernelGGL(k_dij_diag, grid, block, 0, launch_stream, mm, params.cfl, d_dij.ptr,
                         d_scalars.ptr);
  });
  /* Utilities::MPI::min(tau_max), :571. Inside a device-resident RK step only the first stage needs the
   * global minimum (it defines tau); later stages use their local tau_max for the validity check only,
   * whose flag is reduced once at the end of the RK step together with the restart flag. */
  if (!(deferred && rk_stage > 0))
    allreduce_scalar(&d_scalars.ptr->tau_max_bits, 0);
  if (want_tau_early && deferred && rk_stage == 0) {
    /* tau_max = min(tau_max argument, CFL bound over all ranks) is final here: hand it to the host while
     * steps 4-7 of this stage run (time-dependent Dirichlet data of the later stages) */
    join_export();
    HIP_CHECK(hipMemcpyAsync(h_tau_early, &d_scalars.ptr->tau_max_bits, sizeof(unsigned long long),
                             hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipEventRecord(ev_tau, stream));
  }
  /* tau itself is resolved inside the step-4 kernel (finalize_tau) */
  mark(2);

  /* Step 4: low-order update, bounds, r_i, p_ij; ghost r (:597-884) */
  double weight;
  {
    double acc = -1.;
    for (int s = 0; s < stages; ++s)
      acc += w[s];
    weight = -acc;
  }
  StageArgs<DIM> S{};
  S.stages = stages;
  for (int s = 0; s < stages; ++s) {
    S.U[s] = state(h_stage[s]).U.ptr;
    S.prec[s] = state(h_stage[s]).prec.ptr;
    S.w[s] = w[s];
  }
  /* Euler, stages == 0, limiter on: P_ij (part 1) is recomputed in step 5 instead of stored here.
   * A/B on MI355X: -7 % per update in 2-D (k=4, 9 columns). In 3-D (k=5, 27 columns) step 4 drops from 2.01
   * to 1.34 ms on 4.2 M gridpoints but step 5 grows from 2.16 to 2.77-2.94 ms (27 flux evaluations per row at
   * 240 registers): -0.7 % ... +1.5 % per update, inside the run-to-run spread -- so only for dim <= 2. */
  const bool recompute_p = is_euler && DIM <= 2 && stages == 0 && params.limiter_iterations != 0 && !dg;
  /* Euler and EulerAEOS, stages == 0, Q1 stencil widths: step 4 does not touch P_ij; step 5 forms it once from
   * d_ij, m_ij and the per-node vectors, limits it and stores it for steps 6 and 7 (kernels_limiter_stage0.hpp)
   * -- any dimension */
  constexpr int kStage0Width = DIM == 1 ? 3 : (DIM == 2 ? 9 : 27);
  const bool stage0_pij = RYUJIN_STAGE0_PIJ && (is_euler || is_aeos) && stages == 0 && params.limiter_iterations != 0 && !dg &&
                          L.max_row_len <= (uint32_t)kStage0Width;
  stage0_V = false;
  if (params.limiter_iterations == 2 && d_V.n == 0) {
    d_V.alloc((size_t)L.n_relevant * KP);
    d_slice_unlimited.alloc(L.n_slices);
  }
  /* step 5 on small meshes: up to four waves per slice, each taking a share of the columns (decided for the whole
   * mesh, not per launch: the export and the interior part of a split sweep must agree on whether V_i exists) */
  const uint32_t step5_groups = std::min<uint32_t>(
      4u, resident_waves_step5 /
              std::max<uint32_t>(1u, (L.n_slices + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock));
  /* ... and P_ij is stored per slice -- only where steps 6 and 7 will read it (kernels_limiter_stage0.hpp) -- where the
   * update has two limiter passes and one wave per slice, WHILE that pays: its bookkeeping (the prediction and the
   * trigger in step 5, step 6 as three launches) costs a few per cent of the three sweeps, the savings are
   * proportional to the share of unlimited slices. Above RYUJIN_PER_SLICE_MAX_LIMITED (the measured break-even,
   * profiles/r04*_ab_limited_fraction*) the plain kernels run: P_ij stored everywhere, step 6 in one launch. Same
   * bits either way; the fraction is the one step 6 counted between the two latest host synchronisations (1 until
   * the first: the first update of a context runs the plain kernels). */
  const bool per_slice_possible =
      RYUJIN_PER_SLICE_PIJ && stage0_pij && params.limiter_iterations == 2 && step5_groups < 2;
  /* Up to two dimensions, finer still: PER TILE. Step 5 stores a (slice, column) tile iff one of its own l_ij comes
   * out limited or step 6 read the tile in one of the last updates (SliceFlags::needed_tiles); step 6 -- one launch,
   * the plain kernel -- forms the few tiles that are limited through the neighbour's l_ji alone and were not
   * predicted (kernels_limiter_stage0.hpp, next_cached_slice). On the Mach-3 step a third to 45 % of the tiles are
   * stored where 71 - 93 % of the slices would be, and the update is faster than with either alternative at every
   * stage of the flow (profiles/r05t_ab_tile_*). debug_pij_storage: 0 this; 2 per tile with nothing predicted (tests:
   * every tile the neighbour's l_ji limits goes through step 6's repair); 1 per slice, nothing predicted; < 0
   * everywhere, as rounds 1 - 4. Not with the checked build (its kernels read all of P_ij). In 3-D per tile is built
   * (the tiles step 5 did not store formed by a launch behind step 6, kernels_limiter.hpp: kHoDefer), measured as a
   * small loss (RYUJIN_TILE_PIJ_DEFAULT_MAXDIM) and selected by debug_pij_storage = 3 (4: nothing predicted) only:
   * per slice there while few slices are limited, everywhere after that. */
  const int storage = params.debug_pij_storage;
  const bool tile_store = RYUJIN_TILE_PIJ && DIM <= RYUJIN_TILE_PIJ_MAXDIM && per_slice_possible &&
                          (((storage == 0 || storage == 2) && DIM <= RYUJIN_TILE_PIJ_DEFAULT_MAXDIM) || storage == 3 ||
                           storage == 4) &&
                          !params.debug_expensive_bounds_check;
  const bool per_slice =
      per_slice_possible && !tile_store && storage >= 0 && !params.debug_expensive_bounds_check &&
      (storage == 1 || storage == 2 || limited_fraction <= (double)RYUJIN_PER_SLICE_MAX_LIMITED);
  ensure_pij();
  if (per_slice && d_slice_first_stored.n == 0) {
    d_slice_first_stored.alloc(L.n_slices);
    d_slice_todo.alloc(L.n_slices);
  }
  const SliceFlags slice_flags{d_slice_unlimited.ptr, d_slice_first_stored.ptr, d_slice_todo.ptr};
  last_per_slice = per_slice;
  last_tile_store = tile_store;
  const bool tiles_predicted_from_history = tile_store && (storage == 0 || storage == 3);
  constexpr int kHistoryWords = tile_history_words(kStage0Width);
  if (tiles_predicted_from_history && d_slice_needed.n == 0) { /* all ones: the first update stores every tile */
    d_slice_needed.alloc((size_t)L.n_slices * kHistoryWords);
    HIP_CHECK(hipMemsetAsync(d_slice_needed.ptr, 0xff, (size_t)L.n_slices * kHistoryWords * sizeof(uint32_t),
                             launch_stream));
  }
  /* the tiles step 5 did not store are formed outside the sweep of step 6 (kernels_limiter.hpp, kHoDefer) */
  constexpr bool kDeferTiles = DIM >= RYUJIN_TILE_DEFER_MINDIM;
  if (tile_store && kDeferTiles && d_slice_deferred.n == 0)
    d_slice_deferred.alloc(L.n_slices);
  const SliceFlags tile_flags{nullptr, nullptr, nullptr,
                              tiles_predicted_from_history ? d_slice_needed.ptr : nullptr, L.n_slices,
                              (tile_store && kDeferTiles) ? d_slice_deferred.ptr : nullptr};
  last_s0 = Stage0Src{d_scalars.ptr, old.U.ptr, d_alpha.ptr, d_dij.ptr, d_r.ptr, tile_store ? 1 : 0};
  sweep([&](const DeviceMesh &mm, dim3 grid) {
    if constexpr (is_euler) {
      if (dg && stages == 0)
        hipLaunchKernelGGL((k_low_order<DIM, false, true, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (dg)
        hipLaunchKernelGGL((k_low_order<DIM, true, true, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (recompute_p || stage0_pij)
        hipLaunchKernelGGL((k_low_order<DIM, false, false>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (stages == 0)
        hipLaunchKernelGGL((k_low_order<DIM, false>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else
        hipLaunchKernelGGL((k_low_order<DIM, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
    } else if constexpr (is_scalar) {
      if (dg && stages == 0)
        hipLaunchKernelGGL((k_low_order_sc<DIM, false, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (dg)
        hipLaunchKernelGGL((k_low_order_sc<DIM, true, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (stages == 0)
        hipLaunchKernelGGL((k_low_order_sc<DIM, false>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else
        hipLaunchKernelGGL((k_low_order_sc<DIM, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
    } else if constexpr (is_aeos) {
      if (dg && stages == 0)
        hipLaunchKernelGGL((k_low_order_aeos<DIM, false, true, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (dg)
        hipLaunchKernelGGL((k_low_order_aeos<DIM, true, true, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (stage0_pij)
        hipLaunchKernelGGL((k_low_order_aeos<DIM, false, false>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (stages == 0)
        hipLaunchKernelGGL((k_low_order_aeos<DIM, false>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else
        hipLaunchKernelGGL((k_low_order_aeos<DIM, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_alpha.ptr, d_dij.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
    } else {
      /* rows of at most 3 / 9 columns (1-D, 2-D Q1): one walk over the stencil, the shift-free part of the
       * limiter's U_ij_bar parked in LDS (kernels_shallow_water.hpp); wider rows: the two walks of the reference */
      constexpr int kSwWidth = DIM == 1 ? 3 : 9;
      const bool single_walk = RYUJIN_SW_SINGLE_WALK && !dg && L.max_row_len <= (uint32_t)kSwWidth;
      auto launch_single_walk = [&](auto has_stages, auto friction) {
        hipLaunchKernelGGL((k_low_order_sw_single_walk<DIM, decltype(has_stages)::value, kSwWidth,
                                                       decltype(friction)::value>),
                           grid, block, 0, launch_stream, eparams, mm, d_scalars.ptr, weight, S, old.U.ptr,
                           old.prec.ptr, d_Z.ptr, d_alpha.ptr, d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      };
      const bool friction = eparams.manning != 0.;
      if (single_walk && stages == 0 && friction)
        launch_single_walk(std::false_type{}, std::true_type{});
      else if (single_walk && stages == 0)
        launch_single_walk(std::false_type{}, std::false_type{});
      else if (single_walk && friction)
        launch_single_walk(std::true_type{}, std::true_type{});
      else if (single_walk)
        launch_single_walk(std::true_type{}, std::false_type{});
      else if (dg && stages == 0)
        hipLaunchKernelGGL((k_low_order_sw<DIM, false, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_Z.ptr, d_alpha.ptr,
                           d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (dg)
        hipLaunchKernelGGL((k_low_order_sw<DIM, true, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_Z.ptr, d_alpha.ptr,
                           d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else if (stages == 0)
        hipLaunchKernelGGL((k_low_order_sw<DIM, false>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_Z.ptr, d_alpha.ptr,
                           d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
      else
        hipLaunchKernelGGL((k_low_order_sw<DIM, true>), grid, block, 0, launch_stream, eparams, mm,
                           d_scalars.ptr, weight, S, old.U.ptr, old.prec.ptr, d_Z.ptr, d_alpha.ptr,
                           d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr);
    }
  });
  /* EXPENSIVE_BOUNDS_CHECK as a run-time option (Euler): is_admissible() of the low-order update (:851-855) */
  const bool checked = params.debug_expensive_bounds_check != 0;
  if (checked)
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      hipLaunchKernelGGL(k_check_admissible<E>, grid, block, 0, launch_stream, eparams, mm, d_scalars.ptr,
                         (const double *)nw.U.ptr);
    });
  exchange_vector(d_r.ptr, KP, true);
  if (dg && params.limiter_iterations != 0) {
    /* the bounds are extended over the stencil in step 5: their ghost range has to be current
     * (hyperbolic_module.template.h:601-613); SoA, one scalar vector per bound */
    for (int b = 0; b < NB; ++b)
      exchange_vector(d_bounds.ptr + (size_t)b * bounds_stride, 1, true);
  }
  mark(3);

  /* Step 5: second part of p_ij, first l_ij; ghost rows of l_ij (:892-1041) */
  const int n_iterations = params.limiter_iterations;
  if (dg && n_iterations != 0) {
    /* bounds over the stencil (:938-948) with the Description's Limiter::combine_bounds; steps 5-7 read the
     * extended bounds */
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      if constexpr (is_euler)
        hipLaunchKernelGGL(k_bounds_combine_euler, grid, block, 0, launch_stream, mm, d_bounds.ptr,
                           d_bounds_combined.ptr);
      else if constexpr (is_sw)
        hipLaunchKernelGGL(k_bounds_combine_sw, grid, block, 0, launch_stream, mm, d_bounds.ptr,
                           d_bounds_combined.ptr);
      else if constexpr (is_aeos)
        hipLaunchKernelGGL((k_bounds_combine_minmax<4, 0x2u>), grid, block, 0, launch_stream, mm, d_bounds.ptr,
                           d_bounds_combined.ptr);
      else
        hipLaunchKernelGGL((k_bounds_combine_minmax<2, 0x2u>), grid, block, 0, launch_stream, mm, d_bounds.ptr,
                           d_bounds_combined.ptr);
    });
    std::swap(d_bounds.ptr, d_bounds_combined.ptr);
  }
  if (n_iterations != 0) {
    sweep([&](const DeviceMesh &mm, dim3 grid) {
      if constexpr (is_euler || is_aeos) {
        if (stage0_pij) {
          /* small meshes: up to four waves per slice, each taking a share of the columns (decided for the whole
           * mesh, not per launch: the export and the interior part of a split sweep must agree on whether V_i exists) */
          const uint32_t groups = step5_groups;
          auto launch5 = [&](auto ny) {
            constexpr int NY = decltype(ny)::value;
            hipLaunchKernelGGL((k_lij_stage0<E, NY>), dim3(grid.x, NY), block, 0, launch_stream, eparams, mm,
                               d_scalars.ptr, old.U.ptr, d_alpha.ptr, d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr,
                               d_pij.ptr, d_lij.ptr, NY == 1 ? d_V.ptr : nullptr, tile_flags, 0);
            stage0_V = NY == 1 && d_V.ptr != nullptr;
          };
          if (tile_store && groups < 2) { /* (tile_store implies one wave per slice) */
            hipLaunchKernelGGL((k_lij_stage0<E, 1, false, true>), grid, block, 0, launch_stream, eparams, mm,
                               d_scalars.ptr, old.U.ptr, d_alpha.ptr, d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr,
                               d_pij.ptr, d_lij.ptr, d_V.ptr, tile_flags, 0);
            stage0_V = d_V.ptr != nullptr;
            return;
          }
          if (per_slice) {
            hipLaunchKernelGGL((k_lij_stage0<E, 1, true>), grid, block, 0, launch_stream, eparams, mm,
                               d_scalars.ptr, old.U.ptr, d_alpha.ptr, d_dij.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr,
                               d_pij.ptr, d_lij.ptr, d_V.ptr, slice_flags, params.debug_pij_storage);
            stage0_V = true;
          } else if (groups >= 4)
            launch5(std::integral_constant<int, 4>{});
          else if (groups == 3)
            launch5(std::integral_constant<int, 3>{});
          else if (groups == 2)
            launch5(std::integral_constant<int, 2>{});
          else
            launch5(std::integral_constant<int, 1>{});
          return;
        }
      }
      if constexpr (is_euler) {
        if (recompute_p) {
          /* small meshes: up to four waves per slice, each taking a share of the columns (see the kernel), as
           * long as all of them are resident at once (256 CUs x 4 SIMDs x 2 waves of this kernel) */
          const uint32_t groups = std::min<uint32_t>(4u, resident_waves_step5 / std::max<uint32_t>(1u, grid.x * kWavesPerBlock));
          auto launch5 = [&](auto ny) {
            constexpr int NY = decltype(ny)::value;
            hipLaunchKernelGGL((k_pij_lij_recompute<DIM, NY>), dim3(grid.x, NY), block, 0, launch_stream, eparams,
                               mm, d_scalars.ptr, weight, old.U.ptr, d_alpha.ptr, d_dij.ptr, nw.U.ptr, d_r.ptr,
                               d_bounds.ptr, d_pij.ptr, d_lij.ptr);
          };
          if (groups >= 4)
            launch5(std::integral_constant<int, 4>{});
          else if (groups == 3)
            launch5(std::integral_constant<int, 3>{});
          else if (groups == 2)
            launch5(std::integral_constant<int, 2>{});
          else
            launch5(std::integral_constant<int, 1>{});
          return;
        }
      }
      {
        if (dg) {
          if (L.max_row_len > 64)
            hipLaunchKernelGGL((k_pij_lij<E, true, true>), grid, block, tail_queue_bytes(mm), launch_stream, eparams, mm,
                               d_scalars.ptr, nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_V.ptr);
          else
            hipLaunchKernelGGL((k_pij_lij<E, true>), grid, block, tail_queue_bytes(mm), launch_stream, eparams, mm, d_scalars.ptr,
                               nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_V.ptr);
          stage0_V = d_V.ptr != nullptr;
          return;
        }
      }
      if (L.max_row_len > 64)
        hipLaunchKernelGGL((k_pij_lij<E, false, true>), grid, block, tail_queue_bytes(mm), launch_stream, eparams, mm, d_scalars.ptr,
                           nw.U.ptr, d_r.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_V.ptr);
      else
        hipLaunchKernelGGL(k_pij_lij<E>, grid, block, tail_queue_bytes(mm), launch_stream, eparams, mm, d_scalars.ptr, nw.U.ptr,
                           d_r.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_V.ptr);
      stage0_V = d_V.ptr != nullptr;
    });
    if (checked) /* the first limiter pass in the checked control flow (limiter.template.h:110-134,244-322) */
      sweep([&](const DeviceMesh &mm, dim3 grid) {
        hipLaunchKernelGGL(k_check_limiter<E>, grid, block, 0, launch_stream, eparams, mm, d_scalars.ptr,
                           (const double *)nw.U.ptr, (const double *)d_bounds.ptr, (const double *)d_pij.ptr,
                           (const double *)nullptr);
      });
    exchange_matrix(d_lij.ptr, true);
  }
  mark(4);

  /* Steps 6, 7: symmetrise l_ij, high-order update, next l_ij (:1053-1182) */
  /* a pending sadd of the device-resident RK driver is applied by the last sweep */
  const FusedSadd fused_sadd = pending_sadd;
  pending_sadd = FusedSadd{0., 0., nullptr};
  /* ... and so are the precomputed values and Riemann records of the new vector, where the next pre-pass would be a
   * sweep of its own (large meshes: below bc_fold_max_slices the boundary conditions ride on that sweep) */
  bool fuse_precompute = false;
  if constexpr (E::kFusablePrecompute)
    fuse_precompute = RYUJIN_FUSE_PRECOMPUTE && pending_precompute && n_iterations != 0 &&
                      L.n_slices > bc_fold_max_slices &&
                      L.max_row_len <= (uint32_t)(DIM == 1 ? 3 : (DIM == 2 ? 9 : 27));
  pending_precompute = false;
  nw.precomputed = false;
  const FusedPrecompute fused_prec{fuse_precompute ? nw.prec.ptr : nullptr, fuse_precompute ? nw.rrec.ptr : nullptr};
  if (fused_sadd.src && n_iterations == 0)
    throw HipError(RYUJIN_ERR_ARG, "internal: fused sadd without a limiter pass");
  bool step6_flags = false; /* step 6 left SliceFlags::unlimited for every slice: the last sweep may use it */
  for (int pass = 0; pass < n_iterations; ++pass) {
    const bool last_round = (pass + 1 == n_iterations);
    if (n_iterations == 2 && last_round)
      std::swap(d_lij.ptr, d_lij_next.ptr);
    constexpr int kCachedWidth = DIM == 1 ? 3 : (DIM == 2 ? 9 : 27);
    if (last_round) {
      sweep([&](const DeviceMesh &mm, dim3 grid) {
        constexpr int kLastChunk = DIM == 3 ? RYUJIN_LAST_CHUNK_3D : (DIM == 2 ? RYUJIN_LAST_CHUNK_2D : kCachedWidth);
        if (L.max_row_len <= (uint32_t)kCachedWidth)
          hipLaunchKernelGGL((k_high_order_last_cached<E, kCachedWidth, kLastChunk>), grid,
                             block, 0, launch_stream, eparams, mm, nw.U.ptr, d_pij.ptr, d_lij.ptr, fused_sadd,
                             fused_prec, step6_flags ? d_slice_unlimited.ptr : nullptr);
        else
          hipLaunchKernelGGL((k_high_order<E, true>), grid, block, 0, launch_stream, eparams, mm, nw.U.ptr,
                             d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_lij_next.ptr, fused_sadd);
      });
    } else {
      /* 3-D: cache all l_ij and the P_ij of the first RYUJIN_HO_CP_3D columns (0: two-pass kernel) */
      constexpr int kCachedP = DIM == 3 ? (RYUJIN_HO_CP_3D > 0 ? RYUJIN_HO_CP_3D : 27)
                                        : (DIM == 2 ? (RYUJIN_HO_CP_2D < kCachedWidth ? RYUJIN_HO_CP_2D : kCachedWidth) : kCachedWidth);
      sweep([&](const DeviceMesh &mm, dim3 grid) {
        if constexpr (is_euler || is_aeos) {
          if (per_slice) {
            /* three launches over all slices: the light one finishes the slices without a stored P_ij in which
             * nothing was limited (V_i), the repair launch completes the P_ij of the slices that turned out limited
             * without (all of) it, the heavy one runs the limited slices (kernels_limiter.hpp) */
            hipLaunchKernelGGL((k_high_order_next_cached<E, kCachedWidth, kCachedP, false, kHoLight>), grid, block, 0,
                               launch_stream, eparams, mm, nw.U.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr,
                               d_lij_next.ptr, d_V.ptr, last_s0, slice_flags);
            hipLaunchKernelGGL(k_pij_repair<E>, grid, block, 0, launch_stream, mm, last_s0, d_pij.ptr, slice_flags);
            hipLaunchKernelGGL((k_high_order_next_cached<E, kCachedWidth, kCachedP, false, kHoHeavy>), grid, block, 0,
                               launch_stream, eparams, mm, nw.U.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr,
                               d_lij_next.ptr, d_V.ptr, last_s0, slice_flags);
            step6_flags = true;
            return;
          }
        }
        if constexpr (DIM <= 2) {
          /* small meshes: the four waves of a block share one slice (see the kernel) while all of them fit */
          const uint32_t n_launch = mm.slice_end - mm.slice_begin;
          if (!tile_store && L.max_row_len <= (uint32_t)kCachedWidth && n_launch * kWavesPerBlock <= resident_waves_step6) {
            /* (not with P_ij stored per tile: the export part of a large mesh may be this small, and the split
             * variant does not form the tiles step 5 left out) */
            hipLaunchKernelGGL((k_high_order_next_cached<E, kCachedWidth, kCachedWidth, true>), dim3(n_launch),
                               block, 0, launch_stream, eparams, mm, nw.U.ptr, d_bounds.ptr, d_pij.ptr,
                               d_lij.ptr, d_lij_next.ptr, stage0_V ? d_V.ptr : nullptr, last_s0,
                               SliceFlags{stage0_V ? d_slice_unlimited.ptr : nullptr, nullptr, nullptr});
            step6_flags = stage0_V;
            return;
          }
        }
        if ((DIM <= 2 || RYUJIN_HO_CP_3D > 0) && L.max_row_len <= (uint32_t)kCachedWidth) {
          SliceFlags flags6 = tile_flags;
          flags6.unlimited = stage0_V ? d_slice_unlimited.ptr : nullptr;
          bool launched = false;
          if constexpr (kDeferTiles && (is_euler || is_aeos)) {
            if (tile_store && stage0_V) {
              /* the sweep, and behind it the slices that missed a tile (a grid of fixed size walks the list) */
              hipLaunchKernelGGL((k_high_order_next_cached<E, kCachedWidth, kCachedP, false, kHoDefer>), grid, block,
                                 0, launch_stream, eparams, mm, nw.U.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr,
                                 d_lij_next.ptr, d_V.ptr, last_s0, flags6);
              hipLaunchKernelGGL((k_high_order_next_deferred<E, kCachedWidth>),
                                 dim3(std::min<uint32_t>(grid.x * kWavesPerBlock, 512u)), block, 0, launch_stream, eparams, mm,
                                 nw.U.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_lij_next.ptr, d_V.ptr, last_s0,
                                 flags6);
              launched = true;
            }
          }
          if (!launched)
            hipLaunchKernelGGL((k_high_order_next_cached<E, kCachedWidth, kCachedP>), grid, block, 0,
                               launch_stream, eparams, mm, nw.U.ptr, d_bounds.ptr, d_pij.ptr, d_lij.ptr,
                               d_lij_next.ptr, stage0_V ? d_V.ptr : nullptr, last_s0, flags6);
          step6_flags = stage0_V;
        } else if (L.max_row_len > 64)
          hipLaunchKernelGGL((k_high_order<E, false, true>), grid, block, 0, launch_stream, eparams, mm, nw.U.ptr,
                             d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_lij_next.ptr, FusedSadd{0., 0., nullptr});
        else
          hipLaunchKernelGGL((k_high_order<E, false>), grid, block, 0, launch_stream, eparams, mm, nw.U.ptr,
                             d_bounds.ptr, d_pij.ptr, d_lij.ptr, d_lij_next.ptr, FusedSadd{0., 0., nullptr});
      });
      if (checked) /* the update after the first pass (:1121-1126) and the second pass's success (:1155-1161) */
        sweep([&](const DeviceMesh &mm, dim3 grid) {
          hipLaunchKernelGGL(k_check_admissible<E>, grid, block, 0, launch_stream, eparams, mm, d_scalars.ptr,
                             (const double *)nw.U.ptr);
          hipLaunchKernelGGL(k_check_limiter<E>, grid, block, 0, launch_stream, eparams, mm, d_scalars.ptr,
                             (const double *)nw.U.ptr, (const double *)d_bounds.ptr, (const double *)d_pij.ptr,
                             (const double *)d_lij.ptr);
        });
      exchange_matrix(d_lij_next.ptr, true);
    }
    if (checked && last_round) /* the final update (:1121-1126) */
      sweep([&](const DeviceMesh &mm, dim3 grid) {
        hipLaunchKernelGGL(k_check_admissible<E>, grid, block, 0, launch_stream, eparams, mm, d_scalars.ptr,
                           (const double *)nw.U.ptr);
      });
    mark(5 + pass);
  }
  for (int k = 5 + n_iterations; k <= 7; ++k)
    mark(k);
  nw.precomputed = fuse_precompute;

  join_export(); /* (the exchange of U_new stays in flight: whoever reads its ghost range joins it) */
  if (!deferred)
    allreduce_scalar(&d_scalars.ptr->restart_needed, 1); /* MPI::logical_or(restart_needed), :1194 */
  /* (deferred: the restart flag is folded into its accumulator by the next stage's step_begin(), or by
   * time_step() behind the last stage) */

  HIP_CHECK(hipGetLastError());
  if (deferred) {
    *tau_out = std::numeric_limits<double>::quiet_NaN(); /* known at the end of the RK step */
    return RYUJIN_OK;
  }
  HIP_CHECK(hipMemcpyAsync(h_scalars, d_scalars.ptr, sizeof(DeviceScalars), hipMemcpyDeviceToHost,
                           stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  update_limited_fraction();

  if (timers_enabled) {
    for (int k = 0; k < 7; ++k) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
      sweep_ms[k + 1] = ms;
    }
    sweep_ms[0] = 0.;
    if (step2_split) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[8]));
      sweep_ms[0] = ms;
    }
    for (int k = 0; k < 8; ++k)
      sweep_ms_accum[k] += sweep_ms[k];
    ++sweep_updates_accum;
  }

  if (h_scalars->tau_invalid)
    return RYUJIN_ERR_TAU;
  *tau_out = h_scalars->tau;

  if (h_scalars->restart_needed) {
    if (params.id_violation_strategy == RYUJIN_IDV_WARN) {
      n_warnings++;
      return RYUJIN_WARN;
    }
    n_restarts++;
    return RYUJIN_RESTART;
  }
  return RYUJIN_OK;
}

/* Device-resident Runge-Kutta driver: ryujin::TimeIntegrator::step for the
 * schemes that consist solely of prepare_state_vector + step<s> + sadd
 * (source/time_integrator.template.h:207-403), including the bang-bang CFL recovery (:250-274).
 * All stages are enqueued back to back: the tau of the first stage stays on the device, the restart /
 * tau-validity flags of all stages are accumulated there, and the host synchronises ONCE per RK step.
 * A Restart is therefore detected at the end of the RK step instead of inside it; the reference repeats
 * the whole RK step from the untouched state vector anyway, so the outcome is the same. */
template <typename E>
int ryujin_hip_ctx::time_step(int scheme, int h_state, int n_tmp, const int *h_tmp, const double *dirichlet,
                              double tau_max, int cfl_recovery, double cfl_min, double cfl_max,
                              double *tau_out, double t, ryujin_hip_dirichlet_fn dirichlet_fn,
                              void *dirichlet_user)
{
  const int U = h_state;
  int n_stages = 0;
  double tau_factor = 1.;
  switch (scheme) {
  case RYUJIN_SCHEME_ERK_11: n_stages = 1; break;
  case RYUJIN_SCHEME_SSPRK_22: n_stages = 2; break;
  case RYUJIN_SCHEME_ERK_22: n_stages = 2; tau_factor = 2.; break;
  case RYUJIN_SCHEME_SSPRK_33: n_stages = 3; break;
  case RYUJIN_SCHEME_ERK_33: n_stages = 3; tau_factor = 3.; break;
  case RYUJIN_SCHEME_ERK_43: n_stages = 4; tau_factor = 4.; break;
  case RYUJIN_SCHEME_ERK_54: n_stages = 5; tau_factor = 5.; break;
  default: throw HipError(RYUJIN_ERR_ARG, "unknown time stepping scheme");
  }
  const bool erk = scheme != RYUJIN_SCHEME_SSPRK_22 && scheme != RYUJIN_SCHEME_SSPRK_33;
  /* temp_ vectors: SSPRK22 2, SSPRK33 2, ERK s stages: s (time_integrator.template.h:163-205) */
  const int n_needed = erk ? n_stages : 2;
  if (n_tmp < n_needed)
    throw HipError(RYUJIN_ERR_ARG, "time_step: not enough temporary state vectors for this scheme");
  const int *T = h_tmp;
  const double first_tau_max = erk ? tau_max / n_stages : tau_max;

  /* explicit Runge-Kutta stages 2.. as (number of stage vectors, their handles, weights):
   * step_erk_22/33/43/54 (time_integrator.template.h:341-500) */
  struct ErkStage {
    int n;
    int h[4];
    double w[4];
  };
  ErkStage erk_stage[5] = {};
  if (scheme == RYUJIN_SCHEME_ERK_22) {
    erk_stage[1] = {1, {U}, {-1.}};
  } else if (scheme == RYUJIN_SCHEME_ERK_33) {
    erk_stage[1] = {1, {U}, {-1.}};
    erk_stage[2] = {2, {U, T[0]}, {0.75, -2.}};
  } else if (scheme == RYUJIN_SCHEME_ERK_43) {
    erk_stage[1] = {1, {U}, {-1.}};
    erk_stage[2] = {1, {T[0]}, {-1.}};
    erk_stage[3] = {2, {T[0], T[1]}, {5. / 3., -10. / 3.}};
  } else if (scheme == RYUJIN_SCHEME_ERK_54) {
    constexpr double c = 0.2;
    constexpr double a_21 = +0.2;
    constexpr double a_31 = +0.26075582269554909, a_32 = +0.13924417730445096;
    constexpr double a_41 = -0.25856517872570289, a_42 = +0.91136274166280729, a_43 = -0.05279756293710430;
    constexpr double a_51 = +0.21623276431503774, a_52 = +0.51534223099602405, a_53 = -0.81662794199265554,
                     a_54 = +0.88505294668159373;
    constexpr double a_61 = -0.10511678454691901, a_62 = +0.87880047152100838, a_63 = -0.58903404061484477,
                     a_64 = +0.46213380485434047;
    erk_stage[1] = {1, {U}, {(a_31 - a_21) / c}};
    erk_stage[2] = {2, {U, T[0]}, {(a_41 - a_31) / c, (a_42 - a_32) / c}};
    erk_stage[3] = {3, {U, T[0], T[1]}, {(a_51 - a_41) / c, (a_52 - a_42) / c, (a_53 - a_43) / c}};
    erk_stage[4] = {4, {U, T[0], T[1], T[2]},
                    {(a_61 - a_51) / c, (a_62 - a_52) / c, (a_63 - a_53) / c, (a_64 - a_54) / c}};
  }

  auto single_step = [&]() -> int {
    double dummy = 0.;
    deferred = true;
    struct Guard {
      bool &flag, &armed;
      ~Guard() { flag = armed = false; }
    } guard{deferred, pending_precompute};
    int result = -1; /* handle that holds the new solution */
    const double no_limit = std::numeric_limits<double>::max();
    const int none[1] = {0};
    const double no_w[1] = {0.};

    /* Dirichlet data of stage st: time independent (the array, uploaded once), or initial_state(position,
     * t + c_s tau) per stage (time_integrator.template.h:279-510: SSPRK22 t, t+tau; SSPRK33 t, t+tau, t+tau/2;
     * ERK stage s at t + s tau) through the caller's function. tau is known on the host as soon as step 3 of
     * the first stage has run; the first stage's remaining sweeps are already enqueued behind it. */
    const bool time_dependent = dirichlet_fn != nullptr && n_bdry > 0 && needs_dirichlet;
    std::vector<double> stage_data;
    double tau_first = 0.;
    auto dirichlet_at = [&](int st) -> const double * {
      if (!time_dependent)
        return st == 0 ? dirichlet : nullptr;
      double c = 0.;
      if (st > 0)
        c = erk ? (double)st : (scheme == RYUJIN_SCHEME_SSPRK_33 && st == 2 ? 0.5 : 1.);
      stage_data.assign((size_t)n_bdry * K, 0.);
      dirichlet_fn(dirichlet_user, t + c * tau_first, stage_data.data());
      return stage_data.data();
    };
    struct EarlyTau {
      bool &flag;
      ~EarlyTau() { flag = false; }
    } early_tau{want_tau_early};
    if (time_dependent && n_stages > 1) {
      if (!h_tau_early) {
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&h_tau_early), sizeof(unsigned long long)));
        HIP_CHECK(hipEventCreateWithFlags(&ev_tau, hipEventDisableTiming));
      }
      want_tau_early = true;
    }

    rk_stage = 0;
    prepare_state_vector<E>(U, dirichlet_at(0));
    pending_precompute = true; /* every stage's result goes straight into the next prepare_state_vector() */
    step<E>(U, 0, none, no_w, T[0], 0., first_tau_max, &dummy);
    result = T[0];
    if (want_tau_early) {
      HIP_CHECK(hipEventSynchronize(ev_tau));
      tau_first = __builtin_bit_cast(double, *h_tau_early);
      if (!(tau_first > 0.) || std::isinf(tau_first))
        tau_first = 0.; /* invalid tau_max: the step ends in RYUJIN_ERR_TAU whatever the later stages see */
    }
    if (erk) {
      for (int st = 1; st < n_stages; ++st) {
        rk_stage = st;
        prepare_state_vector<E>(T[st - 1], dirichlet_at(st));
        pending_precompute = true;
        step<E>(T[st - 1], erk_stage[st].n, erk_stage[st].h, erk_stage[st].w, T[st], 1. /*device tau*/,
                no_limit, &dummy);
        result = T[st];
      }
    } else {
      /* sadd(dst, s, b, U) right after step(.., dst): folded into the last sweep of that step when there
       * is one (limiter iterations >= 1), saving one pass over the state vectors per stage */
      /* (not with the checked build's extra kernels: they look at the update before the sadd, as the reference does) */
      const bool fuse = params.limiter_iterations >= 1 && !params.debug_expensive_bounds_check;
      auto stage_with_sadd = [&](int h_old, int h_new, double sa, double sb) {
        struct Disarm { /* a step that throws before its last sweep must not leave the sadd armed */
          FusedSadd &p;
          ~Disarm() { p = FusedSadd{0., 0., nullptr}; }
        } disarm{pending_sadd};
        if (fuse)
          pending_sadd = FusedSadd{sa, sb, state(U).U.ptr};
        pending_precompute = fuse; /* (an sadd of its own rewrites the vector) */
        step<E>(h_old, 0, none, no_w, h_new, 1., no_limit, &dummy);
        if (!fuse)
          ryujin_hip_sadd(this, h_new, sa, sb, U);
      };
      rk_stage = 1;
      prepare_state_vector<E>(T[0], dirichlet_at(1));
      if (scheme == RYUJIN_SCHEME_SSPRK_22)
        stage_with_sadd(T[0], T[1], 1. / 2., 1. / 2.);
      else
        stage_with_sadd(T[0], T[1], 1. / 4., 3. / 4.);
      result = T[1];
      if (n_stages >= 3) {
        rk_stage = 2;
        prepare_state_vector<E>(T[1], dirichlet_at(2));
        stage_with_sadd(T[1], T[0], 2. / 3., 1. / 3.);
        result = T[0];
      }
    }
    /* the only host synchronisation of the RK step */
    join_export();
    hipLaunchKernelGGL(k_accumulate_flags, dim3(1), dim3(1), 0, stream, rk_stage, d_scalars.ptr);
    /* MPI::logical_or over the ranks of the flags accumulated over all stages (restart_accum and
     * tau_invalid_accum are adjacent ints): one collective per RK step instead of one per stage */
    static_assert(offsetof(DeviceScalars, tau_invalid_accum) ==
                      offsetof(DeviceScalars, restart_accum) + sizeof(int),
                  "flag accumulators must be adjacent");
    allreduce_scalar(&d_scalars.ptr->restart_accum, 1, 2);
    HIP_CHECK(hipMemcpyAsync(h_scalars, d_scalars.ptr, sizeof(DeviceScalars), hipMemcpyDeviceToHost,
                             stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    update_limited_fraction();
    if (timers_enabled) {
      for (int st = 0; st < n_stages; ++st) {
        for (int k = 0; k < 7; ++k) {
          float ms = 0.f;
          HIP_CHECK(hipEventElapsedTime(&ms, ev_rk[st][k], ev_rk[st][k + 1]));
          sweep_ms_accum[k + 1] += ms;
        }
        if (step2_split) {
          float ms = 0.f;
          HIP_CHECK(hipEventElapsedTime(&ms, ev_rk[st][0], ev_rk[st][8]));
          sweep_ms_accum[0] += ms;
        }
        ++sweep_updates_accum;
      }
    }
    return result;
  };

  if (cfl_recovery == RYUJIN_CFL_RECOVERY_BANG_BANG) {
    params.id_violation_strategy = RYUJIN_IDV_RAISE_EXCEPTION;
    params.cfl = cfl_max;
  }
  int result = single_step();
  /* Both accumulators hold kStageCode - (first stage that raised the flag), max-reduced over the ranks.
   * Under raise_exception the reference throws Restart at the end of the offending stage and never
   * evaluates tau_max of a later one (hyperbolic_module.template.h:1194-1207), whereas here all stages are
   * already enqueued and run on the inadmissible state: an invalid tau_max only counts ("We crashed",
   * :573-576) if it was found in a stage not later than the first Restart. */
  const int first_outcome = ryujin_hip_debug_rk_outcome(h_scalars->restart_accum, h_scalars->tau_invalid_accum,
                                                         params.id_violation_strategy);
  if (first_outcome == RYUJIN_ERR_TAU)
    return RYUJIN_ERR_TAU;
  int status = RYUJIN_OK;
  if (h_scalars->restart_accum) {
    if (params.id_violation_strategy == RYUJIN_IDV_RAISE_EXCEPTION) {
      n_restarts++;
      if (cfl_recovery != RYUJIN_CFL_RECOVERY_BANG_BANG)
        return RYUJIN_RESTART; /* the caller owns the recovery: state vector untouched */
      params.id_violation_strategy = RYUJIN_IDV_WARN;
      params.cfl = cfl_min;
      result = single_step();
      if (h_scalars->tau_invalid_accum)
        return RYUJIN_ERR_TAU;
      if (h_scalars->restart_accum) {
        n_warnings++;
        status = RYUJIN_WARN;
      }
    } else {
      n_warnings++;
      status = RYUJIN_WARN;
    }
  }
  /* state_vector.swap(temp_[..]): the caller's handle keeps naming the solution */
  std::swap(states[U], states[result]);
  *tau_out = tau_factor * h_scalars->tau_rk;
  return status;
}

/* ============================================================================ C ABI */

namespace
{
  template <typename F>
  int guarded(F &&f)
  {
    try {
      return f();
    } catch (const HipError &e) {
      g_error = e.what();
      return e.status;
    } catch (const std::invalid_argument &e) {
      g_error = e.what();
      return RYUJIN_ERR_ARG;
    } catch (const std::exception &e) {
      g_error = e.what();
      return RYUJIN_ERR_HIP;
    }
  }

  /* entry points that touch the device: make the context's device current first (a host thread may drive
   * contexts on several devices) */
  template <typename F>
  int guarded_ctx(ryujin_hip_ctx *ctx, F &&f)
  {
    const int status = guarded([&]() {
      if (!ctx)
        throw HipError(RYUJIN_ERR_ARG, "null context");
      HIP_CHECK(hipSetDevice(ctx->device));
      return f();
    });
    /* in-process transport: a rank that failed must not leave the rank threads of its group waiting for it */
    if (status < 0 && status != RYUJIN_ERR_TAU && ctx && ctx->comm && ctx->comm->local)
      ctx->comm->local->abort();
    return status;
  }

  template <typename E>
  struct EqTag {
    using type = E;
  };

  template <typename F>
  auto dispatch_equation(int equation, int dim, F &&f)
  {
    if (equation == RYUJIN_EQ_SHALLOW_WATER) {
      if (dim == 1)
        return f(EqTag<ShallowWater<1>>{});
      return f(EqTag<ShallowWater<2>>{});
    }
    if (equation == RYUJIN_EQ_SCALAR_CONSERVATION) {
      switch (dim) {
      case 1: return f(EqTag<ScalarConservation<1>>{});
      case 2: return f(EqTag<ScalarConservation<2>>{});
      default: return f(EqTag<ScalarConservation<3>>{});
      }
    }
    if (equation == RYUJIN_EQ_EULER_AEOS) {
      switch (dim) {
      case 1: return f(EqTag<EulerAeos<1>>{});
      case 2: return f(EqTag<EulerAeos<2>>{});
      default: return f(EqTag<EulerAeos<3>>{});
      }
    }
    switch (dim) {
    case 1: return f(EqTag<Euler<1>>{});
    case 2: return f(EqTag<Euler<2>>{});
    default: return f(EqTag<Euler<3>>{});
    }
  }
} // namespace

extern "C" {

const char *ryujin_hip_last_error(void)
{
  return g_error.c_str();
}

const char *ryujin_hip_version(void)
{
  return "ryujin_hip 0.4 (gfx950; Euler, Euler AEOS, shallow water, scalar conservation; SELL-64)";
}

void ryujin_hip_default_params(ryujin_hip_params *p, int equation, int dim)
{
  std::memset(p, 0, sizeof(*p));
  p->equation = equation;
  p->dim = dim;
  p->gamma = 7. / 5.;
  p->reference_density = 1.;
  p->vacuum_state_relaxation_small = 1.e2;
  p->vacuum_state_relaxation_large = 1.e4;
  p->gravity = 9.81;
  p->manning_friction_coefficient = 0.;
  p->reference_water_depth = 1.;
  p->dry_state_relaxation_factor = 2.e-1;
  p->dry_state_relaxation_small = 1.e2;
  p->dry_state_relaxation_large = 1.e4;
  p->cfl = 0.2;
  p->id_violation_strategy = RYUJIN_IDV_WARN;
  p->indicator_evc_factor = 1.;
  p->limiter_iterations = 2;
  p->limiter_newton_tolerance = 1.e-10;
  p->limiter_newton_max_iterations = 2;
  p->limiter_relaxation_factor = 1.;
  p->limiter_limit_on_kinetic_energy = 0;
  p->limiter_limit_on_square_velocity = 1;
  p->riemann_newton_max_iterations = 0;
  p->riemann_newton_tolerance = 1.e-10;
  p->eos = RYUJIN_EOS_POLYTROPIC_GAS;
  p->compute_strict_bounds = 1;
  p->eos_covolume_b = 0.;
  p->eos_q = 0.;
  p->eos_pinf = 0.;
  p->eos_vdw_a = 0.;
  p->eos_gas_constant_R = 287.052874;
  p->jwl_A = 6.3207e13;
  p->jwl_B = -4.472e9;
  p->jwl_R1 = 11.3;
  p->jwl_R2 = 1.13;
  p->jwl_omega = 0.8938;
  p->jwl_rho_0 = 1895.;
  p->jwl_q_0 = 0.;
  p->jwl_cv = 2487. / 1895.;
  p->sc_flux = RYUJIN_FLUX_BURGERS;
  p->sc_flux_polynomial[0][2] = 0.5; /* "0.5*u*u", the default expression of flux_function.h:32 */
  p->sc_derivative_approximation_delta = 1.e-10;
  p->sc_use_greedy_wavespeed = 0;
  p->sc_use_averaged_entropy = 0;
  p->sc_random_entropies = 0;
  p->system_scope_events = 0;
  p->debug_join_exchanges = 0;
  p->debug_bc_fold_max_slices = 0;
  p->debug_no_small_mesh_split = 0;
  p->debug_pij_storage = 0;
  p->debug_expensive_bounds_check = 0;
  p->debug_tile_map = 0;
  p->debug_band_stride = 0;
  p->debug_xcd_chunk = 0;
  /* the profiling scripts wrap bench.py, which takes the defaults: let them select a mapping without a flag */
  if (const char *e = std::getenv("RYUJIN_XCD_CHUNK"))
    p->debug_xcd_chunk = std::atoi(e);
  if (const char *e = std::getenv("RYUJIN_BAND_STRIDE"))
    p->debug_band_stride = std::atoi(e);
}

int ryujin_hip_comm_unique_id(char id[RYUJIN_HIP_UNIQUE_ID_BYTES])
{
  return guarded([&]() {
    static_assert(sizeof(ncclUniqueId) <= RYUJIN_HIP_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId uid;
    NCCL_CHECK(ncclGetUniqueId(&uid));
    std::memset(id, 0, RYUJIN_HIP_UNIQUE_ID_BYTES);
    std::memcpy(id, &uid, sizeof(uid));
    return RYUJIN_OK;
  });
}

int ryujin_hip_comm_init(ryujin_hip_comm **comm, const char id[RYUJIN_HIP_UNIQUE_ID_BYTES], int rank,
                         int n_ranks, int device)
{
  return guarded([&]() {
    auto c = std::make_unique<ryujin_hip_comm>();
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->device = device;
    HIP_CHECK(hipSetDevice(device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NCCL_CHECK(ncclCommInitRank(&c->comm, n_ranks, uid, rank));
    *comm = c.release();
    return RYUJIN_OK;
  });
}

int ryujin_hip_comm_init_local(ryujin_hip_comm **comms, int n_ranks, int device)
{
  return guarded([&]() {
    if (n_ranks < 1)
      throw HipError(RYUJIN_ERR_ARG, "n_ranks must be positive");
    auto *group = new LocalGroup(n_ranks, device);
    group->refs = n_ranks;
    for (int r = 0; r < n_ranks; ++r) {
      auto *c = new ryujin_hip_comm;
      c->rank = r;
      c->n_ranks = n_ranks;
      c->device = device;
      c->local = group;
      comms[r] = c;
    }
    return RYUJIN_OK;
  });
}

int ryujin_hip_comm_init_loopback(ryujin_hip_comm **comm, int rank, int n_ranks, int device)
{
  return guarded([&]() {
    if (!comm || n_ranks < 2 || rank < 0 || rank >= n_ranks)
      throw HipError(RYUJIN_ERR_ARG, "loopback communicator: 0 <= rank < n_ranks, n_ranks >= 2");
    auto *group = new LocalGroup(n_ranks, device);
    group->refs = 1;
    group->loopback = true;
    auto *c = new ryujin_hip_comm;
    c->rank = rank;
    c->n_ranks = n_ranks;
    c->device = device;
    c->local = group;
    *comm = c;
    return RYUJIN_OK;
  });
}

void ryujin_hip_comm_destroy(ryujin_hip_comm *comm)
{
  if (!comm)
    return;
  if (comm->local) {
    bool last;
    {
      std::lock_guard<std::mutex> lock(comm->local->mtx);
      last = --comm->local->refs == 0;
    }
    if (last)
      delete comm->local;
  }
  if (comm->comm)
    (void)ncclCommDestroy(comm->comm);
  delete comm;
}

int ryujin_hip_comm_info(const ryujin_hip_comm *comm, int *rank, int *n_ranks, int *rccl_rank, int *rccl_count,
                         int *rccl_device)
{
  return guarded([&]() {
    if (!comm)
      throw HipError(RYUJIN_ERR_ARG, "null communicator");
    int r = -1, n = -1, d = -1;
    if (comm->comm) {
      NCCL_CHECK(ncclCommUserRank(comm->comm, &r));
      NCCL_CHECK(ncclCommCount(comm->comm, &n));
      NCCL_CHECK(ncclCommCuDevice(comm->comm, &d));
    }
    if (rank)
      *rank = comm->rank;
    if (n_ranks)
      *n_ranks = comm->n_ranks;
    if (rccl_rank)
      *rccl_rank = r;
    if (rccl_count)
      *rccl_count = n;
    if (rccl_device)
      *rccl_device = d;
    return RYUJIN_OK;
  });
}

int ryujin_hip_exchange_info(ryujin_hip_ctx *ctx, int *n_nbr, int *nbr_rank, int max_nbr,
                             unsigned long long *n_exchanges, unsigned long long *n_allreduces)
{
  return guarded([&]() {
    if (!ctx)
      throw HipError(RYUJIN_ERR_ARG, "null context");
    if (n_nbr)
      *n_nbr = ctx->n_nbr;
    if (nbr_rank)
      for (int q = 0; q < ctx->n_nbr && q < max_nbr; ++q)
        nbr_rank[q] = ctx->nbr_rank[q];
    if (n_exchanges)
      *n_exchanges = ctx->n_exchanges;
    if (n_allreduces)
      *n_allreduces = ctx->n_allreduces;
    return RYUJIN_OK;
  });
}

int ryujin_hip_create(ryujin_hip_ctx **ctx, const ryujin_hip_offline *offline,
                      const ryujin_hip_params *params, ryujin_hip_comm *comm, int device)
{
  return guarded([&]() {
    if (!ctx || !offline || !params)
      throw HipError(RYUJIN_ERR_ARG, "null argument");
    auto c = std::make_unique<ryujin_hip_ctx>();
    c->create(*offline, *params, comm, device);
    *ctx = c.release();
    return RYUJIN_OK;
  });
}

void ryujin_hip_destroy(ryujin_hip_ctx *ctx)
{
  delete ctx;
}

int ryujin_hip_state_alloc(ryujin_hip_ctx *ctx, int *handle)
{
  return guarded_ctx(ctx, [&]() {
    int h = -1;
    for (size_t q = 0; q < ctx->states.size(); ++q)
      if (!ctx->states[q]->used) {
        h = (int)q;
        break;
      }
    if (h < 0) {
      ctx->states.push_back(std::make_unique<ryujin_hip_ctx::State>());
      h = (int)ctx->states.size() - 1;
      ctx->states[h]->U.alloc((size_t)ctx->L.n_relevant * ctx->KP);
      ctx->states[h]->prec.alloc((size_t)ctx->L.n_relevant * ctx->NPREC);
      if (ctx->params.equation == RYUJIN_EQ_EULER) /* Euler<dim>::RS: the combined node record in 3-D */
        ctx->states[h]->rrec.alloc((size_t)ctx->L.n_relevant * (ctx->dim == 3 ? Euler<3>::RS : ctx->dim == 2 ? Euler<2>::RS : Euler<1>::RS));
      else if (ctx->params.equation == RYUJIN_EQ_EULER_AEOS)
        ctx->states[h]->rrec.alloc((size_t)ctx->L.n_relevant * ((6 + ctx->dim + 1) / 2 * 2));
      else if (ctx->params.equation == RYUJIN_EQ_SHALLOW_WATER)
        ctx->states[h]->rrec.alloc((size_t)ctx->L.n_relevant * ((2 + ctx->dim + 1) / 2 * 2));
    }
    ctx->states[h]->used = true;
    ctx->states[h]->precomputed = false;
    *handle = h;
    return RYUJIN_OK;
  });
}

int ryujin_hip_state_free(ryujin_hip_ctx *ctx, int handle)
{
  return guarded_ctx(ctx, [&]() {
    ctx->state(handle).used = false;
    return RYUJIN_OK;
  });
}

int ryujin_hip_state_upload(ryujin_hip_ctx *ctx, int handle, const double *U_aos)
{
  return guarded_ctx(ctx, [&]() {
    auto &s = ctx->state(handle);
    const size_t n = ctx->L.n_relevant;
    const int K = ctx->K, KP = ctx->KP;
    ctx->finish();
    s.precomputed = false;
    if (K == KP) {
      HIP_CHECK(hipMemcpy(s.U.ptr, U_aos, n * K * sizeof(double), hipMemcpyHostToDevice));
    } else {
      std::vector<double> tmp(n * KP, 0.);
      for (size_t i = 0; i < n; ++i)
        std::memcpy(&tmp[i * KP], &U_aos[i * K], sizeof(double) * K);
      HIP_CHECK(hipMemcpy(s.U.ptr, tmp.data(), tmp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return RYUJIN_OK;
  });
}

int ryujin_hip_state_download(ryujin_hip_ctx *ctx, int handle, double *U_aos)
{
  return guarded_ctx(ctx, [&]() {
    auto &s = ctx->state(handle);
    const size_t n = ctx->L.n_relevant;
    const int K = ctx->K, KP = ctx->KP;
    ctx->finish();
    if (K == KP) {
      HIP_CHECK(hipMemcpy(U_aos, s.U.ptr, n * K * sizeof(double), hipMemcpyDeviceToHost));
    } else {
      std::vector<double> tmp(n * KP);
      HIP_CHECK(hipMemcpy(tmp.data(), s.U.ptr, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i)
        std::memcpy(&U_aos[i * K], &tmp[i * KP], sizeof(double) * K);
    }
    return RYUJIN_OK;
  });
}

int ryujin_hip_state_download_precomputed(ryujin_hip_ctx *ctx, int handle, double *prec_aos)
{
  return guarded_ctx(ctx, [&]() {
    auto &s = ctx->state(handle);
    ctx->finish();
    HIP_CHECK(hipMemcpy(prec_aos, s.prec.ptr, (size_t)ctx->L.n_relevant * ctx->NPREC * sizeof(double),
                        hipMemcpyDeviceToHost));
    return RYUJIN_OK;
  });
}

int ryujin_hip_host_register(ryujin_hip_ctx *ctx, const void *ptr, size_t bytes)
{
  return guarded_ctx(ctx, [&]() {
    if (!ptr || bytes == 0)
      throw HipError(RYUJIN_ERR_ARG, "host_register: null or empty range");
    /* a range that is registered already is registered AGAIN: the caller's allocator may have released the
     * memory and handed the same address out anew in the meantime, and the old pinning does not cover the new pages */
    const auto known = std::find(ctx->registered_host.begin(), ctx->registered_host.end(), ptr);
    if (known != ctx->registered_host.end()) {
      ctx->finish();
      if (hipHostUnregister(const_cast<void *>(ptr)) != hipSuccess)
        (void)hipGetLastError();
      ctx->registered_host.erase(known);
    }
    if (hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault) != hipSuccess) {
      (void)hipGetLastError(); /* not fatal: transfers of pageable memory are staged by the runtime */
      return RYUJIN_WARN;
    }
    ctx->registered_host.push_back(ptr);
    return RYUJIN_OK;
  });
}

int ryujin_hip_host_unregister(ryujin_hip_ctx *ctx, const void *ptr)
{
  return guarded_ctx(ctx, [&]() {
    const auto it = std::find(ctx->registered_host.begin(), ctx->registered_host.end(), ptr);
    if (it == ctx->registered_host.end())
      return RYUJIN_WARN;
    ctx->finish();
    HIP_CHECK(hipHostUnregister(const_cast<void *>(ptr)));
    ctx->registered_host.erase(it);
    return RYUJIN_OK;
  });
}

int ryujin_hip_state_download_owned(ryujin_hip_ctx *ctx, int handle, double *U_aos)
{
  return guarded_ctx(ctx, [&]() {
    auto &s = ctx->state(handle);
    const size_t n = ctx->L.n_owned;
    const int K = ctx->K, KP = ctx->KP;
    ctx->finish();
    if (K == KP) {
      HIP_CHECK(hipMemcpy(U_aos, s.U.ptr, n * K * sizeof(double), hipMemcpyDeviceToHost));
    } else {
      std::vector<double> tmp(n * KP);
      HIP_CHECK(hipMemcpy(tmp.data(), s.U.ptr, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i)
        std::memcpy(&U_aos[i * K], &tmp[i * KP], sizeof(double) * K);
    }
    return RYUJIN_OK;
  });
}

int ryujin_hip_state_download_prepared(ryujin_hip_ctx *ctx, int handle, double *U_aos)
{
  return guarded_ctx(ctx, [&]() {
    auto &s = ctx->state(handle);
    const int K = ctx->K, KP = ctx->KP;
    const uint32_t n_rows = (uint32_t)ctx->h_bc_rows.size();
    /* boundary rows inside export slices get their boundary conditions from the export part of the pre-pass, on
     * comm_stream (fold_bc): the pack kernel below reads them, so the compute stream joins comm_stream FIRST */
    ctx->wait_comm();
    if (n_rows != 0) {
      if (ctx->d_bc_rows.n == 0) {
        ctx->d_bc_rows.upload(ctx->h_bc_rows);
        ctx->d_bc_pack.alloc((size_t)n_rows * KP);
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_bc_pack), (size_t)n_rows * KP * sizeof(double)));
      }
      hipLaunchKernelGGL(k_pack_vector, dim3(grid_for((size_t)n_rows * KP)), dim3(kBlock), 0, ctx->stream, n_rows,
                         ctx->d_bc_rows.ptr, KP, s.U.ptr, ctx->d_bc_pack.ptr);
      HIP_CHECK(hipGetLastError());
    }
    ctx->finish();
    if (n_rows != 0) {
      HIP_CHECK(hipMemcpy(ctx->h_bc_pack, ctx->d_bc_pack.ptr, (size_t)n_rows * KP * sizeof(double),
                          hipMemcpyDeviceToHost));
      for (uint32_t q = 0; q < n_rows; ++q)
        std::memcpy(&U_aos[(size_t)ctx->h_bc_rows[q] * K], &ctx->h_bc_pack[(size_t)q * KP], sizeof(double) * K);
    }
    /* the ghost range [n_owned, n_relevant) */
    const size_t first = ctx->L.n_owned, n = ctx->L.n_relevant - ctx->L.n_owned;
    if (n != 0) {
      if (K == KP) {
        HIP_CHECK(hipMemcpy(U_aos + first * K, s.U.ptr + first * K, n * K * sizeof(double), hipMemcpyDeviceToHost));
      } else {
        std::vector<double> tmp(n * KP);
        HIP_CHECK(hipMemcpy(tmp.data(), s.U.ptr + first * KP, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i)
          std::memcpy(&U_aos[(first + i) * K], &tmp[i * KP], sizeof(double) * K);
      }
    }
    return RYUJIN_OK;
  });
}

int ryujin_hip_prepare_state_vector(ryujin_hip_ctx *ctx, int handle, double /*t*/,
                                    const double *dirichlet_aos)
{
  return guarded_ctx(ctx, [&]() {
    dispatch_equation(ctx->params.equation, ctx->dim, [&](auto tag) {
      ctx->template prepare_state_vector<typename decltype(tag)::type>(handle, dirichlet_aos);
      return 0;
    });
    return RYUJIN_OK;
  });
}

int ryujin_hip_step(ryujin_hip_ctx *ctx, int h_old, int stages, const int *h_stage,
                    const double *stage_weights, int h_new, double tau_in, double tau_max_in,
                    double *tau_out)
{
  return guarded_ctx(ctx, [&]() {
    if (stages < 0 || stages > 4 || !tau_out)
      throw HipError(RYUJIN_ERR_ARG, "stages must be in [0,4]");
    /* tau_max = min(tau_max argument, CFL bound) must be positive and finite (:571-576). The device
     * minimum works on the bit patterns of positive doubles, so a NaN or non-positive argument -- which
     * makes the reference throw whatever the CFL bound is -- is caught here. */
    if (std::isnan(tau_max_in) || !(tau_max_in > 0.))
      return RYUJIN_ERR_TAU;
    return dispatch_equation(ctx->params.equation, ctx->dim, [&](auto tag) {
      return ctx->template step<typename decltype(tag)::type>(h_old, stages, h_stage, stage_weights,
                                                              h_new, tau_in, tau_max_in, tau_out);
    });
  });
}

int ryujin_hip_time_step_n(ryujin_hip_ctx *ctx, int scheme, int h_state, int n_tmp, const int *h_tmp,
                           const double *dirichlet_aos, double tau_max, int cfl_recovery,
                           double cfl_min, double cfl_max, double *tau_out)
{
  return guarded_ctx(ctx, [&]() {
    if (!h_tmp || !tau_out || n_tmp < 1 || n_tmp > 8)
      throw HipError(RYUJIN_ERR_ARG, "time_step: bad argument");
    if (std::isnan(tau_max) || !(tau_max > 0.))
      return RYUJIN_ERR_TAU; /* as in step() */
    ctx->state(h_state);
    for (int q = 0; q < n_tmp; ++q) {
      ctx->state(h_tmp[q]);
      if (h_tmp[q] == h_state)
        throw HipError(RYUJIN_ERR_ARG, "time_step: temporary vectors must differ from the state vector");
      for (int r = 0; r < q; ++r)
        if (h_tmp[r] == h_tmp[q])
          throw HipError(RYUJIN_ERR_ARG, "time_step: temporary vectors must be distinct");
    }
    return dispatch_equation(ctx->params.equation, ctx->dim, [&](auto tag) {
      return ctx->template time_step<typename decltype(tag)::type>(scheme, h_state, n_tmp, h_tmp,
                                                                   dirichlet_aos, tau_max, cfl_recovery,
                                                                   cfl_min, cfl_max, tau_out);
    });
  });
}

int ryujin_hip_time_step_fn(ryujin_hip_ctx *ctx, int scheme, int h_state, int n_tmp, const int *h_tmp, double t,
                            ryujin_hip_dirichlet_fn dirichlet_fn, void *user, double tau_max, int cfl_recovery,
                            double cfl_min, double cfl_max, double *tau_out)
{
  return guarded_ctx(ctx, [&]() {
    if (!h_tmp || !tau_out || n_tmp < 1 || n_tmp > 8)
      throw HipError(RYUJIN_ERR_ARG, "time_step: bad argument");
    if (std::isnan(tau_max) || !(tau_max > 0.))
      return RYUJIN_ERR_TAU; /* as in step() */
    ctx->state(h_state);
    for (int q = 0; q < n_tmp; ++q) {
      ctx->state(h_tmp[q]);
      if (h_tmp[q] == h_state)
        throw HipError(RYUJIN_ERR_ARG, "time_step: temporary vectors must differ from the state vector");
      for (int r = 0; r < q; ++r)
        if (h_tmp[r] == h_tmp[q])
          throw HipError(RYUJIN_ERR_ARG, "time_step: temporary vectors must be distinct");
    }
    return dispatch_equation(ctx->params.equation, ctx->dim, [&](auto tag) {
      return ctx->template time_step<typename decltype(tag)::type>(scheme, h_state, n_tmp, h_tmp, nullptr, tau_max,
                                                                   cfl_recovery, cfl_min, cfl_max, tau_out, t,
                                                                   dirichlet_fn, user);
    });
  });
}

int ryujin_hip_device_count(int *n_devices)
{
  return guarded([&]() {
    if (!n_devices)
      throw HipError(RYUJIN_ERR_ARG, "null argument");
    HIP_CHECK(hipGetDeviceCount(n_devices));
    return RYUJIN_OK;
  });
}

int ryujin_hip_time_step(ryujin_hip_ctx *ctx, int scheme, int h_state, const int h_tmp[3],
                         const double *dirichlet_aos, double tau_max, int cfl_recovery, double cfl_min,
                         double cfl_max, double *tau_out)
{
  return ryujin_hip_time_step_n(ctx, scheme, h_state, 3, h_tmp, dirichlet_aos, tau_max, cfl_recovery,
                                cfl_min, cfl_max, tau_out);
}

int ryujin_hip_get_timers_accum(ryujin_hip_ctx *ctx, double ms[8], unsigned *n_updates, int reset)
{
  return guarded([&]() {
    if (!ctx || !ms || !n_updates)
      throw HipError(RYUJIN_ERR_ARG, "null argument");
    for (int k = 0; k < 8; ++k)
      ms[k] = ctx->sweep_ms_accum[k];
    *n_updates = ctx->sweep_updates_accum;
    if (reset) {
      for (auto &v : ctx->sweep_ms_accum)
        v = 0.;
      ctx->sweep_updates_accum = 0;
    }
    return RYUJIN_OK;
  });
}

int ryujin_hip_sadd(ryujin_hip_ctx *ctx, int h_dst, double s, double b, int h_src)
{
  return guarded_ctx(ctx, [&]() {
    auto &dst = ctx->state(h_dst);
    auto &src = ctx->state(h_src);
    const size_t n = (size_t)ctx->L.n_relevant * ctx->KP;
    dst.precomputed = false;
    ctx->wait_comm();
    const int grid = (int)std::min<size_t>(2048, (n / 2 + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(k_sadd, dim3(std::max(1, grid)), dim3(kBlock), 0, ctx->stream, n, s, b,
                       dst.U.ptr, src.U.ptr);
    HIP_CHECK(hipGetLastError());
    return RYUJIN_OK;
  });
}

int ryujin_hip_set_cfl(ryujin_hip_ctx *ctx, double cfl)
{
  ctx->params.cfl = cfl;
  return RYUJIN_OK;
}

int ryujin_hip_get_cfl(ryujin_hip_ctx *ctx, double *cfl)
{
  *cfl = ctx->params.cfl;
  return RYUJIN_OK;
}

int ryujin_hip_set_id_violation_strategy(ryujin_hip_ctx *ctx, int strategy)
{
  ctx->params.id_violation_strategy = strategy;
  return RYUJIN_OK;
}

int ryujin_hip_get_alpha(ryujin_hip_ctx *ctx, double *alpha)
{
  return guarded_ctx(ctx, [&]() {
    ctx->finish();
    HIP_CHECK(hipMemcpy(alpha, ctx->d_alpha.ptr, (size_t)ctx->L.n_relevant * sizeof(double),
                        hipMemcpyDeviceToHost));
    return RYUJIN_OK;
  });
}

int ryujin_hip_state_integrals(ryujin_hip_ctx *ctx, int handle, double *out)
{
  return guarded_ctx(ctx, [&]() {
    if (!out)
      throw HipError(RYUJIN_ERR_ARG, "null argument");
    auto &st = ctx->state(handle);
    ctx->wait_comm();
    constexpr uint32_t n_blocks = 512;
    const int K = ctx->K;
    if (ctx->d_integrals.n < (size_t)(n_blocks + 1) * 8)
      ctx->d_integrals.alloc((size_t)(n_blocks + 1) * 8);
    double *partial = ctx->d_integrals.ptr, *result = ctx->d_integrals.ptr + (size_t)n_blocks * 8;
    auto launch = [&](auto tag) {
      constexpr int KK = decltype(tag)::value;
      hipLaunchKernelGGL(k_integrals_partial<KK>, dim3(n_blocks), dim3(kBlock), 0, ctx->stream,
                         ctx->L.n_owned, ctx->d_mi.ptr, st.U.ptr, partial);
      hipLaunchKernelGGL(k_integrals_final<KK>, dim3(1), dim3(64), 0, ctx->stream, n_blocks, partial,
                         result);
    };
    switch (K) {
    case 1: launch(std::integral_constant<int, 1>{}); break;
    case 2: launch(std::integral_constant<int, 2>{}); break;
    case 3: launch(std::integral_constant<int, 3>{}); break;
    case 4: launch(std::integral_constant<int, 4>{}); break;
    default: launch(std::integral_constant<int, 5>{}); break;
    }
    HIP_CHECK(hipGetLastError());
    const ryujin_hip_comm *cm = ctx->comm;
    if (cm && cm->n_ranks > 1 && !cm->local)
      NCCL_CHECK(ncclAllReduce(result, result, K, ncclDouble, ncclSum, cm->comm, ctx->stream));
    double host[8] = {0.};
    HIP_CHECK(hipMemcpyAsync(host, result, sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (cm && cm->n_ranks > 1 && cm->local && !cm->local->loopback) {
      LocalGroup &g = *cm->local;
      for (int q = 0; q < K; ++q)
        g.scratch_vec[(size_t)cm->rank * 8 + q] = host[q];
      g.barrier();
      for (int q = 0; q < K; ++q) {
        double v = 0.;
        for (int r = 0; r < g.n_ranks; ++r)
          v += g.scratch_vec[(size_t)r * 8 + q];
        host[q] = v;
      }
      g.barrier();
    }
    for (int q = 0; q < K; ++q)
      out[q] = host[q];
    return RYUJIN_OK;
  });
}

int ryujin_hip_get_counters(ryujin_hip_ctx *ctx, unsigned *n_restarts, unsigned *n_warnings)
{
  *n_restarts = ctx->n_restarts;
  *n_warnings = ctx->n_warnings;
  return RYUJIN_OK;
}

int ryujin_hip_limiter_statistics(ryujin_hip_ctx *ctx, double *limited_slice_fraction, int *pij_stored,
                                  double *stored_slice_fraction)
{
  return guarded([&]() {
    if (!ctx)
      throw HipError(RYUJIN_ERR_ARG, "null context");
    if (limited_slice_fraction)
      *limited_slice_fraction = ctx->limited_fraction;
    if (pij_stored)
      *pij_stored = ctx->last_per_slice ? 2 : (ctx->last_tile_store ? 3 : 1);
    if (stored_slice_fraction)
      *stored_slice_fraction = ctx->stored_fraction;
    return RYUJIN_OK;
  });
}

int ryujin_hip_tile_statistics(ryujin_hip_ctx *ctx, double *stored_fraction, double *read_fraction,
                               double *formed_by_step6_fraction)
{
  return guarded([&]() {
    if (!ctx)
      throw HipError(RYUJIN_ERR_ARG, "null context");
    const bool tiles = ctx->last_tile_store;
    if (stored_fraction)
      *stored_fraction = tiles ? ctx->stored_fraction : 1.;
    if (read_fraction)
      *read_fraction = tiles ? ctx->tiles_needed_fraction : 1.;
    if (formed_by_step6_fraction)
      *formed_by_step6_fraction = tiles ? ctx->tiles_formed_fraction : 0.;
    return RYUJIN_OK;
  });
}

int ryujin_hip_deferred_slices(ryujin_hip_ctx *ctx, unsigned *n_slices)
{
  return guarded([&]() {
    if (!ctx || !n_slices)
      throw HipError(RYUJIN_ERR_ARG, "null argument");
    *n_slices = ctx->h_scalars ? ctx->h_scalars->n_deferred[0] + ctx->h_scalars->n_deferred[1] : 0u;
    return RYUJIN_OK;
  });
}

int ryujin_hip_layout_info(ryujin_hip_ctx *ctx, unsigned long long *n_tiles, unsigned long long *n_regular_tiles)
{
  return guarded([&]() {
    if (!ctx)
      throw HipError(RYUJIN_ERR_ARG, "null context");
    if (n_tiles)
      *n_tiles = ctx->L.slice_off[ctx->L.n_slices];
    if (n_regular_tiles)
      *n_regular_tiles = (ctx->d_tiles.n != 0 && tile_map_pays(ctx->dim)) ? ctx->L.n_regular_tiles : 0ull;
    return RYUJIN_OK;
  });
}

int ryujin_hip_chain_info(ryujin_hip_ctx *ctx, unsigned long long *n_chained_tiles,
                          unsigned long long *n_chained_entries)
{
  return guarded([&]() {
    if (!ctx)
      throw HipError(RYUJIN_ERR_ARG, "null context");
    /* (what the sweeps of this dimension use: every chained tile with its mask in 3-D, below that the tiles in which
     * only the lane at the end of the wave loads -- 63 entries each; chain_masks_pay(), kernels_euler.hpp) */
    const bool masks = ctx->dim == 3;
    if (n_chained_tiles)
      *n_chained_tiles = ctx->d_tiles.n == 0 ? 0ull : (masks ? ctx->L.n_chained_tiles : ctx->L.n_end_lane_tiles);
    if (n_chained_entries)
      *n_chained_entries = ctx->d_tiles.n == 0 ? 0ull : (masks ? ctx->L.n_chained_entries : 63ull * ctx->L.n_end_lane_tiles);
    return RYUJIN_OK;
  });
}

int ryujin_hip_debug_fetch(ryujin_hip_ctx *ctx, int what, double *out, size_t n_doubles)
{
  return guarded_ctx(ctx, [&]() {
    const auto &L = ctx->L;
    ctx->finish();
    auto fetch_matrix = [&](const double *dev, uint32_t n_comp) {
      if (n_doubles < L.nnz_owned_logical * n_comp)
        throw HipError(RYUJIN_ERR_ARG, "output buffer too small");
      std::vector<double> tmp(L.nnz_total * n_comp);
      HIP_CHECK(hipMemcpy(tmp.data(), dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
      L.gather_logical(tmp, n_comp, out);
    };
    switch (what) {
    case 0: fetch_matrix(ctx->d_dij.ptr, 1); break;
    case 1: fetch_matrix(ctx->d_lij.ptr, 1); break;
    case 2:
      ctx->ensure_pij();
      if (ctx->last_per_slice || ctx->last_tile_store)
        dispatch_equation(ctx->params.equation, ctx->dim, [&](auto tag) {
          using E = typename decltype(tag)::type;
          if constexpr (std::is_same<typename E::Params, EulerParams>::value ||
                        std::is_same<typename E::Params, EulerAeosParams>::value)
            ctx->template store_pij_for_debug<E>();
          return 0;
        });
      fetch_matrix(ctx->d_pij.ptr, (uint32_t)ctx->K);
      break;
    case 5: fetch_matrix(ctx->d_lij_next.ptr, 1); break;
    case 6:
    case 7:
    case 8: { /* plain CSR over ALL locally relevant rows: owned rows, then the ghost rows as received */
      const double *dev = what == 6 ? ctx->d_dij.ptr : (what == 7 ? ctx->d_lij.ptr : ctx->d_lij_next.ptr);
      const uint64_t n_ghost_entries = L.nnz_total - L.nnz_sell;
      if (n_doubles < L.nnz_owned_logical + n_ghost_entries)
        throw HipError(RYUJIN_ERR_ARG, "output buffer too small");
      std::vector<double> tmp(L.nnz_total);
      HIP_CHECK(hipMemcpy(tmp.data(), dev, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
      L.gather_logical(tmp, 1, out);
      std::copy(tmp.begin() + L.nnz_sell, tmp.end(), out + L.nnz_owned_logical);
      break;
    }
    case 9: {
      if (n_doubles < (size_t)L.n_relevant * ctx->K)
        throw HipError(RYUJIN_ERR_ARG, "output buffer too small");
      std::vector<double> tmp((size_t)L.n_relevant * ctx->KP);
      HIP_CHECK(hipMemcpy(tmp.data(), ctx->d_r.ptr, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < L.n_relevant; ++i)
        for (int q = 0; q < ctx->K; ++q)
          out[(size_t)i * ctx->K + q] = tmp[(size_t)i * ctx->KP + q];
      break;
    }
    case 3: {
      if (n_doubles < (size_t)L.n_owned * ctx->NB)
        throw HipError(RYUJIN_ERR_ARG, "output buffer too small");
      std::vector<double> tmp((size_t)ctx->NB * ctx->bounds_stride);
      HIP_CHECK(hipMemcpy(tmp.data(), ctx->d_bounds.ptr, tmp.size() * sizeof(double),
                          hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < L.n_owned; ++i)
        for (int b = 0; b < ctx->NB; ++b)
          out[(size_t)i * ctx->NB + b] = tmp[(size_t)b * ctx->bounds_stride + i];
      break;
    }
    case 4: {
      if (n_doubles < (size_t)L.n_owned * ctx->K)
        throw HipError(RYUJIN_ERR_ARG, "output buffer too small");
      std::vector<double> tmp((size_t)L.n_relevant * ctx->KP);
      HIP_CHECK(hipMemcpy(tmp.data(), ctx->d_r.ptr, tmp.size() * sizeof(double),
                          hipMemcpyDeviceToHost));
      for (uint32_t i = 0; i < L.n_owned; ++i)
        for (int q = 0; q < ctx->K; ++q)
          out[(size_t)i * ctx->K + q] = tmp[(size_t)i * ctx->KP + q];
      break;
    }
    default: throw HipError(RYUJIN_ERR_ARG, "unknown debug_fetch selector");
    }
    return RYUJIN_OK;
  });
}

int ryujin_hip_debug_addresses(ryujin_hip_ctx *ctx, uint64_t out[8])
{
  return guarded([&]() {
    if (!ctx || !out)
      throw HipError(RYUJIN_ERR_ARG, "null argument");
    ctx->ensure_pij();
    const void *p[8] = {ctx->d_cols.ptr, ctx->d_cij.ptr, ctx->d_mij.ptr, ctx->d_dij.ptr,
                        ctx->d_lij.ptr,  ctx->d_lij_next.ptr, ctx->d_pij.ptr, ctx->d_idx_t.ptr};
    for (int q = 0; q < 8; ++q)
      out[q] = (uint64_t)(uintptr_t)p[q];
    return RYUJIN_OK;
  });
}

int ryujin_hip_debug_layout(const ryujin_hip_offline *offline, uint64_t *ptr, uint32_t *col,
                            uint64_t *transposed, const double *data, uint32_t n_comp, double *out)
{
  return guarded([&]() {
    SellLayout L;
    L.build(*offline);
    const RefView ref(*offline);
    std::vector<uint64_t> lptr((size_t)L.n_relevant + 1, 0);
    for (uint32_t i = 0; i < L.n_relevant; ++i)
      lptr[i + 1] = lptr[i] + ref.row_length(i);
    /* device position -> logical index */
    std::vector<uint64_t> logical_of(L.nnz_total, ~uint64_t(0));
    for (uint32_t i = 0; i < L.n_relevant; ++i)
      for (uint32_t c = 0; c < ref.row_length(i); ++c)
        logical_of[L.pos(i, c)] = lptr[i] + c;
    std::vector<double> dev;
    if (data && out)
      dev = L.scatter(*offline, data, n_comp);
    for (uint32_t i = 0; i < L.n_relevant; ++i)
      for (uint32_t c = 0; c < ref.row_length(i); ++c) {
        const uint64_t p = L.pos(i, c), e = lptr[i] + c;
        if (col)
          col[e] = L.cols[p];
        if (transposed)
          transposed[e] = logical_of[L.idx_t[p]];
        if (data && out)
          for (uint32_t d = 0; d < n_comp; ++d)
            out[e * n_comp + d] = dev[L.comp_pos(p, n_comp, d)];
      }
    if (ptr)
      std::copy(lptr.begin(), lptr.end(), ptr);
    return RYUJIN_OK;
  });
}

int ryujin_hip_debug_pow(int device, const double *x, const double *y, double *out, size_t n)
{
  return guarded([&]() {
    HIP_CHECK(hipSetDevice(device));
    DeviceBuffer<double> dx, dy, dout;
    dx.upload(x, n);
    dy.upload(y, n);
    dout.alloc(n);
    hipLaunchKernelGGL(k_debug_pow, dim3(grid_for(n)), dim3(kBlock), 0, nullptr, n, dx.ptr, dy.ptr,
                       dout.ptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, dout.ptr, n * sizeof(double), hipMemcpyDeviceToHost));
    return RYUJIN_OK;
  });
}

/* device functions evaluated on n independent items (parity tests against the reference's unit-test
 * baselines: tests/euler/riemann_solver.cc, tests/euler/limiter.cc, tests/shallow_water/riemann_solver.cc) */
namespace
{
  __global__ void __launch_bounds__(kBlock)
  k_debug_function(const EulerParams PE, const ShallowWaterParams PS, const EulerAeosParams PA, const int which,
                   const size_t n, const double *__restrict__ in, double *__restrict__ out)
  {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n)
      return;
    if (which == RYUJIN_DEBUG_AEOS_RIEMANN) {
      /* through the PRODUCTION evaluation path: per-node records, dij_from_records with n = (1) */
      using A = EulerAeos<1>;
      const double *v = in + q * 10;
      const A::RiemannData rd_i = A::make_riemann_data(PA, v[0], v[1], v[2], v[3], v[4]),
                           rd_j = A::make_riemann_data(PA, v[5], v[6], v[7], v[8], v[9]);
      const double r_i[A::RS] = {rd_i.rho, rd_i.p, rd_i.gamma, rd_i.a, rd_i.alpha, rd_i.alpha_hat, rd_i.u, 0.},
                   r_j[A::RS] = {rd_j.rho, rd_j.p, rd_j.gamma, rd_j.a, rd_j.alpha, rd_j.alpha_hat, rd_j.u, 0.};
      const double n[1] = {1.};
      out[q] = A::dij_from_records(PA, r_i, r_j, n);
      return;
    }
    if (which == RYUJIN_DEBUG_AEOS_DIJ_2D || which == RYUJIN_DEBUG_AEOS_DIJ_RECORDS_2D) {
      using A = EulerAeos<2>;
      const double *v = in + q * 10;
      const double U_i[4] = {v[0], v[1], v[2], v[3]}, U_j[4] = {v[4], v[5], v[6], v[7]}, c[2] = {v[8], v[9]};
      const double p_i = A::precompute_cycle0(PA, U_i).p, p_j = A::precompute_cycle0(PA, U_j).p;
      if (which == RYUJIN_DEBUG_AEOS_DIJ_2D) {
        out[q] = A::dij_from_states(PA, U_i, p_i, U_j, p_j, c);
      } else {
        double r_i[A::RS], r_j[A::RS];
        A::riemann_record(PA, U_i, p_i, r_i);
        A::riemann_record(PA, U_j, p_j, r_j);
        out[q] = A::dij_from_records(PA, r_i, r_j, c);
      }
      return;
    }
    if (which == RYUJIN_DEBUG_AEOS_LIMIT_1D) {
      /* the composition the sweeps use: limit_fast(), and limit() for the undecided pairs */
      const double *v = in + q * 10;
      const double bnd[4] = {v[0], v[1], v[2], v[3]}, U[3] = {v[4], v[5], v[6]}, Pij[3] = {v[7], v[8], v[9]};
      bool success, undecided;
      double l = EulerAeos<1>::limit_fast(PA, bnd, U, Pij, success, undecided);
      if (undecided)
        l = EulerAeos<1>::limit(PA, bnd, U, Pij, success);
      out[q * 3 + 0] = l;
      out[q * 3 + 1] = success ? 1. : 0.;
      out[q * 3 + 2] = undecided ? 1. : 0.;
      return;
    }
    if (which == RYUJIN_DEBUG_EULER_RIEMANN) {
      const double *v = in + q * 8;
      const Euler<1>::RiemannData rd_i{v[0], v[1], v[2], v[3]}, rd_j{v[4], v[5], v[6], v[7]};
      out[q] = Euler<1>::riemann_compute(PE, rd_i, rd_j);
    } else if (which == RYUJIN_DEBUG_EULER_LIMIT_1D) {
      /* the composition the sweeps use: limit_fast(), and limit() for the undecided pairs */
      const double *v = in + q * 9;
      const double bnd[3] = {v[0], v[1], v[2]}, U[3] = {v[3], v[4], v[5]}, Pij[3] = {v[6], v[7], v[8]};
      bool success, undecided;
      double l = Euler<1>::limit_fast(PE, bnd, U, Pij, success, undecided);
      if (undecided)
        l = Euler<1>::limit(PE, bnd, U, Pij, success);
      out[q * 3 + 0] = l;
      out[q * 3 + 1] = success ? 1. : 0.;
      out[q * 3 + 2] = undecided ? 1. : 0.;
    } else if (which == RYUJIN_DEBUG_EULER_LIMIT_2D) {
      EulerParams P2 = PE; /* (the parameter block carries no dimension) */
      const double *v = in + q * 11;
      const double bnd[3] = {v[0], v[1], v[2]}, U[4] = {v[3], v[4], v[5], v[6]}, Pij[4] = {v[7], v[8], v[9], v[10]};
      bool success, undecided, s2;
      double l = Euler<2>::limit_fast(P2, bnd, U, Pij, success, undecided);
      if (undecided)
        l = Euler<2>::limit(P2, bnd, U, Pij, success);
      double t_r;
      const double psi_r = Euler<2>::first_psi_r(P2, bnd, U, Pij, s2, t_r);
      out[q * 5 + 0] = l;
      out[q * 5 + 1] = success ? 1. : 0.;
      out[q * 5 + 2] = undecided ? 1. : 0.;
      out[q * 5 + 3] = t_r;
      out[q * 5 + 4] = psi_r;
    } else if (which == RYUJIN_DEBUG_EULER_LIMIT_CHECKED_1D) {
      const double *v = in + q * 9;
      const double bnd[3] = {v[0], v[1], v[2]}, U[3] = {v[3], v[4], v[5]}, Pij[3] = {v[6], v[7], v[8]};
      bool success;
      out[q * 3 + 0] = Euler<1>::limit_checked(PE, bnd, U, Pij, success);
      out[q * 3 + 1] = success ? 1. : 0.;
      out[q * 3 + 2] = 0.;
    } else if (which == RYUJIN_DEBUG_SW_RIEMANN) {
      const double *v = in + q * 6;
      const ShallowWater<1>::RiemannData rd_i{v[0], v[1], v[2]}, rd_j{v[3], v[4], v[5]};
      out[q * 2 + 0] = ShallowWater<1>::compute_h_star(PS, rd_i, rd_j);
      out[q * 2 + 1] = ShallowWater<1>::lambda_max(PS, rd_i, rd_j);
    } else if (which == RYUJIN_DEBUG_EULER_DIJ_2D) {
      const double *v = in + q * 10;
      const double U_i[4] = {v[0], v[1], v[2], v[3]}, U_j[4] = {v[4], v[5], v[6], v[7]}, c[2] = {v[8], v[9]};
      out[q] = Euler<2>::dij_from_states(PE, U_i, U_j, c);
    } else if (which == RYUJIN_DEBUG_EULER_DIJ_3D) {
      const double *v = in + q * 13;
      const double U_i[5] = {v[0], v[1], v[2], v[3], v[4]}, U_j[5] = {v[5], v[6], v[7], v[8], v[9]},
                   c[3] = {v[10], v[11], v[12]};
      out[q] = Euler<3>::dij_from_states(PE, U_i, U_j, c);
    } else if (which == RYUJIN_DEBUG_EULER_DIJ_RECORDS_2D) {
      const double *v = in + q * 10;
      const double U_i[4] = {v[0], v[1], v[2], v[3]}, U_j[4] = {v[4], v[5], v[6], v[7]}, c[2] = {v[8], v[9]};
      double r_i[Euler<2>::RS], r_j[Euler<2>::RS];
      Euler<2>::riemann_record(PE, U_i, r_i);
      Euler<2>::riemann_record(PE, U_j, r_j);
      out[q] = PE.riemann_newton_max_iterations == 0 && PE.rarefaction_power > 0
                   ? Euler<2>::dij_from_records<false>(PE, r_i, r_j, c)
                   : Euler<2>::dij_from_records<true>(PE, r_i, r_j, c);
    } else if (which == RYUJIN_DEBUG_EULER_DIJ_RECORDS_3D) {
      const double *v = in + q * 13;
      const double U_i[5] = {v[0], v[1], v[2], v[3], v[4]}, U_j[5] = {v[5], v[6], v[7], v[8], v[9]},
                   c[3] = {v[10], v[11], v[12]};
      double r_i[Euler<3>::RS], r_j[Euler<3>::RS];
      Euler<3>::riemann_record(PE, U_i, r_i);
      Euler<3>::riemann_record(PE, U_j, r_j);
      out[q] = PE.riemann_newton_max_iterations == 0 && PE.rarefaction_power > 0
                   ? Euler<3>::dij_from_records<false>(PE, r_i, r_j, c)
                   : Euler<3>::dij_from_records<true>(PE, r_i, r_j, c);
    } else if (which == RYUJIN_DEBUG_EULER_RIEMANN_RECORDS) {
      /* the PRODUCTION evaluation path (per-node records, dij_from_records) on the reference's Riemann data */
      const double *v = in + q * 8;
      double r_i[Euler<1>::RS], r_j[Euler<1>::RS];
      const double u_i[1] = {v[1]}, u_j[1] = {v[5]}, n[1] = {1.};
      Euler<1>::riemann_record_from_primitive(PE, v[0], v[2], v[3], u_i, r_i);
      Euler<1>::riemann_record_from_primitive(PE, v[4], v[6], v[7], u_j, r_j);
      out[q] = PE.riemann_newton_max_iterations == 0 && PE.rarefaction_power > 0
                   ? Euler<1>::dij_from_records<false>(PE, r_i, r_j, n)
                   : Euler<1>::dij_from_records<true>(PE, r_i, r_j, n);
    } else if (which == RYUJIN_DEBUG_SW_RIEMANN_RECORDS) {
      const double *v = in + q * 6;
      const double r_i[ShallowWater<1>::RS] = {v[0], v[2], v[1], 0.}, r_j[ShallowWater<1>::RS] = {v[3], v[5], v[4], 0.};
      const double n[1] = {1.};
      out[q] = ShallowWater<1>::dij_from_records<false>(PS, r_i, r_j, n);
    } else if (which == RYUJIN_DEBUG_SW_DIJ_2D || which == RYUJIN_DEBUG_SW_DIJ_RECORDS_2D) {
      const double *v = in + q * 8;
      const double U_i[3] = {v[0], v[1], v[2]}, U_j[3] = {v[3], v[4], v[5]}, c[2] = {v[6], v[7]};
      if (which == RYUJIN_DEBUG_SW_DIJ_2D) {
        out[q] = ShallowWater<2>::dij_from_states(PS, U_i, U_j, c);
      } else {
        double r_i[ShallowWater<2>::RS], r_j[ShallowWater<2>::RS];
        ShallowWater<2>::riemann_record(PS, U_i, r_i);
        ShallowWater<2>::riemann_record(PS, U_j, r_j);
        out[q] = ShallowWater<2>::dij_from_records<false>(PS, r_i, r_j, c);
      }
    }
  }
} // namespace

int ryujin_hip_debug_function(int device, const ryujin_hip_params *params, int which, const double *in,
                              double *out, size_t n)
{
  return guarded([&]() {
    if (!params || !in || !out)
      throw HipError(RYUJIN_ERR_ARG, "null argument");
    size_t n_in, n_out;
    switch (which) {
    case RYUJIN_DEBUG_EULER_RIEMANN:
    case RYUJIN_DEBUG_EULER_RIEMANN_RECORDS: n_in = 8; n_out = 1; break;
    case RYUJIN_DEBUG_SW_RIEMANN_RECORDS: n_in = 6; n_out = 1; break;
    case RYUJIN_DEBUG_AEOS_RIEMANN:
    case RYUJIN_DEBUG_AEOS_DIJ_2D:
    case RYUJIN_DEBUG_AEOS_DIJ_RECORDS_2D: n_in = 10; n_out = 1; break;
    case RYUJIN_DEBUG_AEOS_LIMIT_1D: n_in = 10; n_out = 3; break;
    case RYUJIN_DEBUG_EULER_LIMIT_1D:
    case RYUJIN_DEBUG_EULER_LIMIT_CHECKED_1D: n_in = 9; n_out = 3; break;
    case RYUJIN_DEBUG_EULER_LIMIT_2D: n_in = 11; n_out = 5; break;
    case RYUJIN_DEBUG_SW_RIEMANN: n_in = 6; n_out = 2; break;
    case RYUJIN_DEBUG_EULER_DIJ_2D:
    case RYUJIN_DEBUG_EULER_DIJ_RECORDS_2D: n_in = 10; n_out = 1; break;
    case RYUJIN_DEBUG_EULER_DIJ_3D:
    case RYUJIN_DEBUG_EULER_DIJ_RECORDS_3D: n_in = 13; n_out = 1; break;
    case RYUJIN_DEBUG_SW_DIJ_2D:
    case RYUJIN_DEBUG_SW_DIJ_RECORDS_2D: n_in = 8; n_out = 1; break;
    default: throw HipError(RYUJIN_ERR_ARG, "unknown debug function");
    }
    HIP_CHECK(hipSetDevice(device));
    DeviceBuffer<double> din, dout;
    din.upload(in, n * n_in);
    dout.alloc(n * n_out);
    hipLaunchKernelGGL(k_debug_function, dim3(grid_for(n)), dim3(kBlock), 0, nullptr,
                       make_euler_params(*params), make_sw_params(*params), make_aeos_params(*params), which, n,
                       din.ptr, dout.ptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpy(out, dout.ptr, n * n_out * sizeof(double), hipMemcpyDeviceToHost));
    return RYUJIN_OK;
  });
}

/* The decision at the end of a device-resident RK step from the two accumulated flags
 * (0 = never raised, otherwise kStageCode - index of the first stage that raised it; see time_step) */
int ryujin_hip_debug_rk_outcome(int restart_accum, int tau_invalid_accum, int id_violation_strategy)
{
  const int t = tau_invalid_accum, r = restart_accum;
  const bool restart_wins = id_violation_strategy == RYUJIN_IDV_RAISE_EXCEPTION && r > t;
  if (t > 0 && !restart_wins)
    return RYUJIN_ERR_TAU;
  if (r > 0)
    return id_violation_strategy == RYUJIN_IDV_RAISE_EXCEPTION ? RYUJIN_RESTART : RYUJIN_WARN;
  return RYUJIN_OK;
}

int ryujin_hip_set_timers(ryujin_hip_ctx *ctx, int enable)
{
  ctx->timers_enabled = enable != 0;
  return RYUJIN_OK;
}

int ryujin_hip_get_timers(ryujin_hip_ctx *ctx, double ms[8])
{
  for (int k = 0; k < 8; ++k)
    ms[k] = ctx->sweep_ms[k];
  return RYUJIN_OK;
}

int ryujin_hip_synchronize(ryujin_hip_ctx *ctx)
{
  return guarded_ctx(ctx, [&]() {
    ctx->finish();
    return RYUJIN_OK;
  });
}

int ryujin_hip_event_record(ryujin_hip_ctx *ctx, int which)
{
  return guarded_ctx(ctx, [&]() {
    if (which < 0 || which > 1)
      throw HipError(RYUJIN_ERR_ARG, "which must be 0 or 1");
    HIP_CHECK(hipEventRecord(ctx->ev_user[which], ctx->stream));
    return RYUJIN_OK;
  });
}

int ryujin_hip_event_elapsed_ms(ryujin_hip_ctx *ctx, double *ms)
{
  return guarded_ctx(ctx, [&]() {
    HIP_CHECK(hipEventSynchronize(ctx->ev_user[1]));
    float f = 0.f;
    HIP_CHECK(hipEventElapsedTime(&f, ctx->ev_user[0], ctx->ev_user[1]));
    *ms = f;
    return RYUJIN_OK;
  });
}

} /* extern "C" */

//
// hyperbolic_module_hip.h -- ryujin::HyperbolicModule<Description, dim, Number> on an MI355X.
//
// Drop-in for source/hyperbolic_module.h of the reference: same class name, template parameters,
// constructor, members and semantics (source/hyperbolic_module.h:72-336,
// source/hyperbolic_module.template.h:28-86, 96-193, 234-1211), implemented on top of the C ABI of
// libryujin_hip.so (include/ryujin_hip.h). TimeLoop (source/time_loop.h:193-204), TimeIntegrator
// (source/time_integrator.h:270,461), ParabolicModule, VTUOutput (source/time_loop.template.h:63-70) and
// Quantities keep compiling and calling it unchanged:
//
//   HyperbolicModule(mpi_communicator, computing_timer, offline_data, hyperbolic_system, initial_values,
//                    subsection)                                         hyperbolic_module.template.h:28-49
//   prepare()                  OfflineData -> ryujin_hip_offline IN MEMORY (ryujin_hip_binding.h),
//                              Description -> RYUJIN_EQ_*, ParameterAcceptor values -> ryujin_hip_params,
//                              MPI communicator -> RCCL communicator, ryujin_hip_create()          :52-86
//   prepare_state_vector(state_vector, t)                                                          :96-193
//   step<stages>(old, stage_state_vectors, stage_weights, new, tau, tau_max) -> tau               :234-1211
//       throws ryujin::Restart after the collective OR exactly where the reference does         :1194-1207
//   cfl(x), cfl(), offline_data(), hyperbolic_system(), initial_precomputed(), alpha(), n_restarts(),
//   n_warnings(), id_violation_strategy_                                       hyperbolic_module.h:225-278
//
// How to switch (contrib/hyperbolic_module_hip.patch): source/hyperbolic_module.h includes this header instead
// of declaring the class when RYUJIN_WITH_HIP is defined, and source/hyperbolic_module.cc (the explicit
// instantiations of the CPU implementation) compiles to nothing. Link libryujin_hip.so.
//
// State vectors. The reference's StateVector is a tuple of HOST vectors owned by the caller
// (source/state_vector.h:47-51), and the reference's TimeIntegrator works on them between the calls: sadd() for
// the SSPRK convex combinations and state_vector.swap(temp_[k]) at the end of every scheme
// (time_integrator.template.h:18-25,279-510). Every host state vector gets a device-resident twin, keyed by the
// DATA POINTER of U (swap() exchanges the storage, not the objects: the twin follows the storage).
//
//   Unmodified caller (the default; nothing but hyperbolic_module.h is patched). The HOST array is the authority
//   at every call, whatever the run-time parameters say:
//     prepare_state_vector(sv, t): upload sv.U, boundary conditions + ghost exchange + precomputation on the
//                                  device, write back ONLY what the call can have changed -- the boundary_map rows
//                                  and the ghost range of U -- and the precomputed values (the caller may read them:
//                                  VTU output, Quantities, compute_error all call prepare_state_vector first,
//                                  time_loop.template.h:374,701,858);
//     step(old, stages, w, new)  : old and the stage vectors are the twins prepared above (the contract of the
//                                  reference, hyperbolic_module.h:207-213); download new.U on the owned range and alpha.
//   One upload and one download of U plus one download of the precomputed values and alpha per update. With
//   `hip pin host vectors = true` the caller's arrays are pinned in place on first sight (ryujin_hip_host_register)
//   and the transfers are direct DMA; the default leaves them pageable (see the parameter's help text for why).
//
//   Patched caller (contrib/time_integrator_hip.patch: TimeIntegrator::step routes the explicit schemes to
//   time_step(), TimeLoop fetches the state before it reads it without asking the module). A whole Runge-Kutta step
//   of TimeIntegrator::step (time_integrator.template.h:207-403) runs inside the library with one host
//   synchronisation. With `hip device resident state vectors = true` time_step() leaves the result on the device
//   (the twin is marked "device ahead"); prepare_state_vector(), step() and synchronize_to_host() bring the host
//   array up to date before anything reads it, and upload it again only if the host copy is the newer one. The
//   parameter changes nothing for calls that do not come through time_step().
//
// Written against the reference snapshot; it needs deal.II and the ryujin headers. deal.II is not installed in the
// build image of ryujin_amd: there the header is TYPE CHECKED -- the reference's unmodified TimeIntegrator (all
// schemes) and VTUOutput are explicitly instantiated on top of it for all four Descriptions, against a patched
// copy of the reference tree and stand-in deal.II headers (tests/test_binding_compile.py, tests/cpp/dealii_mock/);
// it has not been linked or run against deal.II. Everything in it that does not need a deal.II type lives in
// ryujin_hip_binding.h and is compiled AND run there (tests/test_binding_cpp.py).
//
#pragma once

#include <compile_time_options.h>

#include "convenience_macros.h"
#include "initial_values.h"
#include "offline_data.h"
#include "scope.h"
#include "sparse_matrix_simd.h"
#include "state_vector.h"

#include <deal.II/base/mpi.h>
#include <deal.II/base/parameter_acceptor.h>
#include <deal.II/base/smartpointer.h>
#include <deal.II/base/timer.h>

#include <ryujin_hip.h>         /* ryujin_amd/include */
#include "ryujin_hip_binding.h" /* ryujin_amd/contrib */

#include <array>
#include <atomic>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <vector>

namespace ryujin
{
  /* forward declarations of the Descriptions the library implements (source/<equation>/description.h) */
  namespace Euler
  {
    struct Description;
  }
  namespace EulerAEOS
  {
    struct Description;
  }
  namespace ShallowWater
  {
    struct Description;
  }
  namespace ScalarConservation
  {
    struct Description;
  }

  /* source/hyperbolic_module.h:32-47 */
  enum class IDViolationStrategy {
    warn,
    raise_exception,
  };

  /* source/hyperbolic_module.h:49-57 */
  class Restart final
  {
  };


  namespace hip_detail
  {
    template <typename Description>
    struct EquationOf; /* a Description libryujin_hip.so does not implement fails to compile here */
    template <>
    struct EquationOf<Euler::Description> {
      static constexpr int value = RYUJIN_EQ_EULER;
    };
    template <>
    struct EquationOf<ShallowWater::Description> {
      static constexpr int value = RYUJIN_EQ_SHALLOW_WATER;
    };
    template <>
    struct EquationOf<EulerAEOS::Description> {
      static constexpr int value = RYUJIN_EQ_EULER_AEOS;
    };
    template <>
    struct EquationOf<ScalarConservation::Description> {
      static constexpr int value = RYUJIN_EQ_SCALAR_CONSERVATION;
    };
  } // namespace hip_detail


  template <typename Description, int dim, typename Number = double>
  class HyperbolicModule final : public dealii::ParameterAcceptor
  {
    static_assert(std::is_same<Number, double>::value, "libryujin_hip.so computes in double (NUMBER=double)");

  public:
    /* ---- typedefs and constants of the reference (hyperbolic_module.h:80-104) ---- */

    using HyperbolicSystem = typename Description::HyperbolicSystem;
    using View = typename Description::template HyperbolicSystemView<dim, Number>;
    static constexpr auto problem_dimension = View::problem_dimension;
    using state_type = typename View::state_type;
    using precomputed_type = typename View::precomputed_type;
    using initial_precomputed_type = typename View::initial_precomputed_type;
    using StateVector = typename View::StateVector;
    using InitialPrecomputedVector = typename View::InitialPrecomputedVector;
    static constexpr auto n_precomputation_cycles = View::n_precomputation_cycles;

    /** what is copied back to the host vectors after every call (see the header comment) */
    enum class Mirroring { full, device_resident };

    HyperbolicModule(const MPI_Comm &mpi_communicator,
                     std::map<std::string, dealii::Timer> &computing_timer,
                     const OfflineData<dim, Number> &offline_data,
                     const HyperbolicSystem &hyperbolic_system,
                     const InitialValues<Description, dim, Number> &initial_values,
                     const std::string &subsection = "/HyperbolicModule")
        : ParameterAcceptor(subsection)
        , id_violation_strategy_(IDViolationStrategy::warn)
        , indicator_parameters_(subsection + "/indicator")
        , limiter_parameters_(subsection + "/limiter")
        , riemann_solver_parameters_(subsection + "/riemann solver")
        , mpi_communicator_(mpi_communicator)
        , computing_timer_(computing_timer)
        , offline_data_(&offline_data)
        , hyperbolic_system_(&hyperbolic_system)
        , initial_values_(&initial_values)
        , cfl_(0.2)
        , n_restarts_(0)
        , n_warnings_(0)
    {
      hip_device_ = -1;
      add_parameter("hip device", hip_device_,
                    "HIP device ordinal of this MPI rank; -1: rank modulo the number of visible devices");
      hip_system_scope_events_ = false;
      add_parameter("hip system scope events", hip_system_scope_events_,
                    "Create the events that order the compute and the exchange stream with the system-scope "
                    "fence (ryujin_hip_params::system_scope_events)");
      hip_device_resident_ = false;
      add_parameter("hip device resident state vectors", hip_device_resident_,
                    "time_step() only (TimeIntegrator patched with contrib/time_integrator_hip.patch): leave the "
                    "result of a Runge-Kutta step on the device; the host vectors are brought up to date by "
                    "prepare_state_vector(), step() or synchronize_to_host(). Calls of an unmodified TimeIntegrator "
                    "(prepare_state_vector + step) always mirror the host vectors, whatever is set here");
      hip_pin_host_vectors_ = false;
      add_parameter("hip pin host vectors", hip_pin_host_vectors_,
                    "Pin the caller's state vectors in place (hipHostRegister) so that uploads and downloads are "
                    "direct DMA transfers instead of staged copies. Only if every state vector that reaches the "
                    "module lives as long as the module: true for TimeLoop's state vector and TimeIntegrator's "
                    "temporaries; the temporary 'analytic' vector of 'enable compute error' "
                    "(time_loop.template.h:320-333) has to be handed to release_state_vector() before it dies");
      hip_mirror_precomputed_ = true;
      add_parameter("hip mirror precomputed values", hip_mirror_precomputed_,
                    "prepare_state_vector() copies the precomputed values and step() copies alpha back to the host "
                    "vectors (the reference leaves them there for VTU output and Quantities; nothing in "
                    "TimeIntegrator reads them). If false: fetch alpha with synchronize_alpha_to_host()");
    }

    HyperbolicModule(const HyperbolicModule &) = delete;

    ~HyperbolicModule()
    {
      twins_.clear();
      if (ctx_)
        ryujin_hip_destroy(ctx_);
      if (comm_)
        ryujin_hip_comm_destroy(comm_);
    }

    /* ---- prepare(): hyperbolic_module.template.h:52-86 ---- */

    void prepare()
    {
      AssertThrow(limiter_parameters_.iterations() <= 2,
                  dealii::ExcMessage("The number of limiter iterations must be between [0,2]"));

      /* host-side vectors the reference exposes by reference (VTUOutput keeps them, time_loop.template.h:68-69) */
      alpha_.reinit(offline_data_->scalar_partitioner());
      initial_precomputed_ = initial_values_->interpolate_initial_precomputed_vector();

      /* initial_precomputed (shallow water: the bathymetry) as a flat array in local numbering */
      constexpr unsigned int n_ip = View::n_initial_precomputed_values;
      const unsigned int n_relevant = offline_data_->n_locally_relevant();
      std::vector<double> ip(std::size_t(n_relevant) * n_ip);
      if constexpr (n_ip != 0) {
        for (unsigned int i = 0; i < n_relevant; ++i) {
          const auto values = initial_precomputed_.get_tensor(i);
          for (unsigned int d = 0; d < n_ip; ++d)
            ip[std::size_t(i) * n_ip + d] = values[d];
        }
      }

      /* OfflineData -> ryujin_hip_offline, in memory */
      ryujin_hip_binding::fill_from_accessors<dim>(*offline_data_, arrays_, n_ip != 0 ? ip.data() : nullptr, n_ip);

      /* Description and ParameterAcceptor values -> ryujin_hip_params */
      ryujin_hip_default_params(&params_, hip_detail::EquationOf<Description>::value, dim);
      fill_description_params();
      params_.cfl = cfl_;
      params_.id_violation_strategy =
          id_violation_strategy_ == IDViolationStrategy::raise_exception ? RYUJIN_IDV_RAISE_EXCEPTION : RYUJIN_IDV_WARN;
      params_.system_scope_events = hip_system_scope_events_ ? 1 : 0;
      mirroring_ = hip_device_resident_ ? Mirroring::device_resident : Mirroring::full;

      /* one MPI rank per GPU; the MPI communicator becomes an RCCL communicator */
      const int rank = dealii::Utilities::MPI::this_mpi_process(mpi_communicator_);
      const int n_ranks = dealii::Utilities::MPI::n_mpi_processes(mpi_communicator_);
      int device = hip_device_;
      if (device < 0) {
        int n_devices = 1;
        check(ryujin_hip_device_count(&n_devices));
        device = rank % std::max(1, n_devices);
      }
      twins_.clear();
      if (ctx_) {
        ryujin_hip_destroy(ctx_);
        ctx_ = nullptr;
      }
      if (n_ranks > 1 && !comm_) {
        char id[RYUJIN_HIP_UNIQUE_ID_BYTES] = {0};
        if (rank == 0)
          check(ryujin_hip_comm_unique_id(id));
        MPI_Bcast(id, RYUJIN_HIP_UNIQUE_ID_BYTES, MPI_BYTE, 0, mpi_communicator_);
        check(ryujin_hip_comm_init(&comm_, id, rank, n_ranks, device));
      }
      check(ryujin_hip_create(&ctx_, &arrays_.offline, &params_, comm_, device));
      twins_.reset(ctx_);

      /* boundary_map entries whose boundary condition reads Dirichlet data (hyperbolic_system.h:1099-1159) */
      dirichlet_entries_.clear();
      dirichlet_positions_.clear();
      unsigned int e = 0;
      for (const auto &entry : offline_data_->boundary_map()) {
        const auto &[i, normal, normal_mass, boundary_mass, id, position] = entry;
        (void)normal;
        (void)normal_mass;
        (void)boundary_mass;
        if (i >= offline_data_->n_locally_owned())
          continue;
        if (id == Boundary::dirichlet || id == Boundary::dynamic || id == Boundary::dirichlet_momentum) {
          dirichlet_entries_.push_back(e);
          dirichlet_positions_.push_back(position);
        }
        ++e;
      }
      dirichlet_values_.assign(std::size_t(arrays_.offline.n_bdry) * problem_dimension, 0.);
    }

    /* ---- prepare_state_vector(): hyperbolic_module.template.h:96-193 ---- */

    void prepare_state_vector(StateVector &state_vector, Number t) const
    {
      Scope scope(computing_timer_, "time step [H] 1 - update boundary values, precompute values");
      auto &U = std::get<0>(state_vector);
      auto &twin = twin_of(state_vector);
      /* the host array is the authority (the caller may have filled, sadd()ed or swapped it), unless the last
       * writer was time_step() in device-resident mode */
      const bool device_ahead = twin.device_ahead;
      if (!device_ahead)
        check(ryujin_hip_state_upload(ctx_, twin.handle, U.begin()));
      evaluate_dirichlet(t, dirichlet_values_.data());
      check(ryujin_hip_prepare_state_vector(ctx_, twin.handle, t,
                                            dirichlet_entries_.empty() ? nullptr : dirichlet_values_.data()));
      if (device_ahead) {
        check(ryujin_hip_state_download(ctx_, twin.handle, U.begin()));
        check(ryujin_hip_get_alpha(ctx_, alpha_.begin()));
      } else {
        /* boundary_map rows and the ghost range: all this call can have changed in what was just uploaded */
        check(ryujin_hip_state_download_prepared(ctx_, twin.handle, U.begin()));
      }
      if (device_ahead || hip_mirror_precomputed_)
        check(ryujin_hip_state_download_precomputed(ctx_, twin.handle, std::get<1>(state_vector).begin()));
      twin.device_ahead = false;
    }

    /* ---- step<stages>(): hyperbolic_module.template.h:234-1211 ---- */

    template <int stages>
    Number step(const StateVector &old_state_vector,
                std::array<std::reference_wrapper<const StateVector>, stages> stage_state_vectors,
                const std::array<Number, stages> stage_weights,
                StateVector &new_state_vector,
                Number tau = Number(0.),
                std::atomic<Number> tau_max = std::numeric_limits<Number>::max()) const
    {
      /* sweeps 2-7 run back to back on the device; the reference's per-sweep Scope timers collapse into one
       * (ryujin_hip_get_timers() has the per-sweep device times) */
      Scope scope(computing_timer_, "time step [H] 2-7 - device");
      std::array<int, (stages > 0 ? stages : 1)> handles{};
      for (int s = 0; s < stages; ++s)
        handles[s] = prepared_twin_of(stage_state_vectors[s].get());
      const int h_old = prepared_twin_of(old_state_vector);
      auto &twin_new = twin_of(new_state_vector);
      const int h_new = twin_new.handle;

      check(ryujin_hip_set_cfl(ctx_, cfl_));
      check(ryujin_hip_set_id_violation_strategy(
          ctx_, id_violation_strategy_ == IDViolationStrategy::raise_exception ? RYUJIN_IDV_RAISE_EXCEPTION
                                                                               : RYUJIN_IDV_WARN));
      double tau_out = 0.;
      const int status = ryujin_hip_step(ctx_, h_old, stages, handles.data(), stage_weights.data(), h_new, tau,
                                         tau_max.load(), &tau_out);
      AssertThrow(status != RYUJIN_ERR_TAU,
                  dealii::ExcMessage("I'm sorry, Dave. I'm afraid I can't do that.\nWe crashed.")); /* :573-576 */
      check(status);
      update_counters();
      /* step() writes new.U on the owned range only (hyperbolic_module.h:207-213) */
      check(ryujin_hip_state_download_owned(ctx_, h_new, std::get<0>(new_state_vector).begin()));
      twin_new.device_ahead = false;
      if (hip_mirror_precomputed_)
        check(ryujin_hip_get_alpha(ctx_, alpha_.begin()));
      if (status == RYUJIN_RESTART)
        throw Restart(); /* all ranks together: the library reduced the flag over the ranks (:1194-1207) */
      return tau_out;
    }

    /* ---- beyond the reference ---- */

    /** TimeIntegrator::step(state_vector, t, t_final) (time_integrator.template.h:207-403) for the explicit
     * schemes, inside the library: one host synchronisation per Runge-Kutta step, tau and the restart flags
     * stay on the device, Dirichlet data is evaluated at the stage times t + c_s tau
     * (time_integrator.template.h:373-403). `scheme`: RYUJIN_SCHEME_*; `temp`: the integrator's temp_ vectors. */
    template <typename TempVectors>
    Number time_step(int scheme, StateVector &state_vector, TempVectors &temp, Number t,
                     Number t_final = std::numeric_limits<Number>::max(),
                     int cfl_recovery = RYUJIN_CFL_RECOVERY_NONE, Number cfl_min = 0.45, Number cfl_max = 0.9) const
    {
      Scope scope(computing_timer_, "time step [H] 1-7 - device, whole Runge-Kutta step");
      auto &twin = twin_of(state_vector);
      if (!twin.device_ahead)
        check(ryujin_hip_state_upload(ctx_, twin.handle, std::get<0>(state_vector).begin()));
      const int h = twin.handle;
      std::vector<int> h_tmp(temp.size());
      for (std::size_t q = 0; q < temp.size(); ++q) {
        auto &scratch = twin_of(temp[q]); /* scratch on both sides: the host arrays of temp_ are never read */
        scratch.device_ahead = false;
        h_tmp[q] = scratch.handle;
      }
      check(ryujin_hip_set_cfl(ctx_, cfl_));
      double tau_out = 0.;
      const int status = ryujin_hip_time_step_fn(ctx_, scheme, h, (int)h_tmp.size(), h_tmp.data(), t,
                                                 dirichlet_entries_.empty() ? nullptr : &dirichlet_callback,
                                                 const_cast<HyperbolicModule *>(this), t_final - t, cfl_recovery,
                                                 cfl_min, cfl_max, &tau_out);
      AssertThrow(status != RYUJIN_ERR_TAU,
                  dealii::ExcMessage("I'm sorry, Dave. I'm afraid I can't do that.\nWe crashed."));
      check(status);
      update_counters();
      check(ryujin_hip_get_cfl(ctx_, &cfl_));
      if (mirroring_ == Mirroring::full) {
        /* as step(): the owned range of U (the reference's schemes leave ghosts and precomputed values of the
         * result unprepared as well) */
        check(ryujin_hip_state_download_owned(ctx_, h, std::get<0>(state_vector).begin()));
        check(ryujin_hip_get_alpha(ctx_, alpha_.begin()));
        twin.device_ahead = false;
      } else {
        twin.device_ahead = true;
      }
      if (status == RYUJIN_RESTART)
        throw Restart();
      return tau_out;
    }

    /** bring the host vectors up to date with the device twin if time_step() left the twin ahead (U on the owned
     * and the ghost range, the precomputed values, alpha); a no-op otherwise */
    void synchronize_to_host(StateVector &state_vector) const
    {
      auto &twin = twin_of(state_vector);
      if (!twin.device_ahead)
        return;
      check(ryujin_hip_state_download(ctx_, twin.handle, std::get<0>(state_vector).begin()));
      check(ryujin_hip_state_download_precomputed(ctx_, twin.handle, std::get<1>(state_vector).begin()));
      check(ryujin_hip_get_alpha(ctx_, alpha_.begin()));
      twin.device_ahead = false;
    }

    /** alpha of the last step (full mirroring keeps alpha() current by itself) */
    void synchronize_alpha_to_host() const { check(ryujin_hip_get_alpha(ctx_, alpha_.begin())); }

    void mirroring(Mirroring m) const { mirroring_ = m; }

    ryujin_hip_ctx *context() const { return ctx_; }

    /* ---- accessors of the reference: hyperbolic_module.h:225-278 ---- */

    void cfl(Number new_cfl) const
    {
      Assert(cfl_ > Number(0.), dealii::ExcInternalError());
      cfl_ = new_cfl;
    }

    ACCESSOR_READ_ONLY(cfl)
    ACCESSOR_READ_ONLY(offline_data)
    ACCESSOR_READ_ONLY(hyperbolic_system)
    ACCESSOR_READ_ONLY(initial_precomputed)
    ACCESSOR_READ_ONLY(alpha)
    ACCESSOR_READ_ONLY(n_restarts)
    ACCESSOR_READ_ONLY(n_warnings)

    mutable IDViolationStrategy id_violation_strategy_;

  private:
    static void check(int status)
    {
      AssertThrow(status >= 0, dealii::ExcMessage(std::string("ryujin_hip: ") + ryujin_hip_last_error()));
    }

    void update_counters() const
    {
      unsigned r = 0, w = 0;
      check(ryujin_hip_get_counters(ctx_, &r, &w));
      n_restarts_ = r;
      n_warnings_ = w;
    }

    /* the twin of the STORAGE of a host state vector (swap() moves the storage between the tuples); the caller's
     * arrays are pinned in place when the twin is created */
    ryujin_hip_binding::HandleCache::Twin &twin_of(const StateVector &state_vector) const
    {
      const auto &U = std::get<0>(state_vector);
      bool created = false;
      auto &twin = twins_.twin(U.begin(), &created);
      if (created && hip_pin_host_vectors_) {
        const auto &precomputed = std::get<1>(state_vector);
        const std::size_t n_relevant = arrays_.offline.n_relevant;
        const int rc_U = ryujin_hip_host_register(ctx_, U.begin(), n_relevant * problem_dimension * sizeof(Number));
        const int rc_p = ryujin_hip_host_register(ctx_, precomputed.begin(),
                                                  n_relevant * View::n_precomputed_values * sizeof(Number));
        check(rc_U);
        check(rc_p);
        twin.pinned = rc_U == RYUJIN_OK && rc_p == RYUJIN_OK;
      }
      return twin;
    }

  public:
    /**
     * A state vector that has been through prepare_state_vector() is about to be destroyed (the temporary
     * 'analytic' vector of TimeLoop::compute_error, time_loop.template.h:320-333): frees its device twin and, with
     * `hip pin host vectors = true`, ends the pinning of its arrays. Twins are keyed by the address of the storage; a
     * vector that dies without this call leaves a twin -- and a registration -- behind that the next allocation at
     * the same address would inherit. No effect on a vector the module has not seen.
     */
    void release_state_vector(const StateVector &state_vector) const
    {
      const auto &U = std::get<0>(state_vector);
      if (twins_.forget(U.begin())) {
        ryujin_hip_host_unregister(ctx_, U.begin());
        ryujin_hip_host_unregister(ctx_, std::get<1>(state_vector).begin());
      }
    }

  private:
    /* old and stage vectors of step(): "must be prepared" (hyperbolic_module.h:207-213), i.e. a twin exists */
    int prepared_twin_of(const StateVector &state_vector) const
    {
      bool created = false;
      const int h = twins_.twin(std::get<0>(state_vector).begin(), &created).handle;
      AssertThrow(!created, dealii::ExcMessage("HyperbolicModule::step(): the old state vector and the stage state "
                                               "vectors have to be prepared with prepare_state_vector() first"));
      return h;
    }

    /* initial_values_->initial_state(position, t) for the boundary_map entries that read it (:137-139) */
    void evaluate_dirichlet(Number t, double *values) const
    {
      for (std::size_t q = 0; q < dirichlet_entries_.size(); ++q) {
        const auto state = initial_values_->initial_state(dirichlet_positions_[q], t);
        for (unsigned int d = 0; d < problem_dimension; ++d)
          values[std::size_t(dirichlet_entries_[q]) * problem_dimension + d] = state[d];
      }
    }

    static void dirichlet_callback(void *user, double t, double *values)
    {
      static_cast<const HyperbolicModule *>(user)->evaluate_dirichlet(t, values);
    }

    /* "B - Equation" and "/HyperbolicModule/{indicator,limiter,riemann solver}" of the Description */
    void fill_description_params()
    {
      const auto view = hyperbolic_system_->template view<dim, Number>();
      if constexpr (std::is_same<Description, Euler::Description>::value) {
        ryujin_hip_binding::fill_params_euler(params_, view);
        ryujin_hip_binding::fill_params_common(params_, indicator_parameters_, limiter_parameters_,
                                               riemann_solver_parameters_);
      } else if constexpr (std::is_same<Description, ShallowWater::Description>::value) {
        ryujin_hip_binding::fill_params_shallow_water(params_, view, indicator_parameters_, limiter_parameters_);
      } else if constexpr (std::is_same<Description, EulerAEOS::Description>::value) {
        ryujin_hip_binding::fill_params_indicator_limiter(params_, indicator_parameters_, limiter_parameters_);
        params_.reference_density = view.reference_density();
        params_.vacuum_state_relaxation_small = view.vacuum_state_relaxation_small();
        params_.vacuum_state_relaxation_large = view.vacuum_state_relaxation_large();
        params_.compute_strict_bounds = view.compute_strict_bounds() ? 1 : 0;
        /* the equation of state is selected by name in "B - Equation/equation of state"; its parameters live
         * in the EquationOfState object (euler_aeos/equation_of_state_library.h): set params_.eos and the
         * eos_* / jwl_* fields from it before prepare() through eos_parameters() */
        if (eos_parameters_)
          eos_parameters_(params_);
      } else {
        /* scalar conservation: the flux is selected by name ("B - Equation/flux"); as for the EOS */
        params_.indicator_evc_factor = indicator_parameters_.evc_factor();
        params_.limiter_iterations = static_cast<int>(limiter_parameters_.iterations());
        params_.limiter_relaxation_factor = limiter_parameters_.relaxation_factor();
#ifdef EXPENSIVE_BOUNDS_CHECK
        params_.debug_expensive_bounds_check = 1;
#endif
        if (eos_parameters_)
          eos_parameters_(params_);
      }
    }

  public:
    /** EulerAEOS / scalar conservation: a hook that copies the run-time selected equation of state (flux) and
     * its parameters into ryujin_hip_params (fields eos, eos_*, jwl_* / sc_*), called by prepare() */
    void eos_parameters(std::function<void(ryujin_hip_params &)> f) { eos_parameters_ = std::move(f); }

  private:
    /* run-time options of the reference (hyperbolic_module.h:283-291) */
    typename Description::template Indicator<dim, Number>::Parameters indicator_parameters_;
    typename Description::template Limiter<dim, Number>::Parameters limiter_parameters_;
    typename Description::template RiemannSolver<dim, Number>::Parameters riemann_solver_parameters_;

    int hip_device_;
    bool hip_system_scope_events_;
    bool hip_device_resident_;
    bool hip_pin_host_vectors_;
    bool hip_mirror_precomputed_;

    const MPI_Comm &mpi_communicator_;
    std::map<std::string, dealii::Timer> &computing_timer_;

    dealii::SmartPointer<const OfflineData<dim, Number>> offline_data_;
    dealii::SmartPointer<const HyperbolicSystem> hyperbolic_system_;
    dealii::SmartPointer<const InitialValues<Description, dim, Number>> initial_values_;

    mutable Number cfl_;
    mutable unsigned int n_restarts_;
    mutable unsigned int n_warnings_;

    InitialPrecomputedVector initial_precomputed_;
    using ScalarVector = typename Vectors::ScalarVector<Number>;
    mutable ScalarVector alpha_;

    /* the device side */
    ryujin_hip_binding::OfflineArrays arrays_;
    ryujin_hip_params params_{};
    ryujin_hip_comm *comm_ = nullptr;
    ryujin_hip_ctx *ctx_ = nullptr;
    mutable ryujin_hip_binding::HandleCache twins_;
    mutable Mirroring mirroring_ = Mirroring::full;
    std::function<void(ryujin_hip_params &)> eos_parameters_;

    std::vector<unsigned int> dirichlet_entries_; /* index into the boundary_map (= ryujin_hip_offline::b_i order) */
    std::vector<dealii::Point<dim>> dirichlet_positions_;
    mutable std::vector<double> dirichlet_values_;
  };

} /* namespace ryujin */

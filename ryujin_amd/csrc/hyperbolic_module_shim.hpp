// hyperbolic_module_shim.hpp -- C++ host shim with the member signatures of
// ryujin::HyperbolicModule<Description, dim, Number> (source/hyperbolic_module.h:72-336) and the
// TimeIntegrator helper sadd(), forwarding to the C ABI of include/ryujin_hip.h.
//
// This is the binding a ryujin maintainer would add: TimeIntegrator::step_* and TimeLoop keep calling
//   hyperbolic_module_->prepare_state_vector(sv, t);
//   tau = hyperbolic_module_->template step<stages>(old, {stage vectors}, {weights}, new, tau, tau_max);
// unchanged; StateVector becomes a handle to device-resident storage (see INTEGRATION.md for the
// host<->device copies at the I/O boundaries of time_loop).
//
// Header-only, no HIP or deal.II types; link with -lryujin_hip.

#pragma once

#include <algorithm>
#include <array>
#include <functional>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

#include "ryujin_hip.h"

namespace ryujin_hip_shim
{
  /* ryujin::Restart (source/hyperbolic_module.h:49-57) */
  class Restart final
  {
  };

  /* ryujin::IDViolationStrategy (source/hyperbolic_module.h:32-47) */
  enum class IDViolationStrategy : int {
    warn = RYUJIN_IDV_WARN,
    raise_exception = RYUJIN_IDV_RAISE_EXCEPTION
  };

  class HyperbolicModule;

  /* StateVector = (U, precomputed, V) of source/state_vector.h:47-51, device resident */
  class StateVector
  {
  public:
    StateVector() = default;
    StateVector(const StateVector &) = delete;
    StateVector &operator=(const StateVector &) = delete;
    StateVector(StateVector &&o) noexcept { swap(o); }
    StateVector &operator=(StateVector &&o) noexcept
    {
      swap(o);
      return *this;
    }
    ~StateVector()
    {
      if (ctx_ && handle_ >= 0)
        ryujin_hip_state_free(ctx_, handle_);
    }
    void swap(StateVector &o) noexcept
    {
      std::swap(ctx_, o.ctx_);
      std::swap(handle_, o.handle_);
    }
    int handle() const { return handle_; }

  private:
    friend class HyperbolicModule;
    ryujin_hip_ctx *ctx_ = nullptr;
    int handle_ = -1;
  };


  class HyperbolicModule
  {
  public:
    /* The reference constructor takes (mpi_communicator, computing_timer, offline_data,
     * hyperbolic_system, initial_values, subsection); here the flattened equivalents. */
    HyperbolicModule(const ryujin_hip_offline &offline, const ryujin_hip_params &params,
                     ryujin_hip_comm *comm = nullptr, int device = 0)
        : offline_(offline)
        , params_(params)
        , comm_(comm)
        , device_(device)
    {
    }

    HyperbolicModule(const HyperbolicModule &) = delete;
    ~HyperbolicModule()
    {
      if (ctx_)
        ryujin_hip_destroy(ctx_);
    }

    /* prepare(): hyperbolic_module.template.h:52-86 -- call after OfflineData::prepare() */
    void prepare()
    {
      if (ctx_) {
        ryujin_hip_destroy(ctx_);
        ctx_ = nullptr;
      }
      check(ryujin_hip_create(&ctx_, &offline_, &params_, comm_, device_));
      /* problem_dimension of the Description: shallow water dim+1, scalar conservation 1, Euler(AEOS) dim+2 */
      k_ = params_.equation == RYUJIN_EQ_SHALLOW_WATER
               ? params_.dim + 1
               : (params_.equation == RYUJIN_EQ_SCALAR_CONSERVATION ? 1 : params_.dim + 2);
    }

    StateVector create_state_vector() const
    {
      StateVector sv;
      sv.ctx_ = ctx_;
      check(ryujin_hip_state_alloc(ctx_, &sv.handle_));
      return sv;
    }
    /* MultiComponentVector layout U[i*k+d], i < n_locally_relevant */
    void upload(StateVector &sv, const double *U_aos) const
    {
      check(ryujin_hip_state_upload(ctx_, sv.handle_, U_aos));
    }
    void download(const StateVector &sv, double *U_aos) const
    {
      check(ryujin_hip_state_download(ctx_, sv.handle_, U_aos));
    }

    /* Dirichlet data = initial_values_->initial_state(position, t) per boundary_map entry
     * (hyperbolic_module.template.h:137-139); evaluated by the host shim. */
    using dirichlet_function = std::function<void(double t, std::vector<double> &values /*[n_bdry*k]*/)>;
    void set_dirichlet_function(dirichlet_function f) { dirichlet_ = std::move(f); }

    /* prepare_state_vector(state_vector, t): hyperbolic_module.template.h:96-193 */
    void prepare_state_vector(StateVector &state_vector, double t) const
    {
      const double *ptr = nullptr;
      if (dirichlet_) {
        dirichlet_values_.resize((size_t)offline_.n_bdry * k_);
        dirichlet_(t, dirichlet_values_);
        ptr = dirichlet_values_.data();
      }
      check(ryujin_hip_prepare_state_vector(ctx_, state_vector.handle_, t, ptr));
    }

    /* step<stages>(...): hyperbolic_module.template.h:234-1211. Throws Restart exactly where the
     * reference does (after the collective OR, :1194-1207). */
    template <int stages>
    double step(const StateVector &old_state_vector,
                std::array<std::reference_wrapper<const StateVector>, stages> stage_state_vectors,
                const std::array<double, stages> stage_weights, StateVector &new_state_vector,
                double tau = 0., double tau_max = std::numeric_limits<double>::max()) const
    {
      std::array<int, (stages > 0 ? stages : 1)> handles{};
      for (int s = 0; s < stages; ++s)
        handles[s] = stage_state_vectors[s].get().handle_;
      double tau_out = 0.;
      const int status =
          ryujin_hip_step(ctx_, old_state_vector.handle_, stages, handles.data(), stage_weights.data(),
                          new_state_vector.handle_, tau, tau_max, &tau_out);
      if (status == RYUJIN_ERR_TAU)
        throw std::runtime_error("I'm sorry, Dave. I'm afraid I can't do that.\nWe crashed.");
      check(status);
      if (status == RYUJIN_RESTART)
        throw Restart();
      return tau_out;
    }

    /* TimeIntegrator::step(state_vector, t, t_final) (time_integrator.template.h:207-277) for the
     * explicit schemes, executed device-resident with one host synchronisation per Runge-Kutta step
     * (ryujin_hip_time_step). `scheme` is a RYUJIN_SCHEME_* id, `temp` the integrator's temp_[0..2].
     * With cfl_recovery == RYUJIN_CFL_RECOVERY_BANG_BANG the reference's retry loop (:250-274) runs
     * inside the library; otherwise a Restart propagates as in step(). Dirichlet data is evaluated by the
     * dirichlet function at every stage time. */
    template <std::size_t n_temp>
    double time_step(int scheme, StateVector &state_vector, std::array<StateVector, n_temp> &temp, double t,
                     double t_final = std::numeric_limits<double>::max(),
                     int cfl_recovery = RYUJIN_CFL_RECOVERY_NONE, double cfl_min = 0.45,
                     double cfl_max = 0.9) const
    {
      int h_tmp[n_temp]; /* temp_[0..n): 3 for SSPRK22/33 and ERK11/22/33, 4 for ERK43, 5 for ERK54 */
      for (std::size_t q = 0; q < n_temp; ++q)
        h_tmp[q] = temp[q].handle_;
      double tau_out = 0.;
      /* Dirichlet data at the stage times t + c_s tau (time_integrator.template.h:373-403): the library calls
       * back once per stage, for the later stages as soon as tau exists on the host */
      const int status = ryujin_hip_time_step_fn(ctx_, scheme, state_vector.handle_, (int)n_temp, h_tmp, t,
                                                 dirichlet_ ? &dirichlet_trampoline : nullptr,
                                                 const_cast<HyperbolicModule *>(this), t_final - t, cfl_recovery,
                                                 cfl_min, cfl_max, &tau_out);
      if (status == RYUJIN_ERR_TAU)
        throw std::runtime_error("I'm sorry, Dave. I'm afraid I can't do that.\nWe crashed.");
      check(status);
      if (status == RYUJIN_RESTART)
        throw Restart();
      return tau_out;
    }

    /* accessors: hyperbolic_module.h:225-278 */
    void cfl(double new_cfl) const { check(ryujin_hip_set_cfl(ctx_, new_cfl)); }
    double cfl() const
    {
      double v = 0.;
      check(ryujin_hip_get_cfl(ctx_, &v));
      return v;
    }
    std::vector<double> alpha() const
    {
      std::vector<double> a(offline_.n_relevant);
      check(ryujin_hip_get_alpha(ctx_, a.data()));
      return a;
    }
    unsigned int n_restarts() const { return counters().first; }
    unsigned int n_warnings() const { return counters().second; }

    /* public member of the reference (hyperbolic_module.h:276) */
    void id_violation_strategy(IDViolationStrategy s) const
    {
      check(ryujin_hip_set_id_violation_strategy(ctx_, static_cast<int>(s)));
    }

    ryujin_hip_ctx *context() const { return ctx_; }

  private:
    static void dirichlet_trampoline(void *user, double time, double *values)
    {
      const auto *self = static_cast<const HyperbolicModule *>(user);
      self->dirichlet_values_.resize((size_t)self->offline_.n_bdry * self->k_);
      self->dirichlet_(time, self->dirichlet_values_);
      std::copy(self->dirichlet_values_.begin(), self->dirichlet_values_.end(), values);
    }

    static void check(int status)
    {
      if (status < 0)
        throw std::runtime_error(std::string("ryujin_hip: ") + ryujin_hip_last_error());
    }
    std::pair<unsigned, unsigned> counters() const
    {
      unsigned r = 0, w = 0;
      check(ryujin_hip_get_counters(ctx_, &r, &w));
      return {r, w};
    }

    ryujin_hip_offline offline_;
    ryujin_hip_params params_;
    ryujin_hip_comm *comm_;
    int device_;
    ryujin_hip_ctx *ctx_ = nullptr;
    int k_ = 0;
    dirichlet_function dirichlet_;
    mutable std::vector<double> dirichlet_values_;
  };

  /* sadd(dst, s, b, src): time_integrator.template.h:18-25 */
  inline void sadd(const HyperbolicModule &m, StateVector &dst, double s, double b, const StateVector &src)
  {
    if (ryujin_hip_sadd(m.context(), dst.handle(), s, b, src.handle()) < 0)
      throw std::runtime_error(std::string("ryujin_hip: ") + ryujin_hip_last_error());
  }
} // namespace ryujin_hip_shim

// Type-level check of the drop-in boundary (tests/test_binding_compile.py): the reference's
// UNMODIFIED TimeIntegrator (source/time_integrator.template.h:207-403 and the other schemes) and VTUOutput instantiated on top of
// contrib/hyperbolic_module_hip.h -- the class `source/hyperbolic_module.h` turns into under RYUJIN_WITH_HIP
// (contrib/hyperbolic_module_hip.patch) -- for the Descriptions the library implements, plus
// contrib/ryujin_export_offline.h. Compiled against a patched COPY of the reference tree (made by the test in a
// temporary directory) and the deal.II stand-in headers of tests/cpp/dealii_mock/. Nothing here is executed.
#include "time_integrator.template.h"
#include "vtu_output.template.h"

#include "ryujin_export_offline.h"

#include RYUJIN_DESCRIPTION_HEADER /* "euler/description.h", ... : one equation per translation unit, as the reference
                                      builds them (source/<equation>/CMakeLists.txt adds its directory to the path) */

namespace ryujin
{
  using Description = RYUJIN_DESCRIPTION;

  template class HyperbolicModule<Description, 1, double>;
  template class HyperbolicModule<Description, 2, double>;
  template class HyperbolicModule<Description, 3, double>;
  template class TimeIntegrator<Description, 1, double>;
  template class TimeIntegrator<Description, 2, double>;
  template class TimeIntegrator<Description, 3, double>;
  /* reads hyperbolic_module.initial_precomputed() and .alpha() (vtu_output.template.h:63-70,96-104) */
  template class VTUOutput<Description, 2, double>;

  /* the calls TimeLoop and VTUOutput make (time_loop.template.h:63-70,249,374,701,858,1220-1226), the exporter as
   * contrib/ryujin_export_offline.patch calls it, and the members the adapter adds */
  template <int dim>
  void time_loop_calls(HyperbolicModule<Description, dim, double> &m,
                       typename Description::template HyperbolicSystemView<dim, double>::StateVector &sv,
                       const OfflineData<dim, double> &offline_data, const MPI_Comm comm)
  {
    using View = typename Description::template HyperbolicSystemView<dim, double>;
    using StateVector = typename View::StateVector;
    m.prepare();
    m.prepare_state_vector(sv, 0.);
    const Vectors::ScalarVector<double> &alpha = m.alpha(); /* vtu_output.template.h reads it as such */
    const typename HyperbolicModule<Description, dim, double>::InitialPrecomputedVector &ip = m.initial_precomputed();
    (void)alpha;
    (void)ip;
    m.cfl(0.5);
    const double cfl = m.cfl();
    const unsigned int n_restarts = m.n_restarts(), n_warnings = m.n_warnings();
    (void)cfl;
    (void)n_restarts;
    (void)n_warnings;
    const OfflineData<dim, double> &od = m.offline_data();
    const typename Description::HyperbolicSystem &hs = m.hyperbolic_system();
    (void)od;
    (void)hs;
    m.id_violation_strategy_ = IDViolationStrategy::raise_exception;
    try {
      StateVector &new_sv = sv;
      const double tau = m.template step<0>(sv, {}, {}, new_sv, 0., 1.);
      (void)tau;
    } catch (const Restart &) {
    }
    std::array<StateVector, 3> temp;
    const double tau_rk = m.time_step(RYUJIN_SCHEME_ERK_33, sv, temp, 0., 1.);
    (void)tau_rk;
    m.synchronize_to_host(sv);
    constexpr unsigned int n_ip = View::n_initial_precomputed_values;
    std::vector<double> values(std::size_t(offline_data.n_locally_relevant()) * n_ip);
    export_offline_data(offline_data, "dump", comm, n_ip != 0 ? values.data() : nullptr, n_ip);
  }
  template void time_loop_calls<1>(HyperbolicModule<Description, 1, double> &,
                                   Description::HyperbolicSystemView<1, double>::StateVector &,
                                   const OfflineData<1, double> &, const MPI_Comm);
  template void time_loop_calls<2>(HyperbolicModule<Description, 2, double> &,
                                   Description::HyperbolicSystemView<2, double>::StateVector &,
                                   const OfflineData<2, double> &, const MPI_Comm);
  template void time_loop_calls<3>(HyperbolicModule<Description, 3, double> &,
                                   Description::HyperbolicSystemView<3, double>::StateVector &,
                                   const OfflineData<3, double> &, const MPI_Comm);
} // namespace ryujin

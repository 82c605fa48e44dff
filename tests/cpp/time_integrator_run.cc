// The drop-in boundary EXECUTED (tests/test_binding_run.py): the reference's own
// TimeIntegrator (source/time_integrator.template.h, its step(), step_ssprk_33(), step_erk_33() ..., sadd() and the
// StateVector swap), MultiComponentVector, SparsityPatternSIMD::reinit and SparseMatrixSIMD run here on top of
// contrib/hyperbolic_module_hip.h + libryujin_hip.so, on the configuration of
// tests/euler/check-mass-conservation_01.prm (65 x 65 slip box [0,20]^2, uniform Mach-3 state, cfl 0.9, no recovery).
//
// What is real and what is a stand-in:
//   real, from a patched temporary COPY of the reference tree (tests/helpers_reference_tree.py): time_integrator.h,
//     time_integrator.template.h (unmodified, or with contrib/time_integrator_hip.patch), parabolic_module.template.h,
//     state_vector.h, multicomponent_vector.h, sparse_matrix_simd.h + .template.h, offline_data.h (the class; filled
//     below), initial_values.h (the class), euler/description.h and everything it includes, scope.h;
//   real: contrib/hyperbolic_module_hip.h, contrib/ryujin_hip_binding.h, libryujin_hip.so, libryujin_synth.so;
//   stand-in: deal.II (tests/cpp/dealii_mock/, one MPI rank, the containers with just the behaviour the code above
//     uses), and the three collaborators whose implementations need deal.II's grid and FE stack:
//     OfflineData::setup()/assemble() (explicitly specialised below: they fill the reference's private members from
//     the synthetic generator instead of assembling them), the constructors of Discretization and InitialValues, and
//     InitialValues::interpolate_initial_precomputed_vector().
// Test scaffolding for the BOUNDARY: not an oracle, not a build of the reference's hot path (which is what
// libryujin_hip.so replaces).
//
//   time_integrator_run export <prefix> [box:<cells> | step:<cells per unit>]
//     contrib/ryujin_export_offline.h EXECUTED -- export_offline_data() walks the reference's
//     own SparsityPatternSIMD / SparseMatrixSIMD (SIMD-interleaved rows, get_entry / get_tensor) and the OfflineData
//     accessors and writes <prefix>-0.ryjoffl; tests/test_offline_export_run.py imports the dump and runs the kernels on
//     it against the oracle. (Mach-3 step: Dirichlet inflow, do-nothing outflow, slip walls, coupling boundary pairs.)
//   time_integrator_run <scheme> <n_steps> <device_resident 0|1> [n_cells]
//     environment: RYUJIN_TEST_PIN=1 -> "hip pin host vectors = true"; RYUJIN_TEST_NO_DERIVED=1 -> "hip mirror
//     precomputed values = false"
//     scheme: "ssprk33", "erk33", "erk11", "ssprk22", "erk22", "erk43", "erk54"
//   prints "t mean_rho" per step (%.14e), then "checksum <fnv1a of the final U on the host>" and the
//   adapter's transfer accounting.
#include "time_integrator.template.h"

#include "parabolic_module.template.h"
#include "sparse_matrix_simd.template.h"

#include "euler/description.h"

#include "ryujin_export_offline.h"

#include <ryujin_synth.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

namespace ryujin
{
  using Description = Euler::Description;
  constexpr int dim = 2;
  using Number = double;

  /* the mesh the stand-ins below serve (set by main before anything is constructed) */
  static const ryujin_hip_offline *g_offline = nullptr;
  static const double *g_positions = nullptr, *g_bdry_positions = nullptr;


  /* ---- stand-ins for the collaborators that need deal.II's grid/FE stack ------------------------------- */

  template <>
  Discretization<dim>::Discretization(const MPI_Comm &mpi_communicator, const std::string &subsection)
      : ParameterAcceptor(subsection)
      , mpi_communicator_(mpi_communicator)
  {
    ansatz_ = Ansatz::cg_q1;
    refinement_ = 6;
    mapping_ = std::make_unique<dealii::Mapping<dim>>(); /* (an identity the exporter hands on, see below) */
  }


  template <>
  OfflineData<dim, Number>::OfflineData(const MPI_Comm &mpi_communicator,
                                        const Discretization<dim> &discretization,
                                        const std::string &subsection)
      : ParameterAcceptor(subsection)
      , discretization_(&discretization)
      , mpi_communicator_(mpi_communicator)
  {
  }


  /* OfflineData::setup() (offline_data.template.h:120-420): partitioners, index ranges, sparsity pattern. The stencil
   * goes through the reference's own SparsityPatternSIMD::reinit(). */
  template <>
  void OfflineData<dim, Number>::setup(const unsigned int problem_dimension, const unsigned int n_precomputed_values)
  {
    const ryujin_hip_offline &o = *g_offline;
    AssertThrow(o.n_owned == o.n_relevant, dealii::ExcMessage("one rank"));
    dof_handler_ = std::make_unique<dealii::DoFHandler<dim>>();

    dealii::IndexSet owned(o.n_owned), ghost(o.n_owned);
    owned.add_range(0, o.n_owned);
    scalar_partitioner_ = std::make_shared<dealii::Utilities::MPI::Partitioner>(owned, ghost, mpi_communicator_);
    hyperbolic_vector_partitioner_ = Vectors::create_vector_partitioner(scalar_partitioner_, problem_dimension);
    precomputed_vector_partitioner_ = Vectors::create_vector_partitioner(scalar_partitioner_, n_precomputed_values);

    n_export_indices_ = o.n_export;
    n_locally_internal_ = o.n_internal;
    n_locally_owned_ = o.n_owned;
    n_locally_relevant_ = o.n_relevant;

    sparsity_pattern_.reinit(o.n_relevant, o.n_relevant);
    for (unsigned int i = 0; i < o.n_relevant; ++i)
      for (uint64_t e = o.row_starts[i]; e < o.row_starts[i + 1]; ++e)
        sparsity_pattern_.add(i, o.columns[e]);

    sparsity_pattern_simd_.reinit(n_locally_internal_, sparsity_pattern_, scalar_partitioner_);
  }


  /* OfflineData::assemble() (offline_data.template.h:430-1050): the matrices, the boundary map, the coupling pairs.
   * Written through the reference's own SparseMatrixSIMD::write_entry() / MultiComponentVector interfaces. */
  template <>
  void OfflineData<dim, Number>::assemble()
  {
    const ryujin_hip_offline &o = *g_offline;

    mass_matrix_.reinit(sparsity_pattern_simd_);
    cij_matrix_.reinit(sparsity_pattern_simd_);
    lumped_mass_matrix_.reinit(scalar_partitioner_);
    lumped_mass_matrix_inverse_.reinit(scalar_partitioner_);

    for (unsigned int i = 0; i < o.n_relevant; ++i) {
      /* the generator's rows are diagonal first and then in its own order; the reference's are diagonal first and
       * then ascending (dealii::SparsityPattern): look every entry up by its column */
      const unsigned int row_length = sparsity_pattern_simd_.row_length(i);
      const unsigned int *js = sparsity_pattern_simd_.columns(i);
      const unsigned int stride = sparsity_pattern_simd_.stride_of_row(i);
      AssertThrow(row_length == o.row_starts[i + 1] - o.row_starts[i], dealii::ExcInternalError());
      for (unsigned int col_idx = 0; col_idx < row_length; ++col_idx) {
        const unsigned int j = js[col_idx * stride];
        uint64_t e = o.row_starts[i];
        while (e < o.row_starts[i + 1] && o.columns[e] != j)
          ++e;
        AssertThrow(e < o.row_starts[i + 1], dealii::ExcInternalError());
        mass_matrix_.write_entry(o.mij[e], i, col_idx);
        dealii::Tensor<1, dim, Number> c_ij;
        for (int d = 0; d < dim; ++d)
          c_ij[d] = o.cij[e * dim + d];
        cij_matrix_.write_entry(c_ij, i, col_idx);
      }
      lumped_mass_matrix_.local_element(i) = o.mi[i];
      lumped_mass_matrix_inverse_.local_element(i) = o.mi_inv[i];
    }
    measure_of_omega_ = o.measure_of_omega;

    boundary_map_.clear();
    for (unsigned int q = 0; q < o.n_bdry; ++q) {
      dealii::Tensor<1, dim, Number> normal;
      dealii::Point<dim> position;
      for (int d = 0; d < dim; ++d) {
        normal[d] = o.b_normal[q * dim + d];
        position[d] = g_bdry_positions ? g_bdry_positions[q * dim + d] : 0.;
      }
      boundary_map_.push_back({o.b_i[q], normal, Number(0.), Number(0.), dealii::types::boundary_id(o.b_id[q]), position});
    }
    coupling_boundary_pairs_.clear();
    for (unsigned int q = 0; q < o.n_pairs; ++q)
      coupling_boundary_pairs_.push_back({o.p_i[q], o.p_col[q], o.p_j[q]});
  }


  template <>
  void OfflineData<dim, Number>::create_multigrid_data()
  {
  }


  template <>
  InitialValues<Description, dim, Number>::InitialValues(const HyperbolicSystem &hyperbolic_system,
                                                         const OfflineData<dim, Number> &offline_data,
                                                         const std::string &subsection)
      : ParameterAcceptor(subsection)
      , hyperbolic_system_(&hyperbolic_system)
      , offline_data_(&offline_data)
  {
    /* "configuration = uniform", primitive state (1.4, 3, 1) along +x (initial_state_uniform.h:36-38) */
    initial_state_ = [](const dealii::Point<dim> &, Number) {
      state_type U;
      U[0] = 1.4;
      U[1] = 1.4 * 3.;
      U[2] = 0.;
      U[3] = 1. / 0.4 + 0.5 * 1.4 * 9.;
      return U;
    };
    initial_precomputed_ = [](const dealii::Point<dim> &) { return initial_precomputed_type(); };
  }


  template <>
  InitialValues<Description, dim, Number>::InitialPrecomputedVector
  InitialValues<Description, dim, Number>::interpolate_initial_precomputed_vector() const
  {
    InitialPrecomputedVector v;
    v.reinit_with_scalar_partitioner(offline_data_->scalar_partitioner());
    return v;
  }
} // namespace ryujin


/* the two deal.II entry points the exporter calls beyond the containers (stand-ins: the support points are the
 * generator's, one rank: global = local index) */
namespace dealii
{
  template <>
  DoFHandler<2, 2>::DoFHandler()
  {
  }
  namespace DoFTools
  {
    template <>
    void map_dofs_to_support_points<2, 2>(const Mapping<2, 2> &, const DoFHandler<2, 2> &,
                                          std::map<types::global_dof_index, Point<2>> &out)
    {
      for (unsigned int i = 0; i < ryujin::g_offline->n_relevant; ++i) {
        Point<2> x;
        for (int d = 0; d < 2; ++d)
          x[d] = ryujin::g_positions[i * 2 + d];
        out[i] = x;
      }
    }
  } // namespace DoFTools
} // namespace dealii


static unsigned long long fnv1a(const void *data, std::size_t bytes)
{
  const unsigned char *p = static_cast<const unsigned char *>(data);
  unsigned long long h = 1469598103934665603ull;
  for (std::size_t i = 0; i < bytes; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}


int main(int argc, char **argv)
{
  using namespace ryujin;
  const std::string scheme = argc > 1 ? argv[1] : "ssprk33";
  const bool export_mode = scheme == "export";
  const std::string export_prefix = export_mode && argc > 2 ? argv[2] : "offline";
  const std::string export_mesh = export_mode && argc > 3 ? argv[3] : "box:16";
  const int n_steps = !export_mode && argc > 2 ? std::atoi(argv[2]) : 6;
  const bool device_resident = !export_mode && argc > 3 && std::atoi(argv[3]) != 0;
  const int n_cells = export_mode ? std::atoi(export_mesh.substr(export_mesh.find(':') + 1).c_str())
                                  : (argc > 4 ? std::atoi(argv[4]) : 64);

  ryujin_synth_spec spec{};
  spec.dim = 2;
  spec.n_cells[0] = spec.n_cells[1] = n_cells;
  spec.n_cells[2] = 1;
  spec.upper[0] = spec.upper[1] = 20.;
  for (int f = 0; f < 4; ++f)
    spec.bc[f] = RYUJIN_BC_SLIP;
  spec.n_ranks = 1;
  if (export_mode && export_mesh.rfind("step:", 0) == 0) {
    /* ryujin_amd/offline.py: mach3_step_2d(n) -- [0,3]x[0,1] minus [0.6,3]x[0,0.2] */
    spec.n_cells[0] = 3 * n_cells;
    spec.upper[0] = 3.;
    spec.upper[1] = 1.;
    spec.bc[0] = RYUJIN_BC_DIRICHLET;
    spec.bc[1] = RYUJIN_BC_DO_NOTHING;
    spec.cut_kind = RYUJIN_CUT_BOX;
    spec.cut_lo[0] = 0.6;
    spec.cut_lo[1] = -1.;
    spec.cut_hi[0] = 4.;
    spec.cut_hi[1] = 0.2;
    spec.cut_bc = RYUJIN_BC_SLIP;
  }
  ryujin_synth *mesh = ryujin_synth_build(&spec);
  if (!mesh) {
    std::fprintf(stderr, "%s\n", ryujin_synth_last_error());
    return 1;
  }
  g_offline = ryujin_synth_offline(mesh);
  g_positions = ryujin_synth_positions(mesh);
  g_bdry_positions = ryujin_synth_bdry_positions(mesh);

  int status = 0;
  try {
    const MPI_Comm mpi_communicator = MPI_COMM_WORLD;
    std::map<std::string, dealii::Timer> computing_timer;

    /* the members of TimeLoop (time_loop.h:180-204), constructed in its order */
    Description::HyperbolicSystem hyperbolic_system("/B - Equation");
    Description::ParabolicSystem parabolic_system("/B - Equation");
    Discretization<dim> discretization(mpi_communicator, "/C - Discretization");
    OfflineData<dim, Number> offline_data(mpi_communicator, discretization, "/D - OfflineData");
    InitialValues<Description, dim, Number> initial_values(hyperbolic_system, offline_data, "/E - InitialValues");
    HyperbolicModule<Description, dim, Number> hyperbolic_module(
        mpi_communicator, computing_timer, offline_data, hyperbolic_system, initial_values, "/F - HyperbolicModule");
    ParabolicModule<Description, dim, Number> parabolic_module(mpi_communicator,
                                                               computing_timer,
                                                               offline_data,
                                                               hyperbolic_system,
                                                               parabolic_system,
                                                               initial_values,
                                                               "/G - ParabolicModule");
    TimeIntegrator<Description, dim, Number> time_integrator(
        mpi_communicator, offline_data, hyperbolic_module, parabolic_module, "/H - TimeIntegrator");

    /* what the prm file sets (tests/euler/check-mass-conservation_01.prm): "cfl min = cfl max = 0.9",
     * "cfl recovery strategy = none", "time stepping scheme = ..." -- through the ParameterAcceptor interface */
    dealii::ParameterAcceptor::prm.set("/H - TimeIntegrator", "cfl min", "0.9");
    dealii::ParameterAcceptor::prm.set("/H - TimeIntegrator", "cfl max", "0.9");
    dealii::ParameterAcceptor::prm.set("/H - TimeIntegrator", "cfl recovery strategy", "none");
    const std::map<std::string, std::string> names = {{"ssprk22", "ssprk 22"}, {"ssprk33", "ssprk 33"},
                                                      {"erk11", "erk 11"},     {"erk22", "erk 22"},
                                                      {"erk33", "erk 33"},     {"erk43", "erk 43"},
                                                      {"erk54", "erk 54"}};
    dealii::ParameterAcceptor::prm.set(
        "/H - TimeIntegrator", "time stepping scheme", export_mode ? std::string("ssprk 33") : names.at(scheme));
    dealii::ParameterAcceptor::prm.set(
        "/F - HyperbolicModule", "hip device resident state vectors", device_resident ? "true" : "false");
    if (std::getenv("RYUJIN_TEST_PIN"))
      dealii::ParameterAcceptor::prm.set("/F - HyperbolicModule", "hip pin host vectors", "true");
    if (std::getenv("RYUJIN_TEST_NO_DERIVED"))
      dealii::ParameterAcceptor::prm.set("/F - HyperbolicModule", "hip mirror precomputed values", "false");
    dealii::ParameterAcceptor::initialize();

    /* TimeLoop::run() (time_loop.template.h:240-300) */
    const auto prec = Description::HyperbolicSystemView<dim, Number>::n_precomputed_values;
    offline_data.prepare(HyperbolicModule<Description, dim, Number>::problem_dimension, prec);
    if (export_mode) {
      /* TimeLoop::run() with contrib/ryujin_export_offline.patch: behind OfflineData::prepare() */
      export_offline_data(offline_data, export_prefix, mpi_communicator);
      std::printf("exported %u rows to %s-0.ryjoffl\n", offline_data.n_locally_owned(), export_prefix.c_str());
      ryujin_synth_free(mesh);
      return 0;
    }
    hyperbolic_module.prepare();
    parabolic_module.prepare();
    time_integrator.prepare();

    using StateVector = HyperbolicModule<Description, dim, Number>::StateVector;
    StateVector state_vector;
    Vectors::reinit_state_vector<Description>(state_vector, offline_data);
    auto &U = std::get<0>(state_vector);
    const unsigned int n_owned = offline_data.n_locally_owned();
    for (unsigned int i = 0; i < n_owned; ++i)
      U.write_tensor(initial_values.initial_state(dealii::Point<dim>(), 0.), i);

    const auto &lumped_mass_matrix = offline_data.lumped_mass_matrix();
    Number t = 0.;
    dealii::Timer wall;
    for (int cycle = 0; cycle < n_steps; ++cycle) {
      if (cycle >= 1)
        wall.start(); /* (the first step creates the twins) */
      t += time_integrator.step(state_vector, t);
      if (cycle >= 1)
        wall.stop();

      /* Quantities::accumulate reads the state vector (time_loop.template.h:311); with the patched TimeLoop it is
       * fetched first */
#ifdef RYUJIN_TEST_PATCHED_TIME_LOOP
      hyperbolic_module.synchronize_to_host(state_vector);
#endif
      auto &U_now = std::get<0>(state_vector);
      Number mass = 0., rho = 0.;
      for (unsigned int i = 0; i < n_owned; ++i) {
        mass += lumped_mass_matrix.local_element(i);
        rho += lumped_mass_matrix.local_element(i) * U_now.get_tensor(i)[0];
      }
      std::printf("%.14e %.14e\n", t, rho / mass);
    }
    if (n_steps > 1) /* stderr: the stdout of two runs is compared byte for byte */
      std::fprintf(stderr, "gridpoints %u ms_per_time_step %.4f (TimeIntegrator::step alone, steps 2..%d)\n", n_owned,
                   wall.wall_time() / (n_steps - 1) * 1e3, n_steps);
    auto &U_final = std::get<0>(state_vector);
    std::printf("checksum %016llx\n", fnv1a(U_final.begin(), std::size_t(n_owned) * 4 * sizeof(Number)));
    std::printf("n_restarts %u n_warnings %u\n", hyperbolic_module.n_restarts(), hyperbolic_module.n_warnings());
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    status = 2;
  }
  ryujin_synth_free(mesh);
  return status;
}

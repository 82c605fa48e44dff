//
// ryujin_export_offline.h -- OfflineData exporter for the ryujin source tree (drop into source/).
//
// Writes, one file per MPI rank, exactly the arrays the MI355X hot path consumes (include/ryujin_hip.h,
// struct ryujin_hip_offline; wire format include/ryujin_offline_io.h), so that any mesh ryujin can set up --
// curved boundaries (geometry_step.h rounded corner, geometry_cylinder.h), locally refined or unstructured
// triangulations, any partition -- runs through libryujin_hip.so unchanged, and so that a run on a machine
// WITHOUT deal.II can be validated against a real ryujin run elsewhere.
//
// Uses ONLY public interfaces of the reference (no accessor has to be added to any class):
//   OfflineData<dim,Number>           source/offline_data.h:121-264   (index ranges, boundary_map,
//                                     coupling_boundary_pairs, matrices, lumped masses, partitioner)
//   SparsityPatternSIMD<simd_length>  source/sparse_matrix_simd.h:96-106 (columns(row), row_length(row),
//                                     stride_of_row(row), n_rows())
//   SparseMatrixSIMD<Number,n>        source/sparse_matrix_simd.h:203-221 (get_entry / get_tensor)
//   dealii::Utilities::MPI::Partitioner (import_targets, import_indices, ghost_targets)
// The matrices are written as plain diagonal-first CSR (simd_length = 1): get_entry()/get_tensor() undo the
// SIMD interleave of rows [0, n_internal) (sparse_matrix_simd.h:403-418), the library re-tiles for 64-wide
// wavefronts anyway. The protected members SparsityPatternSIMD::send_targets / entries_to_be_sent are not
// needed either: the lists are rebuilt from the partitioner by the rule of
// sparse_matrix_simd.template.h:196-264, stated once in include/ryujin_exchange_lists.h (pinned against
// tests/common/sparsity_pattern_simd_01 in tests/test_send_lists_golden.py of the ryujin_amd repository).
// The loops that walk OfflineData are the ones of the in-memory adapter (contrib/hyperbolic_module_hip.h): both
// call ryujin_hip_binding::fill_from_accessors() (contrib/ryujin_hip_binding.h), which is compiled and tested
// without deal.II against a mock with the reference's accessor names (tests/test_binding_cpp.py).
//
// Link against libryujin_synth.so (host-only C++, provides ryujin_offline_write) and add the include directory
// of ryujin_amd (ryujin_hip.h, ryujin_offline_io.h). Call after OfflineData::prepare():
//
//   ryujin::export_offline_data(offline_data, base_name + "-offline", mpi_communicator);
//
// (contrib/ryujin_export_offline.patch adds exactly this call to TimeLoop::run behind a run-time parameter.)
//
#pragma once

#include "offline_data.h"

#include <deal.II/base/mpi.h>
#include <deal.II/dofs/dof_tools.h>

#include <ryujin_offline_io.h>  /* from ryujin_amd/include */
#include "ryujin_hip_binding.h" /* from ryujin_amd/contrib: fill_from_accessors(), shared with the adapter */

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace ryujin
{
  /**
   * Write `<prefix>-<rank>.ryjoffl`.
   *
   * @param initial_precomputed  optional [n_locally_relevant * n_initial_precomputed] values in local
   *        numbering (shallow water: the bathymetry from
   *        InitialValues::interpolate_initial_precomputed_vector(), hyperbolic_module.template.h:84-85)
   */
  template <int dim, typename Number>
  void export_offline_data(const OfflineData<dim, Number> &offline_data,
                           const std::string &prefix,
                           const MPI_Comm &mpi_communicator,
                           const double *initial_precomputed = nullptr,
                           const unsigned int n_initial_precomputed = 0)
  {
    static_assert(std::is_same<Number, double>::value, "the hot path computes in double");

    /* the same loops that fill the struct in memory for hyperbolic_module_hip.h */
    ryujin_hip_binding::OfflineArrays arrays;
    ryujin_hip_binding::fill_from_accessors<dim>(offline_data, arrays, initial_precomputed, n_initial_precomputed);

    /* ---- support points (optional payload: lets the importing side evaluate initial / Dirichlet data) ---- */

    const auto &partitioner = *offline_data.scalar_partitioner();
    const unsigned int n_relevant = offline_data.n_locally_relevant();
    std::vector<double> positions(static_cast<std::size_t>(n_relevant) * dim, 0.);
    {
      std::map<dealii::types::global_dof_index, dealii::Point<dim>> support_points;
      dealii::DoFTools::map_dofs_to_support_points(offline_data.discretization().mapping(),
                                                   offline_data.dof_handler(),
                                                   support_points);
      for (unsigned int i = 0; i < n_relevant; ++i) {
        const auto it = support_points.find(partitioner.local_to_global(i));
        if (it != support_points.end())
          for (unsigned int d = 0; d < dim; ++d)
            positions[static_cast<std::size_t>(i) * dim + d] = it->second[d];
      }
    }

    const unsigned int rank = dealii::Utilities::MPI::this_mpi_process(mpi_communicator);
    const std::string path = prefix + "-" + std::to_string(rank) + ".ryjoffl";
    if (ryujin_offline_write(path.c_str(), &arrays.offline, dim, n_initial_precomputed, positions.data(),
                             arrays.b_positions.data()) != 0)
      throw std::runtime_error(std::string("export_offline_data: ") + ryujin_offline_io_last_error());
  }
} // namespace ryujin

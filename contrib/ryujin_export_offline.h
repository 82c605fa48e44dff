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
// needed either: the lists are rebuilt here from the partitioner by the rule of
// sparse_matrix_simd.template.h:196-264 (pinned against tests/common/sparsity_pattern_simd_01 in
// tests/test_send_lists_golden.py of the ryujin_amd repository).
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

#include <ryujin_offline_io.h> /* from ryujin_amd/include */

#include <algorithm>
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

    const auto &sparsity = offline_data.sparsity_pattern_simd();
    const auto &cij_matrix = offline_data.cij_matrix();
    const auto &mass_matrix = offline_data.mass_matrix();
    const auto &lumped_mass_matrix = offline_data.lumped_mass_matrix();
    const auto &lumped_mass_matrix_inverse = offline_data.lumped_mass_matrix_inverse();
    const auto &partitioner = *offline_data.scalar_partitioner();

    const unsigned int n_owned = offline_data.n_locally_owned();
    const unsigned int n_relevant = offline_data.n_locally_relevant();

    /* ---- stencil: diagonal-first CSR through the public accessors ---- */

    std::vector<uint64_t> row_starts(n_relevant + 1, 0);
    for (unsigned int i = 0; i < n_relevant; ++i)
      row_starts[i + 1] = row_starts[i] + sparsity.row_length(i);
    const std::size_t nnz = row_starts[n_relevant];

    std::vector<uint32_t> columns(nnz);
    std::vector<double> cij(nnz * dim), mij(nnz);
    const bool dg = offline_data.discretization().have_discontinuous_ansatz();
    std::vector<double> incidence(dg ? nnz : 0), mass_matrix_inverse(dg ? nnz : 0);

    for (unsigned int i = 0; i < n_relevant; ++i) {
      const unsigned int *js = sparsity.columns(i);
      const unsigned int stride = sparsity.stride_of_row(i);
      const unsigned int row_length = sparsity.row_length(i);
      for (unsigned int col_idx = 0; col_idx < row_length; ++col_idx) {
        const std::size_t e = row_starts[i] + col_idx;
        columns[e] = js[col_idx * stride];
        const auto c_ij = cij_matrix.template get_tensor<Number>(i, col_idx);
        for (unsigned int d = 0; d < dim; ++d)
          cij[e * dim + d] = c_ij[d];
        mij[e] = mass_matrix.template get_entry<Number>(i, col_idx);
        if (dg) {
          incidence[e] = offline_data.incidence_matrix().template get_entry<Number>(i, col_idx);
          mass_matrix_inverse[e] =
              offline_data.mass_matrix_inverse().template get_entry<Number>(i, col_idx);
        }
      }
      if (row_length == 0 || columns[row_starts[i]] != i)
        throw std::runtime_error("export_offline_data: row does not start with its diagonal");
    }

    std::vector<double> mi(n_relevant), mi_inv(n_relevant);
    for (unsigned int i = 0; i < n_relevant; ++i) {
      mi[i] = lumped_mass_matrix.local_element(i);
      mi_inv[i] = lumped_mass_matrix_inverse.local_element(i);
    }

    /* ---- boundary map and coupling pairs, SoA in container (= application) order ---- */

    const auto &boundary_map = offline_data.boundary_map();
    std::vector<uint32_t> b_i;
    std::vector<double> b_normal, b_positions;
    std::vector<uint8_t> b_id;
    for (const auto &entry : boundary_map) {
      const auto &[i, normal, normal_mass, boundary_mass, id, position] = entry;
      (void)normal_mass;
      (void)boundary_mass;
      if (i >= n_owned) /* cannot happen: construct_boundary_map stores locally owned DoFs only */
        continue;       /* (offline_data.template.h:1259-1261) */
      b_i.push_back(i);
      for (unsigned int d = 0; d < dim; ++d) {
        b_normal.push_back(normal[d]);
        b_positions.push_back(position[d]);
      }
      /* ryujin::Boundary (discretization.h) and RYUJIN_BC_* (ryujin_hip.h) enumerate alike:
       * do_nothing 0, periodic 1, slip 2, no_slip 3, dirichlet 4, dynamic 5, dirichlet_momentum 6 */
      b_id.push_back(static_cast<uint8_t>(id));
    }

    const auto &coupling_boundary_pairs = offline_data.coupling_boundary_pairs();
    std::vector<uint32_t> p_i, p_col, p_j;
    for (const auto &[i, col_idx, j] : coupling_boundary_pairs) {
      p_i.push_back(i);
      p_col.push_back(col_idx);
      p_j.push_back(j);
    }

    /* ---- exchange pattern from the scalar partitioner ---- */

    /* ghost ranges by owner: ghosts are stored sorted by owner rank (dealii Partitioner) */
    std::map<unsigned int, std::pair<uint32_t, uint32_t>> ghost_range; /* rank -> [begin, end) local */
    {
      uint32_t begin = n_owned;
      for (const auto &[rank, count] : partitioner.ghost_targets()) {
        ghost_range[rank] = {begin, begin + count};
        begin += count;
      }
      if (begin != n_relevant)
        throw std::runtime_error("export_offline_data: ghost targets do not cover the ghost range");
    }
    /* import (= export, in ryujin's words) indices by target rank, in the partitioner's order */
    std::map<unsigned int, std::vector<uint32_t>> send_rows;
    {
      std::vector<uint32_t> flat;
      for (const auto &range : partitioner.import_indices())
        for (unsigned int i = range.first; i < range.second; ++i)
          flat.push_back(i);
      std::size_t pos = 0;
      for (const auto &[rank, count] : partitioner.import_targets()) {
        auto &rows = send_rows[rank];
        rows.assign(flat.begin() + pos, flat.begin() + pos + count);
        pos += count;
      }
    }
    std::vector<int> nbr_rank;
    for (const auto &it : ghost_range)
      nbr_rank.push_back(it.first);
    for (const auto &it : send_rows)
      if (!ghost_range.count(it.first))
        nbr_rank.push_back(it.first);
    std::sort(nbr_rank.begin(), nbr_rank.end());

    std::vector<uint32_t> send_off{0}, send_idx, recv_off, row_send_off{0}, row_send_row, row_send_col;
    {
      uint32_t cursor = n_owned;
      for (const int rank : nbr_rank) {
        recv_off.push_back(cursor);
        if (ghost_range.count(rank))
          cursor = ghost_range[rank].second;
      }
      recv_off.push_back(cursor);
    }
    for (const int rank : nbr_rank) {
      const auto rows = send_rows.count(rank) ? send_rows[rank] : std::vector<uint32_t>();
      send_idx.insert(send_idx.end(), rows.begin(), rows.end());
      send_off.push_back(send_idx.size());
      /* ghost rows hold the diagonal and the transposes of owned entries only: for every exported row
       * its diagonal and the entries whose column lies in the ghost range received from the same rank
       * (sparse_matrix_simd.template.h:249-261); a rank that sends us nothing gets nothing (:229-247) */
      if (ghost_range.count(rank)) {
        const auto [lo, hi] = ghost_range[rank];
        for (const uint32_t row : rows) {
          row_send_row.push_back(row);
          row_send_col.push_back(0);
          for (uint64_t e = row_starts[row] + 1; e < row_starts[row + 1]; ++e)
            if (columns[e] >= lo && columns[e] < hi) {
              row_send_row.push_back(row);
              row_send_col.push_back(static_cast<uint32_t>(e - row_starts[row]));
            }
        }
      }
      row_send_off.push_back(row_send_row.size());
    }

    /* ---- support points (optional payload: lets the importing side evaluate initial / Dirichlet data) ---- */

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

    /* ---- fill the struct and write ---- */

    ryujin_hip_offline o{};
    o.n_export = offline_data.n_export_indices();
    o.n_internal = offline_data.n_locally_internal();
    o.n_owned = n_owned;
    o.n_relevant = n_relevant;
    o.simd_length = 1; /* plain CSR, see above */
    o.row_starts = row_starts.data();
    o.columns = columns.data();
    o.cij = cij.data();
    o.mij = mij.data();
    o.mi = mi.data();
    o.mi_inv = mi_inv.data();
    o.measure_of_omega = offline_data.measure_of_omega();
    o.n_bdry = b_i.size();
    o.b_i = b_i.data();
    o.b_normal = b_normal.data();
    o.b_id = b_id.data();
    o.n_pairs = p_i.size();
    o.p_i = p_i.data();
    o.p_col = p_col.data();
    o.p_j = p_j.data();
    o.initial_precomputed = n_initial_precomputed ? initial_precomputed : nullptr;
    o.n_nbr = nbr_rank.size();
    o.nbr_rank = nbr_rank.data();
    o.send_off = send_off.data();
    o.send_idx = send_idx.data();
    o.recv_off = recv_off.data();
    o.row_send_off = row_send_off.data();
    o.row_send_row = row_send_row.data();
    o.row_send_col = row_send_col.data();
    o.discontinuous_ansatz = dg ? 1 : 0;
    o.incidence = dg ? incidence.data() : nullptr;
    o.mass_matrix_inverse = dg ? mass_matrix_inverse.data() : nullptr;

    const unsigned int rank = dealii::Utilities::MPI::this_mpi_process(mpi_communicator);
    const std::string path = prefix + "-" + std::to_string(rank) + ".ryjoffl";
    if (ryujin_offline_write(path.c_str(), &o, dim, n_initial_precomputed, positions.data(),
                             b_positions.data()) != 0)
      throw std::runtime_error(std::string("export_offline_data: ") + ryujin_offline_io_last_error());
  }
} // namespace ryujin

//
// ryujin_hip_binding.h -- the deal.II-free half of the ryujin-side binding of libryujin_hip.so.
//
// Everything the adapter class (contrib/hyperbolic_module_hip.h) and the exporter
// (contrib/ryujin_export_offline.h) do that does not need a deal.II type lives here as templates over the
// ACCESSOR NAMES of the reference, so that it is compiled and tested without deal.II against a mock that
// serves synthetic data through the same names (tests/cpp/binding_fill.cc, tests/test_binding_cpp.py):
//
//   fill_from_accessors<dim>(offline_data, arrays)   OfflineData -> struct ryujin_hip_offline, IN MEMORY
//       uses only public interfaces of the reference:
//         OfflineData<dim,Number>           source/offline_data.h:121-264
//         SparsityPatternSIMD<simd_length>  source/sparse_matrix_simd.h:96-106  (columns, row_length, stride_of_row)
//         SparseMatrixSIMD<Number,n>        source/sparse_matrix_simd.h:203-221 (get_entry, get_tensor)
//         dealii::Utilities::MPI::Partitioner (ghost_targets, import_targets, import_indices)
//       matrices as plain diagonal-first CSR (simd_length = 1): get_entry / get_tensor undo the SIMD interleave
//       (sparse_matrix_simd.h:403-418); the library re-tiles for 64-wide wavefronts anyway. The ghost-row send
//       lists come from THE statement of the rule (include/ryujin_exchange_lists.h).
//   fill_params_euler / _shallow_water / _common      ParameterAcceptor values -> ryujin_hip_params
//       (source/euler/hyperbolic_system.h:135-160, euler/{indicator,limiter,riemann_solver}.h,
//        source/shallow_water/hyperbolic_system.h:132-165, shallow_water/limiter.h:62-68)
//   HandleCache                                       storage of a host StateVector -> device-resident twin
//
#pragma once

#include <ryujin_exchange_lists.h>
#include <ryujin_hip.h>

#include <algorithm>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace ryujin_hip_binding
{
  /* owning storage behind a ryujin_hip_offline */
  struct OfflineArrays {
    std::vector<uint64_t> row_starts;
    std::vector<uint32_t> columns;
    std::vector<double> cij, mij, mi, mi_inv, incidence, mass_matrix_inverse, initial_precomputed;
    std::vector<uint32_t> b_i;
    std::vector<double> b_normal, b_positions;
    std::vector<uint8_t> b_id;
    std::vector<uint32_t> p_i, p_col, p_j;
    std::vector<int> nbr_rank;
    std::vector<uint32_t> send_off, send_idx, recv_off, row_send_off, row_send_row, row_send_col;
    ryujin_hip_offline offline{}; /* pointers into the vectors above: valid while *this is alive and unmodified */
  };


  /**
   * OfflineData (after OfflineData::prepare()) -> flat arrays in local numbering, plain CSR.
   * @param initial_precomputed [n_locally_relevant * n_initial_precomputed] or nullptr (shallow water: the
   *        bathymetry from InitialValues::interpolate_initial_precomputed_vector(),
   *        hyperbolic_module.template.h:84-85)
   */
  template <int dim, typename OfflineDataType>
  void fill_from_accessors(const OfflineDataType &offline_data, OfflineArrays &a,
                           const double *initial_precomputed = nullptr,
                           const unsigned int n_initial_precomputed = 0)
  {
    using Number = double; /* the hot path computes in double (CMakeLists.txt:69 NUMBER=double) */

    const auto &sparsity = offline_data.sparsity_pattern_simd();
    const auto &cij_matrix = offline_data.cij_matrix();
    const auto &mass_matrix = offline_data.mass_matrix();
    const auto &lumped_mass_matrix = offline_data.lumped_mass_matrix();
    const auto &lumped_mass_matrix_inverse = offline_data.lumped_mass_matrix_inverse();
    const auto &partitioner = *offline_data.scalar_partitioner();

    const unsigned int n_owned = offline_data.n_locally_owned();
    const unsigned int n_relevant = offline_data.n_locally_relevant();

    /* ---- stencil: diagonal-first CSR through the public accessors ---- */

    a.row_starts.assign(n_relevant + 1, 0);
    for (unsigned int i = 0; i < n_relevant; ++i)
      a.row_starts[i + 1] = a.row_starts[i] + sparsity.row_length(i);
    const std::size_t nnz = a.row_starts[n_relevant];

    a.columns.resize(nnz);
    a.cij.resize(nnz * dim);
    a.mij.resize(nnz);
    const bool dg = offline_data.discretization().have_discontinuous_ansatz();
    a.incidence.assign(dg ? nnz : 0, 0.);
    a.mass_matrix_inverse.assign(dg ? nnz : 0, 0.);

    for (unsigned int i = 0; i < n_relevant; ++i) {
      const unsigned int *js = sparsity.columns(i);
      const unsigned int stride = sparsity.stride_of_row(i);
      const unsigned int row_length = sparsity.row_length(i);
      for (unsigned int col_idx = 0; col_idx < row_length; ++col_idx) {
        const std::size_t e = a.row_starts[i] + col_idx;
        a.columns[e] = js[col_idx * stride];
        const auto c_ij = cij_matrix.template get_tensor<Number>(i, col_idx);
        for (unsigned int d = 0; d < dim; ++d)
          a.cij[e * dim + d] = c_ij[d];
        a.mij[e] = mass_matrix.template get_entry<Number>(i, col_idx);
        if (dg) {
          a.incidence[e] = offline_data.incidence_matrix().template get_entry<Number>(i, col_idx);
          a.mass_matrix_inverse[e] = offline_data.mass_matrix_inverse().template get_entry<Number>(i, col_idx);
        }
      }
      if (row_length == 0 || a.columns[a.row_starts[i]] != i)
        throw std::runtime_error("ryujin_hip_binding: row " + std::to_string(i) +
                                 " does not start with its diagonal");
    }

    a.mi.resize(n_relevant);
    a.mi_inv.resize(n_relevant);
    for (unsigned int i = 0; i < n_relevant; ++i) {
      a.mi[i] = lumped_mass_matrix.local_element(i);
      a.mi_inv[i] = lumped_mass_matrix_inverse.local_element(i);
    }

    /* ---- boundary map and coupling pairs, SoA in container (= application) order ---- */

    a.b_i.clear();
    a.b_normal.clear();
    a.b_positions.clear();
    a.b_id.clear();
    for (const auto &entry : offline_data.boundary_map()) {
      /* (i, normal, normal_mass, boundary_mass, id, position), offline_data.h:66-72 */
      const unsigned int i = std::get<0>(entry);
      const auto &normal = std::get<1>(entry);
      const auto &position = std::get<5>(entry);
      if (i >= n_owned) /* cannot happen: construct_boundary_map stores locally owned DoFs only */
        continue;       /* (offline_data.template.h:1259-1261) */
      a.b_i.push_back(i);
      for (unsigned int d = 0; d < dim; ++d) {
        a.b_normal.push_back(normal[d]);
        a.b_positions.push_back(position[d]);
      }
      /* ryujin::Boundary (discretization.h:28-112) and RYUJIN_BC_* (ryujin_hip.h) enumerate alike:
       * do_nothing 0, periodic 1, slip 2, no_slip 3, dirichlet 4, dynamic 5, dirichlet_momentum 6 */
      a.b_id.push_back(static_cast<uint8_t>(std::get<4>(entry)));
    }

    a.p_i.clear();
    a.p_col.clear();
    a.p_j.clear();
    for (const auto &pair : offline_data.coupling_boundary_pairs()) {
      a.p_i.push_back(std::get<0>(pair));
      a.p_col.push_back(std::get<1>(pair));
      a.p_j.push_back(std::get<2>(pair));
    }

    /* ---- exchange pattern from the scalar partitioner ---- */

    /* ghost ranges by owner: ghosts are stored sorted by owner rank (dealii Partitioner) */
    std::map<unsigned int, std::pair<uint32_t, uint32_t>> ghost_range; /* rank -> [begin, end) local */
    {
      uint32_t begin = n_owned;
      for (const auto &target : partitioner.ghost_targets()) {
        ghost_range[target.first] = {begin, begin + target.second};
        begin += target.second;
      }
      if (begin != n_relevant)
        throw std::runtime_error("ryujin_hip_binding: ghost targets do not cover the ghost range");
    }
    /* import (= export, in ryujin's words) indices by target rank, in the partitioner's order */
    std::map<unsigned int, std::vector<uint32_t>> send_rows;
    {
      std::vector<uint32_t> flat;
      for (const auto &range : partitioner.import_indices())
        for (unsigned int i = range.first; i < range.second; ++i)
          flat.push_back(i);
      std::size_t pos = 0;
      for (const auto &target : partitioner.import_targets()) {
        auto &rows = send_rows[target.first];
        rows.assign(flat.begin() + pos, flat.begin() + pos + target.second);
        pos += target.second;
      }
    }
    a.nbr_rank.clear();
    for (const auto &it : ghost_range)
      a.nbr_rank.push_back(it.first);
    for (const auto &it : send_rows)
      if (!ghost_range.count(it.first))
        a.nbr_rank.push_back(it.first);
    std::sort(a.nbr_rank.begin(), a.nbr_rank.end());

    a.send_off.assign(1, 0);
    a.send_idx.clear();
    a.recv_off.clear();
    a.row_send_off.assign(1, 0);
    a.row_send_row.clear();
    a.row_send_col.clear();
    {
      uint32_t cursor = n_owned;
      for (const int rank : a.nbr_rank) {
        a.recv_off.push_back(cursor);
        if (ghost_range.count(rank))
          cursor = ghost_range[rank].second;
      }
      a.recv_off.push_back(cursor);
    }
    for (const int rank : a.nbr_rank) {
      const std::vector<uint32_t> rows = send_rows.count(rank) ? send_rows[rank] : std::vector<uint32_t>();
      a.send_idx.insert(a.send_idx.end(), rows.begin(), rows.end());
      a.send_off.push_back(static_cast<uint32_t>(a.send_idx.size()));
      /* ghost rows: the rule of sparse_matrix_simd.template.h:196-264, stated once */
      const auto range = ghost_range.count(rank) ? ghost_range[rank] : std::pair<uint32_t, uint32_t>{0, 0};
      const std::size_t n_entries =
          ryujin_ghost_row_send_entries(a.row_starts.data(), a.columns.data(), rows.data(), rows.size(),
                                        range.first, range.second, nullptr, nullptr);
      const std::size_t first = a.row_send_row.size();
      a.row_send_row.resize(first + n_entries);
      a.row_send_col.resize(first + n_entries);
      ryujin_ghost_row_send_entries(a.row_starts.data(), a.columns.data(), rows.data(), rows.size(), range.first,
                                    range.second, a.row_send_row.data() + first, a.row_send_col.data() + first);
      a.row_send_off.push_back(static_cast<uint32_t>(a.row_send_row.size()));
    }

    if (n_initial_precomputed != 0 && initial_precomputed != nullptr)
      a.initial_precomputed.assign(initial_precomputed,
                                   initial_precomputed + std::size_t(n_relevant) * n_initial_precomputed);
    else
      a.initial_precomputed.clear();

    /* ---- the view ---- */

    ryujin_hip_offline &o = a.offline;
    o = ryujin_hip_offline{};
    o.n_export = offline_data.n_export_indices();
    o.n_internal = offline_data.n_locally_internal();
    o.n_owned = n_owned;
    o.n_relevant = n_relevant;
    o.simd_length = 1; /* plain CSR, see above */
    o.row_starts = a.row_starts.data();
    o.columns = a.columns.data();
    o.cij = a.cij.data();
    o.mij = a.mij.data();
    o.mi = a.mi.data();
    o.mi_inv = a.mi_inv.data();
    o.measure_of_omega = offline_data.measure_of_omega();
    o.n_bdry = static_cast<uint32_t>(a.b_i.size());
    o.b_i = a.b_i.data();
    o.b_normal = a.b_normal.data();
    o.b_id = a.b_id.data();
    o.n_pairs = static_cast<uint32_t>(a.p_i.size());
    o.p_i = a.p_i.data();
    o.p_col = a.p_col.data();
    o.p_j = a.p_j.data();
    o.initial_precomputed = a.initial_precomputed.empty() ? nullptr : a.initial_precomputed.data();
    o.n_nbr = static_cast<int>(a.nbr_rank.size());
    o.nbr_rank = a.nbr_rank.data();
    o.send_off = a.send_off.data();
    o.send_idx = a.send_idx.data();
    o.recv_off = a.recv_off.data();
    o.row_send_off = a.row_send_off.data();
    o.row_send_row = a.row_send_row.data();
    o.row_send_col = a.row_send_col.data();
    o.discontinuous_ansatz = dg ? 1 : 0;
    o.incidence = dg ? a.incidence.data() : nullptr;
    o.mass_matrix_inverse = dg ? a.mass_matrix_inverse.data() : nullptr;
  }


  /* ---- ParameterAcceptor values -> ryujin_hip_params ------------------------------------------- */

  /* "/HyperbolicModule/{indicator,limiter,riemann solver}" of the Euler Description
   * (euler/indicator.h:34, euler/limiter.h:51-54, euler/riemann_solver.h:41-42) */
  template <typename IndicatorParameters, typename LimiterParameters>
  void fill_params_indicator_limiter(ryujin_hip_params &p, const IndicatorParameters &indicator,
                                     const LimiterParameters &limiter)
  {
    p.indicator_evc_factor = indicator.evc_factor();
    p.limiter_iterations = static_cast<int>(limiter.iterations());
    p.limiter_newton_tolerance = limiter.newton_tolerance();
    p.limiter_newton_max_iterations = static_cast<int>(limiter.newton_max_iterations());
    p.limiter_relaxation_factor = limiter.relaxation_factor();
    /* ryujin configured with EXPENSIVE_BOUNDS_CHECK (compile_time_options.h.in:12-15; implied by DEBUG): the
     * checked control flow of the limiter and is_admissible behind steps 4, 6, 7 on the device as well (every
     * Description) */
#ifdef EXPENSIVE_BOUNDS_CHECK
    p.debug_expensive_bounds_check = 1;
#endif
  }

  /* (the Riemann solver of the EulerAEOS Description has no run-time parameters, euler_aeos/riemann_solver.h:19-27:
   * it takes fill_params_indicator_limiter() alone) */
  template <typename IndicatorParameters, typename LimiterParameters, typename RiemannSolverParameters>
  void fill_params_common(ryujin_hip_params &p, const IndicatorParameters &indicator,
                          const LimiterParameters &limiter, const RiemannSolverParameters &riemann_solver)
  {
    fill_params_indicator_limiter(p, indicator, limiter);
    p.riemann_newton_tolerance = riemann_solver.newton_tolerance();
    p.riemann_newton_max_iterations = static_cast<int>(riemann_solver.newton_max_iterations());
  }

  /* "B - Equation" of the Euler Description through its view (euler/hyperbolic_system.h:135-160) */
  template <typename View>
  void fill_params_euler(ryujin_hip_params &p, const View &view)
  {
    p.gamma = view.gamma();
    p.reference_density = view.reference_density();
    p.vacuum_state_relaxation_small = view.vacuum_state_relaxation_small();
    p.vacuum_state_relaxation_large = view.vacuum_state_relaxation_large();
  }

  /* "B - Equation" of the shallow-water Description (shallow_water/hyperbolic_system.h:132-165); its Riemann
   * solver has no run-time parameters, its limiter two more (shallow_water/limiter.h:67-68) */
  template <typename View, typename IndicatorParameters, typename LimiterParameters>
  void fill_params_shallow_water(ryujin_hip_params &p, const View &view, const IndicatorParameters &indicator,
                                 const LimiterParameters &limiter)
  {
    p.gravity = view.gravity();
    p.manning_friction_coefficient = view.manning_friction_coefficient();
    p.reference_water_depth = view.reference_water_depth();
    p.dry_state_relaxation_factor = view.dry_state_relaxation_factor();
    p.dry_state_relaxation_small = view.dry_state_relaxation_small();
    p.dry_state_relaxation_large = view.dry_state_relaxation_large();
    p.indicator_evc_factor = indicator.evc_factor();
    p.limiter_iterations = static_cast<int>(limiter.iterations());
    p.limiter_newton_tolerance = limiter.newton_tolerance();
    p.limiter_newton_max_iterations = static_cast<int>(limiter.newton_max_iterations());
    p.limiter_relaxation_factor = limiter.relaxation_factor();
    p.limiter_limit_on_kinetic_energy = limiter.limit_on_kinetic_energy() ? 1 : 0;
    p.limiter_limit_on_square_velocity = limiter.limit_on_square_velocity() ? 1 : 0;
#ifdef EXPENSIVE_BOUNDS_CHECK
    p.debug_expensive_bounds_check = 1;
#endif
  }


  /* ---- host StateVector <-> device-resident twin ------------------------------------------------ */

  /**
   * The reference's StateVector (source/state_vector.h:47-51) lives on the host and is owned by the caller;
   * its device twin lives behind a handle of the library. Twins are keyed by the DATA POINTER of U
   * (std::get<0>(state_vector).begin()), not by the address of the StateVector object: the reference's
   * TimeIntegrator ends every scheme with state_vector.swap(temp_[k]) (time_integrator.template.h:296,325,346 ...),
   * which exchanges the storage of the two tuples and leaves their addresses where they were -- the twin has to
   * follow the storage. A twin is created on first use and released with the cache (prepare() clears it; storage
   * that is reallocated leaves a stale twin behind until then).
   *
   * `device_ahead`: the device copy is NEWER than the host array (left behind by time_step() in device-resident
   * mode); whoever reads the host array next has to fetch it first. Everything else in the adapter treats the
   * host array as the authority.
   */
  class HandleCache
  {
  public:
    struct Twin {
      int handle = -1;
      bool device_ahead = false;
      bool pinned = false; /* the host arrays were handed to ryujin_hip_host_register() */
    };

    HandleCache() = default;
    HandleCache(const HandleCache &) = delete;
    HandleCache &operator=(const HandleCache &) = delete;
    ~HandleCache() { clear(); }

    void reset(ryujin_hip_ctx *ctx)
    {
      clear();
      ctx_ = ctx;
    }

    /* the twin of the storage at `key`; *created is set if it did not exist yet */
    Twin &twin(const void *key, bool *created = nullptr)
    {
      const auto it = twins_.find(key);
      if (created)
        *created = it == twins_.end();
      if (it != twins_.end())
        return it->second;
      int h = -1;
      if (!ctx_ || ryujin_hip_state_alloc(ctx_, &h) != RYUJIN_OK)
        throw std::runtime_error(std::string("ryujin_hip: ") + ryujin_hip_last_error());
      Twin &t = twins_[key];
      t.handle = h;
      return t;
    }

    int handle(const void *key, bool *created = nullptr) { return twin(key, created).handle; }

    std::size_t size() const { return twins_.size(); }

    /* the storage at `key` goes away (or may be handed out again by the allocator): its twin is freed; returns whether
     * the caller's arrays had been pinned for it (they have to be unregistered BEFORE the memory is released) */
    bool forget(const void *key)
    {
      const auto it = twins_.find(key);
      if (it == twins_.end())
        return false;
      const bool pinned = it->second.pinned;
      if (ctx_)
        ryujin_hip_state_free(ctx_, it->second.handle);
      twins_.erase(it);
      return pinned;
    }

    void clear()
    {
      if (ctx_)
        for (const auto &it : twins_)
          ryujin_hip_state_free(ctx_, it.second.handle);
      twins_.clear();
    }

  private:
    ryujin_hip_ctx *ctx_ = nullptr;
    std::map<const void *, Twin> twins_;
  };
} // namespace ryujin_hip_binding

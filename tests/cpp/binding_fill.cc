// The deal.II-free half of the ryujin-side binding (contrib/ryujin_hip_binding.h) compiled and run WITHOUT deal.II:
// a mock serves the arrays of the synthetic generator through the accessor names of the reference's OfflineData
// (source/offline_data.h:121-264), SparsityPatternSIMD (sparse_matrix_simd.h:96-106), SparseMatrixSIMD
// (:203-221) and dealii::Utilities::MPI::Partitioner; fill_from_accessors() must reproduce the generator's own
// ryujin_hip_offline -- index ranges, stencil, matrices, boundary map, coupling pairs and all exchange lists --
// bit for bit, on every rank of a slab partition. Built and run by tests/test_binding_cpp.py.
//
//   binding_fill <dim> <n_ranks> [cells_per_unit]
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <tuple>
#include <utility>
#include <vector>

#include "ryujin_hip_binding.h"
#include "ryujin_synth.h"

namespace mock
{
  /* SparsityPatternSIMD: columns(row) points at the first column index of the row, consecutive column indices
   * of a row are stride_of_row(row) apart (simd_length for the interleaved rows in the reference; here 2 for
   * every row, with junk in between, so that a binding that ignores the stride fails) */
  struct SparsityPattern {
    std::vector<uint64_t> start;
    std::vector<unsigned int> storage;
    unsigned int row_length(unsigned int i) const { return (unsigned int)((start[i + 1] - start[i]) / 2); }
    const unsigned int *columns(unsigned int i) const { return storage.data() + start[i]; }
    unsigned int stride_of_row(unsigned int) const { return 2; }
  };

  template <int n_comp>
  struct Matrix {
    const uint64_t *row_starts;
    const double *data;
    template <typename Number>
    Number get_entry(unsigned int i, unsigned int col_idx) const
    {
      return data[(row_starts[i] + col_idx) * n_comp];
    }
    template <typename Number>
    std::array<Number, n_comp> get_tensor(unsigned int i, unsigned int col_idx) const
    {
      std::array<Number, n_comp> t;
      for (int d = 0; d < n_comp; ++d)
        t[d] = data[(row_starts[i] + col_idx) * n_comp + d];
      return t;
    }
  };

  struct Vector {
    const double *data;
    double local_element(unsigned int i) const { return data[i]; }
  };

  struct Partitioner {
    std::vector<std::pair<unsigned int, unsigned int>> ghost, import, indices;
    const std::vector<std::pair<unsigned int, unsigned int>> &ghost_targets() const { return ghost; }
    const std::vector<std::pair<unsigned int, unsigned int>> &import_targets() const { return import; }
    const std::vector<std::pair<unsigned int, unsigned int>> &import_indices() const { return indices; }
  };

  struct Discretization {
    bool have_discontinuous_ansatz() const { return false; }
  };

  template <int dim>
  struct OfflineData {
    using Point = std::array<double, dim>;
    using BoundaryDescription = std::tuple<unsigned int, Point, double, double, unsigned char, Point>;
    using CouplingDescription = std::tuple<unsigned int, unsigned int, unsigned int>;

    const ryujin_hip_offline *o;
    SparsityPattern sparsity;
    std::shared_ptr<const Partitioner> partitioner;
    std::vector<BoundaryDescription> bmap;
    std::vector<CouplingDescription> pairs;
    Discretization discretization_;

    OfflineData(const ryujin_hip_offline *off, const double *b_positions)
        : o(off)
    {
      sparsity.start.assign(o->n_relevant + 1, 0);
      for (unsigned int i = 0; i < o->n_relevant; ++i)
        sparsity.start[i + 1] = sparsity.start[i] + 2 * (o->row_starts[i + 1] - o->row_starts[i]);
      sparsity.storage.assign(sparsity.start[o->n_relevant], 0xdeadbeefu);
      for (unsigned int i = 0; i < o->n_relevant; ++i)
        for (uint64_t e = o->row_starts[i]; e < o->row_starts[i + 1]; ++e)
          sparsity.storage[sparsity.start[i] + 2 * (e - o->row_starts[i])] = o->columns[e];
      auto p = std::make_shared<Partitioner>();
      for (int q = 0; q < o->n_nbr; ++q) {
        if (o->recv_off[q + 1] > o->recv_off[q])
          p->ghost.push_back({(unsigned int)o->nbr_rank[q], o->recv_off[q + 1] - o->recv_off[q]});
        if (o->send_off[q + 1] > o->send_off[q])
          p->import.push_back({(unsigned int)o->nbr_rank[q], o->send_off[q + 1] - o->send_off[q]});
      }
      /* import_indices: the flat send list compressed into half-open ranges, as dealii stores it */
      const uint32_t n_send = o->n_nbr ? o->send_off[o->n_nbr] : 0;
      for (uint32_t e = 0; e < n_send;) {
        uint32_t f = e + 1;
        while (f < n_send && o->send_idx[f] == o->send_idx[f - 1] + 1)
          ++f;
        p->indices.push_back({o->send_idx[e], o->send_idx[f - 1] + 1});
        e = f;
      }
      partitioner = p;
      for (uint32_t b = 0; b < o->n_bdry; ++b) {
        Point normal, position;
        for (int d = 0; d < dim; ++d) {
          normal[d] = o->b_normal[(size_t)b * dim + d];
          position[d] = b_positions[(size_t)b * dim + d];
        }
        bmap.emplace_back(o->b_i[b], normal, 0., 0., o->b_id[b], position);
      }
      for (uint32_t q = 0; q < o->n_pairs; ++q)
        pairs.emplace_back(o->p_i[q], o->p_col[q], o->p_j[q]);
    }

    const SparsityPattern &sparsity_pattern_simd() const { return sparsity; }
    Matrix<dim> cij_matrix() const { return {o->row_starts, o->cij}; }
    Matrix<1> mass_matrix() const { return {o->row_starts, o->mij}; }
    Matrix<1> incidence_matrix() const { return {o->row_starts, o->mij}; }
    Matrix<1> mass_matrix_inverse() const { return {o->row_starts, o->mij}; }
    Vector lumped_mass_matrix() const { return {o->mi}; }
    Vector lumped_mass_matrix_inverse() const { return {o->mi_inv}; }
    std::shared_ptr<const Partitioner> scalar_partitioner() const { return partitioner; }
    const Discretization &discretization() const { return discretization_; }
    unsigned int n_export_indices() const { return o->n_export; }
    unsigned int n_locally_internal() const { return o->n_internal; }
    unsigned int n_locally_owned() const { return o->n_owned; }
    unsigned int n_locally_relevant() const { return o->n_relevant; }
    double measure_of_omega() const { return o->measure_of_omega; }
    const std::vector<BoundaryDescription> &boundary_map() const { return bmap; }
    const std::vector<CouplingDescription> &coupling_boundary_pairs() const { return pairs; }
  };

  /* ParameterAcceptor stand-ins with the reference's accessor names */
  struct EulerView {
    double gamma() const { return 1.3; }
    double reference_density() const { return 2.; }
    double vacuum_state_relaxation_small() const { return 3.; }
    double vacuum_state_relaxation_large() const { return 4.; }
  };
  struct SwView {
    double gravity() const { return 9.; }
    double manning_friction_coefficient() const { return 0.1; }
    double reference_water_depth() const { return 2.; }
    double dry_state_relaxation_factor() const { return 0.3; }
    double dry_state_relaxation_small() const { return 5.; }
    double dry_state_relaxation_large() const { return 6.; }
  };
  struct Indicator {
    double evc_factor() const { return 0.5; }
  };
  struct Limiter {
    unsigned int iterations() const { return 1; }
    double newton_tolerance() const { return 1e-9; }
    unsigned int newton_max_iterations() const { return 3; }
    double relaxation_factor() const { return 2.; }
    bool limit_on_kinetic_energy() const { return true; }
    bool limit_on_square_velocity() const { return false; }
  };
  struct Riemann {
    double newton_tolerance() const { return 1e-8; }
    unsigned int newton_max_iterations() const { return 4; }
  };
} // namespace mock

template <typename T>
static bool same(const char *what, const T *a, const T *b, size_t n)
{
  if (n == 0 || std::memcmp(a, b, n * sizeof(T)) == 0)
    return true;
  std::fprintf(stderr, "MISMATCH %s\n", what);
  return false;
}

template <int dim>
static int run(int n_ranks, int cells)
{
  int failures = 0;
  for (int rank = 0; rank < n_ranks; ++rank) {
    ryujin_synth_spec spec{};
    spec.dim = dim;
    spec.n_cells[0] = 3 * cells;
    spec.n_cells[1] = cells;
    spec.n_cells[2] = dim == 3 ? cells : 1;
    spec.upper[0] = 3.;
    spec.upper[1] = spec.upper[2] = 1.;
    spec.bc[0] = RYUJIN_BC_DIRICHLET;
    spec.bc[1] = RYUJIN_BC_DO_NOTHING;
    for (int f = 2; f < 6; ++f)
      spec.bc[f] = RYUJIN_BC_SLIP;
    spec.cut_kind = RYUJIN_CUT_BOX; /* the forward-facing step: coupling boundary pairs along the corner */
    spec.cut_lo[0] = 0.6;
    spec.cut_lo[1] = spec.cut_lo[2] = -1.;
    spec.cut_hi[0] = 4.;
    spec.cut_hi[1] = 0.2;
    spec.cut_hi[2] = 2.;
    spec.cut_bc = RYUJIN_BC_SLIP;
    spec.n_ranks = n_ranks;
    spec.rank = rank;
    ryujin_synth *mesh = ryujin_synth_build(&spec);
    if (!mesh) {
      std::fprintf(stderr, "%s\n", ryujin_synth_last_error());
      return 1;
    }
    const ryujin_hip_offline *o = ryujin_synth_offline(mesh);
    const mock::OfflineData<dim> offline_data(o, ryujin_synth_bdry_positions(mesh));

    ryujin_hip_binding::OfflineArrays arrays;
    ryujin_hip_binding::fill_from_accessors<dim>(offline_data, arrays);
    const ryujin_hip_offline &f = arrays.offline;

    bool ok = f.n_export == o->n_export && f.n_internal == o->n_internal && f.n_owned == o->n_owned &&
              f.n_relevant == o->n_relevant && f.simd_length == 1 && f.measure_of_omega == o->measure_of_omega &&
              f.n_bdry == o->n_bdry && f.n_pairs == o->n_pairs && f.n_nbr == o->n_nbr &&
              f.discontinuous_ansatz == 0 && f.initial_precomputed == nullptr;
    if (!ok)
      std::fprintf(stderr, "MISMATCH scalars (rank %d)\n", rank);
    const size_t nnz = o->row_starts[o->n_relevant];
    ok = ok && same("row_starts", f.row_starts, o->row_starts, o->n_relevant + 1);
    ok = ok && same("columns", f.columns, o->columns, nnz);
    ok = ok && same("cij", f.cij, o->cij, nnz * dim);
    ok = ok && same("mij", f.mij, o->mij, nnz);
    ok = ok && same("mi", f.mi, o->mi, o->n_relevant);
    ok = ok && same("mi_inv", f.mi_inv, o->mi_inv, o->n_relevant);
    ok = ok && same("b_i", f.b_i, o->b_i, o->n_bdry);
    ok = ok && same("b_normal", f.b_normal, o->b_normal, (size_t)o->n_bdry * dim);
    ok = ok && same("b_id", f.b_id, o->b_id, o->n_bdry);
    ok = ok && same("b_positions", arrays.b_positions.data(), ryujin_synth_bdry_positions(mesh),
                    (size_t)o->n_bdry * dim);
    ok = ok && same("p_i", f.p_i, o->p_i, o->n_pairs) && same("p_col", f.p_col, o->p_col, o->n_pairs) &&
         same("p_j", f.p_j, o->p_j, o->n_pairs);
    if (o->n_nbr) {
      ok = ok && same("nbr_rank", f.nbr_rank, o->nbr_rank, o->n_nbr);
      ok = ok && same("send_off", f.send_off, o->send_off, o->n_nbr + 1);
      ok = ok && same("send_idx", f.send_idx, o->send_idx, o->send_off[o->n_nbr]);
      ok = ok && same("recv_off", f.recv_off, o->recv_off, o->n_nbr + 1);
      ok = ok && same("row_send_off", f.row_send_off, o->row_send_off, o->n_nbr + 1);
      ok = ok && same("row_send_row", f.row_send_row, o->row_send_row, o->row_send_off[o->n_nbr]);
      ok = ok && same("row_send_col", f.row_send_col, o->row_send_col, o->row_send_off[o->n_nbr]);
    }
    /* with a bathymetry */
    std::vector<double> Z(o->n_relevant);
    for (unsigned int i = 0; i < o->n_relevant; ++i)
      Z[i] = 0.25 * i;
    ryujin_hip_binding::fill_from_accessors<dim>(offline_data, arrays, Z.data(), 1);
    ok = ok && arrays.offline.initial_precomputed != nullptr &&
         same("initial_precomputed", arrays.offline.initial_precomputed, Z.data(), o->n_relevant);
    std::printf("rank %d of %d: n_owned %u n_relevant %u nnz %zu n_bdry %u n_pairs %u n_nbr %d row entries sent %u: %s\n",
                rank, n_ranks, o->n_owned, o->n_relevant, nnz, o->n_bdry, o->n_pairs, o->n_nbr,
                o->n_nbr ? o->row_send_off[o->n_nbr] : 0u, ok ? "identical" : "MISMATCH");
    failures += ok ? 0 : 1;
    ryujin_synth_free(mesh);
  }
  return failures;
}

int main(int argc, char **argv)
{
  const int dim = argc > 1 ? std::atoi(argv[1]) : 2;
  const int n_ranks = argc > 2 ? std::atoi(argv[2]) : 3;
  const int cells = argc > 3 ? std::atoi(argv[3]) : (dim == 2 ? 20 : 6);
  int failures = dim == 2 ? run<2>(n_ranks, cells) : run<3>(n_ranks, cells);

  /* parameters */
  ryujin_hip_params p{};
  ryujin_hip_binding::fill_params_euler(p, mock::EulerView());
  ryujin_hip_binding::fill_params_common(p, mock::Indicator(), mock::Limiter(), mock::Riemann());
  bool ok = p.gamma == 1.3 && p.reference_density == 2. && p.vacuum_state_relaxation_small == 3. &&
            p.vacuum_state_relaxation_large == 4. && p.indicator_evc_factor == 0.5 && p.limiter_iterations == 1 &&
            p.limiter_newton_tolerance == 1e-9 && p.limiter_newton_max_iterations == 3 &&
            p.limiter_relaxation_factor == 2. && p.riemann_newton_tolerance == 1e-8 &&
            p.riemann_newton_max_iterations == 4;
  ryujin_hip_params q{};
  ryujin_hip_binding::fill_params_shallow_water(q, mock::SwView(), mock::Indicator(), mock::Limiter());
  ok = ok && q.gravity == 9. && q.manning_friction_coefficient == 0.1 && q.reference_water_depth == 2. &&
       q.dry_state_relaxation_factor == 0.3 && q.dry_state_relaxation_small == 5. &&
       q.dry_state_relaxation_large == 6. && q.limiter_limit_on_kinetic_energy == 1 &&
       q.limiter_limit_on_square_velocity == 0 && q.limiter_iterations == 1;
  std::printf("parameters: %s\n", ok ? "identical" : "MISMATCH");
  failures += ok ? 0 : 1;

  ryujin_hip_binding::HandleCache cache; /* (allocating a twin needs a context, i.e. a GPU: not here) */
  failures += cache.size() == 0 ? 0 : 1;
  return failures;
}

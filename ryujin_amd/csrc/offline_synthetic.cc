// Synthetic OfflineData generator for structured Q1 meshes (host code, C ABI in
// include/ryujin_synth.h). See that header for the reference citations. Written
// from the closed-form element integrals, not from the
// reference's deal.II assembly.

#include "ryujin_synth.h"
#include "ryujin_exchange_lists.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace
{
  thread_local std::string g_error;

  constexpr uint32_t kInvalid = 0xFFFFFFFFu;

  struct Entry {
    uint32_t j;
    double m;
    double c[3];
  };

  struct BdryEntry {
    double normal[3];
    double boundary_mass;
    int id;
  };
} // namespace

struct ryujin_synth {
  ryujin_synth_spec spec{};
  int dim = 0;
  int64_t nc[3] = {1, 1, 1}; /* cells */
  int64_t nn[3] = {1, 1, 1}; /* nodes */
  double h[3] = {1, 1, 1};

  /* slab of node planes owned by this rank: [x0, x1) */
  int64_t x0 = 0, x1 = 0;
  /* local node box: planes [bx0, bx1) */
  int64_t bx0 = 0, bx1 = 0;

  std::vector<uint32_t> node_id; /* local box -> local index */

  ryujin_hip_offline off{};
  std::vector<uint64_t> row_starts;
  std::vector<uint32_t> columns;
  std::vector<double> cij, mij, mi, mi_inv;
  std::vector<uint32_t> b_i;
  std::vector<double> b_normal, b_pos;
  std::vector<uint8_t> b_id;
  std::vector<uint32_t> p_i, p_col, p_j;
  std::vector<int> nbr_rank;
  std::vector<uint32_t> send_off, send_idx, recv_off;
  std::vector<uint32_t> row_send_off, row_send_row, row_send_col;
  std::vector<double> positions;
  std::vector<uint64_t> global_ids;
  uint64_t n_global = 0;

  /* ---- geometry predicates --------------------------------------------- */

  bool cell_in_domain(int64_t cx, int64_t cy, int64_t cz) const
  {
    return cx >= 0 && cx < nc[0] && cy >= 0 && cy < nc[1] && cz >= 0 && cz < nc[2];
  }

  bool cell_active(int64_t cx, int64_t cy, int64_t cz) const
  {
    if (!cell_in_domain(cx, cy, cz))
      return false;
    if (spec.cut_kind == RYUJIN_CUT_NONE)
      return true;
    const double c[3] = {spec.lower[0] + (cx + 0.5) * h[0],
                         spec.lower[1] + (cy + 0.5) * h[1],
                         spec.lower[2] + (cz + 0.5) * h[2]};
    if (spec.cut_kind == RYUJIN_CUT_BOX) {
      bool inside = true;
      for (int d = 0; d < dim; ++d)
        inside = inside && c[d] > spec.cut_lo[d] && c[d] < spec.cut_hi[d];
      return !inside;
    }
    if (spec.cut_kind == RYUJIN_CUT_CYLINDER) {
      const double dx = c[0] - spec.cyl_center[0];
      const double dy = (dim >= 2) ? c[1] - spec.cyl_center[1] : 0.;
      return dx * dx + dy * dy > spec.cyl_radius * spec.cyl_radius;
    }
    return true;
  }

  /* node (ix,iy,iz) is touched by at least one active cell */
  bool node_active(int64_t ix, int64_t iy, int64_t iz) const
  {
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < (dim >= 2 ? 2 : 1); ++b)
        for (int c = 0; c < (dim >= 3 ? 2 : 1); ++c)
          if (cell_active(ix - 1 + a, dim >= 2 ? iy - 1 + b : 0, dim >= 3 ? iz - 1 + c : 0))
            return true;
    return false;
  }

  /* node of plane `ix` touched by an active cell of cell column `cx` */
  bool node_touched_from_column(int64_t cx, int64_t iy, int64_t iz) const
  {
    for (int b = 0; b < (dim >= 2 ? 2 : 1); ++b)
      for (int c = 0; c < (dim >= 3 ? 2 : 1); ++c)
        if (cell_active(cx, dim >= 2 ? iy - 1 + b : 0, dim >= 3 ? iz - 1 + c : 0))
          return true;
    return false;
  }

  size_t box_index(int64_t ix, int64_t iy, int64_t iz) const
  {
    return (size_t)(((iz * nn[1]) + iy) * (bx1 - bx0) + (ix - bx0));
  }

  uint32_t local_id(int64_t ix, int64_t iy, int64_t iz) const
  {
    if (ix < bx0 || ix >= bx1 || iy < 0 || iy >= nn[1] || iz < 0 || iz >= nn[2])
      return kInvalid;
    return node_id[box_index(ix, iy, iz)];
  }

  /* x-slab partition of the node planes. Without a cut-out every plane carries the same number of
   * gridpoints and the planes are dealt out evenly; with one (forward-facing step, cylinder) the planes
   * are weighted by the active cells next to them, so that all ranks own (nearly) the same number of
   * gridpoints -- what a graph partitioner would give the reference -- instead of the same length of
   * channel. Deterministic: every rank computes the same boundaries. */
  void slab(int n_ranks, int rank, int64_t &a, int64_t &b) const
  {
    const int64_t n_planes = nn[0];
    if (spec.cut_kind == RYUJIN_CUT_NONE || n_ranks == 1) {
      a = n_planes * rank / n_ranks;
      b = n_planes * (rank + 1) / n_ranks;
      return;
    }
    std::vector<double> cells((size_t)nc[0], 0.);
    for (int64_t cx = 0; cx < nc[0]; ++cx) {
      int64_t n = 0;
      for (int64_t cz = 0; cz < nc[2]; ++cz)
        for (int64_t cy = 0; cy < nc[1]; ++cy)
          n += cell_active(cx, cy, cz) ? 1 : 0;
      cells[(size_t)cx] = (double)n;
    }
    std::vector<double> prefix((size_t)n_planes + 1, 0.); /* weight of the planes [0, ix) */
    for (int64_t ix = 0; ix < n_planes; ++ix) {
      const double left = ix > 0 ? cells[(size_t)ix - 1] : 0., right = ix < nc[0] ? cells[(size_t)ix] : 0.;
      prefix[(size_t)ix + 1] = prefix[(size_t)ix] + 0.5 * (left + right);
    }
    const double total = prefix[(size_t)n_planes];
    auto boundary = [&](int r) -> int64_t {
      if (r <= 0)
        return 0;
      if (r >= n_ranks)
        return n_planes;
      const double target = total * (double)r / (double)n_ranks;
      int64_t ix = std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin();
      /* every rank keeps at least two planes */
      ix = std::max<int64_t>(ix, 2 * (int64_t)r);
      ix = std::min<int64_t>(ix, n_planes - 2 * (int64_t)(n_ranks - r));
      return ix;
    };
    a = boundary(rank);
    b = boundary(rank + 1);
  }

  bool build();
};

bool ryujin_synth::build()
{
  dim = spec.dim;
  if (dim < 1 || dim > 3) {
    g_error = "dim must be 1, 2 or 3";
    return false;
  }
  for (int d = 0; d < 3; ++d) {
    nc[d] = d < dim ? (int64_t)spec.n_cells[d] : 1;
    nn[d] = d < dim ? nc[d] + 1 : 1;
    h[d] = d < dim ? (spec.upper[d] - spec.lower[d]) / (double)nc[d] : 1.;
    if (d < dim && (nc[d] < 1 || !(h[d] > 0.))) {
      g_error = "invalid mesh extents";
      return false;
    }
  }
  const int R = std::max(1, spec.n_ranks);
  const int r = spec.rank;
  if (r < 0 || r >= R) {
    g_error = "rank out of range";
    return false;
  }
  slab(R, r, x0, x1);
  if (R > 1 && x1 - x0 < 2) {
    g_error = "slab partition needs at least two node planes per rank";
    return false;
  }
  bx0 = std::max<int64_t>(0, x0 - 1);
  bx1 = std::min<int64_t>(nn[0], x1 + 1);

  const bool have_left = r > 0;
  const bool have_right = r + 1 < R;

  /* ---- local numbering --------------------------------------------------
   * [left export layer][right export layer][remaining owned, x fastest]
   * [ghosts of rank r-1][ghosts of rank r+1]; ghost order = owner's order. */

  node_id.assign((size_t)((bx1 - bx0) * nn[1] * nn[2]), kInvalid);
  uint32_t next = 0;

  auto number_plane_from_column = [&](int64_t ix, int64_t cx) {
    for (int64_t iz = 0; iz < nn[2]; ++iz)
      for (int64_t iy = 0; iy < nn[1]; ++iy)
        if (node_touched_from_column(cx, iy, iz))
          node_id[box_index(ix, iy, iz)] = next++;
  };

  uint32_t n_export_left = 0, n_export_right = 0;
  if (have_left) {
    number_plane_from_column(x0, x0 - 1);
    n_export_left = next;
  }
  if (have_right) {
    number_plane_from_column(x1 - 1, x1 - 1);
    n_export_right = next - n_export_left;
  }
  const uint32_t n_export = next;

  /* Remaining owned nodes: lexicographic (x fastest), or -- experiment knob of the measurement scripts,
   * RYUJIN_SYNTH_TILE=tx,ty,tz -- tile by tile, lexicographic inside a tile: the library is numbering agnostic, a
   * numbering whose stencil neighbours stay within a few dozen slices keeps the per-node gathers in L2. */
  int64_t tile[3] = {nn[0], 1, 1};
  bool tiled = false;
  if (const char *e = std::getenv("RYUJIN_SYNTH_TILE")) {
    long long a = 0, b = 0, c = 0;
    if (std::sscanf(e, "%lld,%lld,%lld", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) {
      tile[0] = a;
      tile[1] = b;
      tile[2] = c;
      tiled = true;
    }
  }
  if (!tiled) {
    for (int64_t iz = 0; iz < nn[2]; ++iz)
      for (int64_t iy = 0; iy < nn[1]; ++iy)
        for (int64_t ix = x0; ix < x1; ++ix) {
          const size_t b = box_index(ix, iy, iz);
          if (node_id[b] == kInvalid && node_active(ix, iy, iz))
            node_id[b] = next++;
        }
  } else {
    for (int64_t tz = 0; tz < nn[2]; tz += tile[2])
      for (int64_t ty = 0; ty < nn[1]; ty += tile[1])
        for (int64_t tx = x0; tx < x1; tx += tile[0])
          for (int64_t iz = tz; iz < std::min(nn[2], tz + tile[2]); ++iz)
            for (int64_t iy = ty; iy < std::min(nn[1], ty + tile[1]); ++iy)
              for (int64_t ix = tx; ix < std::min(x1, tx + tile[0]); ++ix) {
                const size_t b = box_index(ix, iy, iz);
                if (node_id[b] == kInvalid && node_active(ix, iy, iz))
                  node_id[b] = next++;
              }
  }
  const uint32_t n_owned = next;

  uint32_t n_ghost_left = 0, n_ghost_right = 0;
  if (have_left) {
    number_plane_from_column(x0 - 1, x0 - 1);
    n_ghost_left = next - n_owned;
  }
  if (have_right) {
    number_plane_from_column(x1, x1 - 1);
    n_ghost_right = next - n_owned - n_ghost_left;
  }
  const uint32_t n_relevant = next;

  /* ---- positions / global ids ----------------------------------------- */

  positions.assign((size_t)n_relevant * dim, 0.);
  global_ids.assign(n_relevant, 0);
  std::vector<std::array<int32_t, 3>> grid_index(n_relevant);
  for (int64_t iz = 0; iz < nn[2]; ++iz)
    for (int64_t iy = 0; iy < nn[1]; ++iy)
      for (int64_t ix = bx0; ix < bx1; ++ix) {
        const uint32_t i = node_id[box_index(ix, iy, iz)];
        if (i == kInvalid)
          continue;
        const int64_t idx[3] = {ix, iy, iz};
        for (int d = 0; d < dim; ++d)
          positions[(size_t)i * dim + d] = spec.lower[d] + idx[d] * h[d];
        global_ids[i] = (uint64_t)((iz * nn[1] + iy) * nn[0] + ix);
        grid_index[i] = {(int32_t)ix, (int32_t)iy, (int32_t)iz};
      }

  /* ---- element matrices -------------------------------------------------
   * 1-D on [0,h]: M[a][b] = int phi_a phi_b = h/3 (a==b), h/6;
   *               D[a][b] = int phi_a phi_b' = -1/2 (b==0), +1/2 (b==1). */
  double M[3][2][2], D[2][2];
  for (int d = 0; d < 3; ++d)
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        M[d][a][b] = d < dim ? (a == b ? h[d] / 3. : h[d] / 6.) : 1.;
  for (int a = 0; a < 2; ++a) {
    D[a][0] = -0.5;
    D[a][1] = 0.5;
  }

  /* ---- rows ------------------------------------------------------------- */

  row_starts.assign((size_t)n_relevant + 1, 0);
  columns.clear();
  mij.clear();
  cij.clear();
  mi.assign(n_relevant, 0.);
  mi_inv.assign(n_relevant, 0.);

  const int ny_off = dim >= 2 ? 1 : 0, nz_off = dim >= 3 ? 1 : 0;

  /* rows are assembled in contiguous chunks, one per thread, and concatenated in order afterwards */
  int n_chunks = 1;
#ifdef _OPENMP
  n_chunks = std::min(16, std::max(1, omp_get_max_threads()));
#endif
  struct Chunk {
    std::vector<uint32_t> cols, lengths;
    std::vector<double> m, c;
  };
  std::vector<Chunk> chunks((size_t)n_chunks);
#pragma omp parallel for schedule(static, 1) num_threads(n_chunks)
  for (int ch = 0; ch < n_chunks; ++ch) {
  Chunk &out = chunks[(size_t)ch];
  const uint32_t row_begin = (uint32_t)((uint64_t)n_relevant * ch / n_chunks);
  const uint32_t row_end = (uint32_t)((uint64_t)n_relevant * (ch + 1) / n_chunks);
  std::vector<Entry> row;
  for (uint32_t i = row_begin; i < row_end; ++i) {
    const auto gi = grid_index[i];
    const bool ghost_row = i >= n_owned;
    row.clear();
    for (int dz = -nz_off; dz <= nz_off; ++dz)
      for (int dy = -ny_off; dy <= ny_off; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          const int off[3] = {dx, dy, dz};
          const uint32_t j = local_id(gi[0] + dx, gi[1] + dy, gi[2] + dz);
          if (j == kInvalid)
            continue;
          /* ghost rows only keep the diagonal and owned columns */
          if (ghost_row && j != i && j >= n_owned)
            continue;
          Entry e{j, 0., {0., 0., 0.}};
          bool shared = false;
          /* cells adjacent to both nodes */
          int lo[3], hi[3];
          for (int d = 0; d < 3; ++d) {
            if (d >= dim) {
              lo[d] = hi[d] = 0;
            } else if (off[d] == 0) {
              lo[d] = gi[d] - 1;
              hi[d] = gi[d];
            } else if (off[d] > 0) {
              lo[d] = hi[d] = gi[d];
            } else {
              lo[d] = hi[d] = gi[d] - 1;
            }
          }
          for (int cz = lo[2]; cz <= hi[2]; ++cz)
            for (int cy = lo[1]; cy <= hi[1]; ++cy)
              for (int cx = lo[0]; cx <= hi[0]; ++cx) {
                if (!cell_active(cx, cy, cz))
                  continue;
                shared = true;
                const int cc[3] = {cx, cy, cz};
                int a[3], b[3];
                for (int d = 0; d < 3; ++d) {
                  a[d] = d < dim ? gi[d] - cc[d] : 0;
                  b[d] = d < dim ? a[d] + off[d] : 0;
                }
                double m = 1.;
                for (int d = 0; d < dim; ++d)
                  m *= M[d][a[d]][b[d]];
                e.m += m;
                for (int d = 0; d < dim; ++d) {
                  double c = D[a[d]][b[d]];
                  for (int q = 0; q < dim; ++q)
                    if (q != d)
                      c *= M[q][a[q]][b[q]];
                  e.c[d] += c;
                }
              }
          if (shared)
            row.push_back(e);
        }
    /* diagonal first, then ascending local index */
    std::sort(row.begin(), row.end(), [i](const Entry &x, const Entry &y) {
      if ((x.j == i) != (y.j == i))
        return x.j == i;
      return x.j < y.j;
    });
    double mass = 0.;
    for (const auto &e : row) {
      out.cols.push_back(e.j);
      out.m.push_back(e.m);
      for (int d = 0; d < dim; ++d)
        out.c.push_back(e.c[d]);
      mass += e.m;
    }
    out.lengths.push_back((uint32_t)row.size());
    if (!ghost_row) {
      mi[i] = mass;
      mi_inv[i] = 1. / mass;
    }
  }
  }
  {
    size_t nnz = 0;
    for (const auto &ch : chunks)
      nnz += ch.cols.size();
    columns.reserve(nnz);
    mij.reserve(nnz);
    cij.reserve(nnz * dim);
    uint32_t i = 0;
    for (auto &ch : chunks) {
      columns.insert(columns.end(), ch.cols.begin(), ch.cols.end());
      mij.insert(mij.end(), ch.m.begin(), ch.m.end());
      cij.insert(cij.end(), ch.c.begin(), ch.c.end());
      for (const uint32_t len : ch.lengths) {
        row_starts[i + 1] = row_starts[i] + len;
        ++i;
      }
      ch = Chunk{}; /* release the chunk: peak memory stays at the final arrays plus one chunk */
    }
  }

  /* lumped mass of ghosts: the full row sum (computed as the owner does) */
  for (uint32_t i = n_owned; i < n_relevant; ++i) {
    const auto gi = grid_index[i];
    double mass = 0.;
    /* sum over all active cells touching the node: m_i = sum_cells prod_d h_d/2 */
    for (int c = 0; c <= nz_off; ++c)
      for (int b = 0; b <= ny_off; ++b)
        for (int a = 0; a <= 1; ++a)
          if (cell_active(gi[0] - 1 + a, dim >= 2 ? gi[1] - 1 + b : 0,
                          dim >= 3 ? gi[2] - 1 + c : 0)) {
            double v = 1.;
            for (int d = 0; d < dim; ++d)
              v *= 0.5 * h[d];
            mass += v;
          }
    mi[i] = mass;
    mi_inv[i] = 1. / mass;
  }

  /* ---- |Omega| and global size ------------------------------------------ */

  {
    uint64_t n_active_cells = 0;
    for (int64_t cz = 0; cz < nc[2]; ++cz)
      for (int64_t cy = 0; cy < nc[1]; ++cy)
        for (int64_t cx = 0; cx < nc[0]; ++cx)
          n_active_cells += cell_active(cx, cy, cz) ? 1 : 0;
    double vol = 1.;
    for (int d = 0; d < dim; ++d)
      vol *= h[d];
    off.measure_of_omega = vol * (double)n_active_cells;

    if (spec.cut_kind == RYUJIN_CUT_NONE) {
      n_global = (uint64_t)(nn[0] * nn[1] * nn[2]);
    } else {
      n_global = 0;
      for (int64_t iz = 0; iz < nn[2]; ++iz)
        for (int64_t iy = 0; iy < nn[1]; ++iy)
          for (int64_t ix = 0; ix < nn[0]; ++ix)
            n_global += node_active(ix, iy, iz) ? 1 : 0;
    }
  }

  /* ---- boundary map -------------------------------------------------------
   * Face contributions collected cell by cell (x fastest) and face by face
   * (-x,+x,-y,+y,-z,+z), then merged per node exactly as the reference's
   * filter does (offline_data.template.h:1292-1336). */

  std::multimap<uint32_t, BdryEntry> preliminary;
  std::vector<char> is_bdry(n_relevant, 0);
  {
    const int64_t cx_lo = std::max<int64_t>(0, bx0 - 1), cx_hi = std::min<int64_t>(nc[0], bx1);
    for (int64_t cz = 0; cz < nc[2]; ++cz)
      for (int64_t cy = 0; cy < nc[1]; ++cy)
        for (int64_t cx = cx_lo; cx < cx_hi; ++cx) {
          if (!cell_active(cx, cy, cz))
            continue;
          const int64_t cc[3] = {cx, cy, cz};
          for (int f = 0; f < 2 * dim; ++f) {
            const int fd = f / 2, side = f % 2;
            int64_t nb[3] = {cx, cy, cz};
            nb[fd] += side ? 1 : -1;
            if (cell_active(nb[0], nb[1], nb[2]))
              continue;
            const int id =
                cell_in_domain(nb[0], nb[1], nb[2]) ? spec.cut_bc : spec.bc[f];
            if (id == RYUJIN_BC_PERIODIC)
              continue;
            double area = 1.;
            int n_face_nodes = 1;
            for (int d = 0; d < dim; ++d)
              if (d != fd) {
                area *= h[d];
                n_face_nodes *= 2;
              }
            const double bmass = area / n_face_nodes;
            /* nodes of the face */
            for (int t = 0; t < n_face_nodes; ++t) {
              int64_t idx[3] = {0, 0, 0};
              int bit = 0;
              for (int d = 0; d < dim; ++d) {
                if (d == fd)
                  idx[d] = cc[d] + side;
                else
                  idx[d] = cc[d] + ((t >> bit++) & 1);
              }
              const uint32_t i = local_id(idx[0], idx[1], idx[2]);
              if (i == kInvalid)
                continue;
              is_bdry[i] = 1;
              if (i >= n_owned)
                continue;
              BdryEntry be{};
              be.normal[fd] = (side ? 1. : -1.) * bmass;
              be.boundary_mass = bmass;
              be.id = id;
              preliminary.insert({i, be});
            }
          }
        }
  }

  std::multimap<uint32_t, BdryEntry> filtered;
  for (auto entry : preliminary) {
    bool inserted = false;
    auto range = filtered.equal_range(entry.first);
    for (auto it = range.first; it != range.second; ++it) {
      BdryEntry &nw = entry.second;
      BdryEntry &old = it->second;
      if (old.id != nw.id)
        continue;
      double dot = 0., n1 = 0., n2 = 0.;
      for (int d = 0; d < dim; ++d) {
        dot += old.normal[d] * nw.normal[d];
        n1 += old.normal[d] * old.normal[d];
        n2 += nw.normal[d] * nw.normal[d];
      }
      if (dot / std::sqrt(n1) / std::sqrt(n2) > 0.50) {
        for (int d = 0; d < dim; ++d)
          old.normal[d] += nw.normal[d];
        old.boundary_mass += nw.boundary_mass;
        inserted = true;
        continue;
      } else if (dim == 2) {
        if (nw.id == RYUJIN_BC_SLIP) {
          nw.id = RYUJIN_BC_NO_SLIP;
          old.id = RYUJIN_BC_NO_SLIP;
        }
      }
    }
    if (!inserted)
      filtered.insert(entry);
  }

  b_i.clear();
  b_normal.clear();
  b_id.clear();
  b_pos.clear();
  for (const auto &it : filtered) {
    double norm = 0.;
    for (int d = 0; d < dim; ++d)
      norm += it.second.normal[d] * it.second.normal[d];
    const double normal_mass = std::sqrt(norm) + std::numeric_limits<double>::epsilon();
    b_i.push_back(it.first);
    for (int d = 0; d < dim; ++d) {
      b_normal.push_back(it.second.normal[d] / normal_mass);
      b_pos.push_back(positions[(size_t)it.first * dim + d]);
    }
    b_id.push_back((uint8_t)it.second.id);
  }

  /* ---- coupling boundary pairs ------------------------------------------ */

  p_i.clear();
  p_col.clear();
  p_j.clear();
  for (uint32_t i = 0; i < n_owned; ++i) {
    if (!is_bdry[i])
      continue;
    const uint64_t rs = row_starts[i], re = row_starts[i + 1];
    if (re - rs == 1)
      continue;
    for (uint64_t e = rs + 1; e < re; ++e) {
      const uint32_t j = columns[e];
      if (is_bdry[j]) {
        p_i.push_back(i);
        p_col.push_back((uint32_t)(e - rs));
        p_j.push_back(j);
      }
    }
  }

  /* ---- exchange pattern --------------------------------------------------- */

  nbr_rank.clear();
  send_off.assign(1, 0);
  send_idx.clear();
  recv_off.assign(1, n_owned);
  row_send_off.assign(1, 0);
  row_send_row.clear();
  row_send_col.clear();

  auto add_neighbour = [&](int rank_p, uint32_t exp_begin, uint32_t exp_end,
                           uint32_t ghost_begin, uint32_t ghost_end) {
    nbr_rank.push_back(rank_p);
    for (uint32_t i = exp_begin; i < exp_end; ++i)
      send_idx.push_back(i);
    send_off.push_back((uint32_t)send_idx.size());
    recv_off.push_back(ghost_end);
    /* matrix rows: diagonal + entries whose column is a ghost owned by p -- the one statement of the rule of
     * sparse_matrix_simd.template.h:196-264 (include/ryujin_exchange_lists.h) */
    const std::vector<uint32_t> exported(send_idx.end() - (exp_end - exp_begin), send_idx.end());
    const size_t n_entries = ryujin_ghost_row_send_entries(row_starts.data(), columns.data(), exported.data(),
                                                           exported.size(), ghost_begin, ghost_end, nullptr,
                                                           nullptr);
    const size_t first = row_send_row.size();
    row_send_row.resize(first + n_entries);
    row_send_col.resize(first + n_entries);
    ryujin_ghost_row_send_entries(row_starts.data(), columns.data(), exported.data(), exported.size(),
                                  ghost_begin, ghost_end, row_send_row.data() + first,
                                  row_send_col.data() + first);
    row_send_off.push_back((uint32_t)row_send_row.size());
  };

  if (have_left)
    add_neighbour(r - 1, 0, n_export_left, n_owned, n_owned + n_ghost_left);
  if (have_right)
    add_neighbour(r + 1, n_export_left, n_export_left + n_export_right,
                  n_owned + n_ghost_left, n_owned + n_ghost_left + n_ghost_right);

  /* ---- publish ------------------------------------------------------------ */

  off.n_export = n_export;
  off.n_internal = n_owned; /* simd_length 1: every owned row is "internal" */
  off.n_owned = n_owned;
  off.n_relevant = n_relevant;
  off.simd_length = 1;
  off.row_starts = row_starts.data();
  off.columns = columns.data();
  off.cij = cij.data();
  off.mij = mij.data();
  off.mi = mi.data();
  off.mi_inv = mi_inv.data();
  off.n_bdry = (uint32_t)b_i.size();
  off.b_i = b_i.data();
  off.b_normal = b_normal.data();
  off.b_id = b_id.data();
  off.n_pairs = (uint32_t)p_i.size();
  off.p_i = p_i.data();
  off.p_col = p_col.data();
  off.p_j = p_j.data();
  off.initial_precomputed = nullptr;
  off.n_nbr = (int)nbr_rank.size();
  off.nbr_rank = nbr_rank.data();
  off.send_off = send_off.data();
  off.send_idx = send_idx.data();
  off.recv_off = recv_off.data();
  off.row_send_off = row_send_off.data();
  off.row_send_row = row_send_row.data();
  off.row_send_col = row_send_col.data();
  return true;
}

extern "C" {

ryujin_synth *ryujin_synth_build(const ryujin_synth_spec *spec)
{
  if (!spec) {
    g_error = "null spec";
    return nullptr;
  }
  auto *s = new ryujin_synth;
  s->spec = *spec;
  try {
    if (!s->build()) {
      delete s;
      return nullptr;
    }
  } catch (const std::exception &e) {
    g_error = e.what();
    delete s;
    return nullptr;
  }
  return s;
}

void ryujin_synth_free(ryujin_synth *s)
{
  delete s;
}

const char *ryujin_synth_last_error(void)
{
  return g_error.c_str();
}

const ryujin_hip_offline *ryujin_synth_offline(const ryujin_synth *s)
{
  return &s->off;
}

uint64_t ryujin_synth_nnz(const ryujin_synth *s)
{
  return s->columns.size();
}

uint64_t ryujin_synth_n_global(const ryujin_synth *s)
{
  return s->n_global;
}

const double *ryujin_synth_positions(const ryujin_synth *s)
{
  return s->positions.data();
}

const uint64_t *ryujin_synth_global_ids(const ryujin_synth *s)
{
  return s->global_ids.data();
}

const double *ryujin_synth_bdry_positions(const ryujin_synth *s)
{
  return s->b_pos.data();
}

size_t ryujin_synth_ghost_row_send_entries(const uint64_t *row_starts, const uint32_t *columns,
                                           const uint32_t *exported_rows, size_t n_exported,
                                           uint32_t ghost_begin, uint32_t ghost_end, uint32_t *out_row,
                                           uint32_t *out_col)
{
  return ryujin_ghost_row_send_entries(row_starts, columns, exported_rows, n_exported, ghost_begin, ghost_end,
                                       out_row, out_col);
}

} /* extern "C" */

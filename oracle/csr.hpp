// ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Import of the reference's SparsityPatternSIMD storage scheme into a plain CSR with
// diagonal-first rows (the logical (row, col_idx) view the sweeps use), and the
// transposed-position table. Restates source/sparse_matrix_simd.h:311-350,403-418 and
// source/sparse_matrix_simd.template.h:96-127.

#pragma once

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "ryujin_hip.h"

namespace oracle
{
  struct RefLayout {
    uint32_t n_internal, n_relevant, sl;
    const uint64_t *row_starts;

    explicit RefLayout(const ryujin_hip_offline &o)
        : n_internal(o.n_internal)
        , n_relevant(o.n_relevant)
        , sl(o.simd_length ? o.simd_length : 1)
        , row_starts(o.row_starts)
    {
    }

    /* sparse_matrix_simd.h:340-350 */
    uint32_t row_length(uint32_t row) const
    {
      if (row < n_internal) {
        const uint32_t g = row / sl;
        return (uint32_t)((row_starts[g + 1] - row_starts[g]) / sl);
      }
      return (uint32_t)(row_starts[row + 1] - row_starts[row]);
    }

    /* position of column index / scalar entry; sparse_matrix_simd.h:326-335 */
    uint64_t col_pos(uint32_t row, uint32_t col_idx) const
    {
      if (row < n_internal)
        return row_starts[row / sl] + (uint64_t)col_idx * sl + row % sl;
      return row_starts[row] + col_idx;
    }

    /* flat data index of component d of entry (row,col_idx); sparse_matrix_simd.h:403-418 */
    uint64_t data_pos(uint32_t row, uint32_t col_idx, uint32_t n_comp, uint32_t d) const
    {
      if (row < n_internal)
        return (row_starts[row / sl] + (uint64_t)col_idx * sl) * n_comp + (uint64_t)d * sl + row % sl;
      return (row_starts[row] + col_idx) * n_comp + d;
    }
  };

  struct CSR {
    uint32_t n_rows = 0;
    std::vector<uint64_t> ptr;      /* [n_rows+1] */
    std::vector<uint32_t> col;      /* [nnz], col 0 of each row = diagonal */
    std::vector<uint64_t> transpose; /* [nnz] position of (j,i) or ~0 if absent */

    uint64_t nnz() const { return col.size(); }

    void import(const ryujin_hip_offline &o)
    {
      const RefLayout ref(o);
      n_rows = o.n_relevant;
      ptr.assign((size_t)n_rows + 1, 0);
      for (uint32_t i = 0; i < n_rows; ++i)
        ptr[i + 1] = ptr[i] + ref.row_length(i);
      col.resize(ptr[n_rows]);
      bool diagonal_first = true;
#pragma omp parallel for schedule(static) reduction(&& : diagonal_first)
      for (uint32_t i = 0; i < n_rows; ++i) {
        const uint32_t len = ref.row_length(i);
        for (uint32_t c = 0; c < len; ++c)
          col[ptr[i] + c] = o.columns[ref.col_pos(i, c)];
        if (len > 0 && col[ptr[i]] != i)
          diagonal_first = false;
      }
      if (!diagonal_first)
        throw std::runtime_error("row does not start with its diagonal");
      /* transposed positions: rows are (diag, ascending...) */
      transpose.assign(col.size(), ~uint64_t(0));
#pragma omp parallel for schedule(static)
      for (uint32_t i = 0; i < n_rows; ++i)
        for (uint64_t e = ptr[i]; e < ptr[i + 1]; ++e) {
          const uint32_t j = col[e];
          if (j == i) {
            transpose[e] = ptr[i];
            continue;
          }
          if (j >= n_rows)
            continue;
          const auto b = col.begin() + ptr[j] + 1, en = col.begin() + ptr[j + 1];
          const auto it = std::lower_bound(b, en, i);
          if (it != en && *it == i)
            transpose[e] = (uint64_t)(it - col.begin());
        }
    }

    /* gather an n_comp matrix from the reference layout into CSR order (AoS per entry) */
    std::vector<double> gather(const ryujin_hip_offline &o, const double *data, uint32_t n_comp) const
    {
      const RefLayout ref(o);
      std::vector<double> out((size_t)nnz() * n_comp);
#pragma omp parallel for schedule(static)
      for (uint32_t i = 0; i < n_rows; ++i)
        for (uint64_t e = ptr[i]; e < ptr[i + 1]; ++e)
          for (uint32_t d = 0; d < n_comp; ++d)
            out[e * n_comp + d] = data[ref.data_pos(i, (uint32_t)(e - ptr[i]), n_comp, d)];
      return out;
    }
  };
} // namespace oracle

/*
 * ryujin_exchange_lists.h -- the ghost-row send-list rule of SparsityPatternSIMD, stated ONCE.
 *
 * The reference's SparseMatrixSIMD::update_ghost_rows (source/sparse_matrix_simd.h:649-763) sends, to every
 * rank that holds some of our owned rows as ghost rows, the entries listed in
 * SparsityPatternSIMD::entries_to_be_sent, built in source/sparse_matrix_simd.template.h:196-264:
 *
 *   for every row exported to rank p, in export order: the diagonal, and then every entry (row, col_idx)
 *   whose column is a ghost DoF RECEIVED from the same rank p  (:249-261)
 *
 * -- a ghost row only holds the diagonal and the transposes of locally owned entries (:61-74), so this is
 * exactly what the receiver stores, in its storage order. A rank that sends us nothing receives nothing
 * either (:229-247, the unmatched import target).
 *
 * ryujin_hip_offline::row_send_row / row_send_col (include/ryujin_hip.h) are these lists. Everything that
 * builds them uses THIS function: the synthetic generator (ryujin_amd/csrc/offline_synthetic.cc), the test
 * partitioner (tests/helpers_unstructured.py through ryujin_synth_ghost_row_send_entries) and the deal.II-side
 * adapter / exporter (contrib/ryujin_offline_fill.h). It is pinned against the reference's own 4-rank
 * baseline tests/common/sparsity_pattern_simd_01.mpirun=4.output in tests/test_send_lists_golden.py.
 *
 * Header-only C (static inline), no dependencies.
 */
#ifndef RYUJIN_EXCHANGE_LISTS_H
#define RYUJIN_EXCHANGE_LISTS_H

#include <stddef.h>
#include <stdint.h>

/*
 * row_starts/columns: the rank's stencil as plain diagonal-first CSR in local numbering (entry (row, c) at
 *   columns[row_starts[row] + c]; simd_length == 1 storage of ryujin_hip_offline).
 * exported_rows[n_exported]: the owned rows rank p holds as ghosts, in export order (the partitioner's
 *   import_indices for p).
 * [ghost_begin, ghost_end): local index range of the ghost DoFs received from p; pass ghost_begin ==
 *   ghost_end if p sends us nothing -- then nothing is sent to it.
 * out_row/out_col: receive the (row, col_idx) pairs; may both be NULL to count only.
 * Returns the number of entries.
 */
static inline size_t ryujin_ghost_row_send_entries(const uint64_t *row_starts, const uint32_t *columns,
                                                   const uint32_t *exported_rows, size_t n_exported,
                                                   uint32_t ghost_begin, uint32_t ghost_end,
                                                   uint32_t *out_row, uint32_t *out_col)
{
  size_t n = 0;
  size_t q;
  uint64_t e;
  if (ghost_begin >= ghost_end)
    return 0;
  for (q = 0; q < n_exported; ++q) {
    const uint32_t row = exported_rows[q];
    if (out_row) {
      out_row[n] = row;
      out_col[n] = 0;
    }
    ++n;
    for (e = row_starts[row] + 1; e < row_starts[row + 1]; ++e)
      if (columns[e] >= ghost_begin && columns[e] < ghost_end) {
        if (out_row) {
          out_row[n] = row;
          out_col[n] = (uint32_t)(e - row_starts[row]);
        }
        ++n;
      }
  }
  return n;
}

#endif /* RYUJIN_EXCHANGE_LISTS_H */

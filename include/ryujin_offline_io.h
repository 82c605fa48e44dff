/*
 * ryujin_offline_io.h -- wire format for OfflineData dumps.
 *
 * The hot path's input contract is `ryujin_hip_offline` (ryujin_hip.h): the flat arrays behind the
 * accessors of the reference's OfflineData (source/offline_data.h:121-264). This file format is exactly
 * those arrays, written by a small exporter on the deal.II side (INTEGRATION.md shows the ~40 lines that
 * fill the struct from an `OfflineData<dim>` and call ryujin_offline_write) and read here, so that meshes
 * the synthetic generator cannot produce (curved boundaries: source/geometry_step.h:93-127 rounded
 * corner, source/geometry_cylinder.h; unstructured or locally refined triangulations) run unchanged.
 *
 * Layout (little endian, every section 8-byte aligned):
 *   char     magic[8] = "RYJOFFL1"
 *   uint32   version (1), dim, n_initial_precomputed, flags (bit 0: positions, bit 1: boundary positions,
 *            bit 2: discontinuous ansatz)
 *   uint32   n_export, n_internal, n_owned, n_relevant, simd_length, n_bdry, n_pairs, n_nbr
 *   uint64   nnz (== row_starts[n_relevant])
 *   float64  measure_of_omega
 *   sections, each `uint64 n_bytes` followed by the payload padded to 8 bytes, in this order:
 *     row_starts u64[n_relevant+1], columns u32[nnz], cij f64[nnz*dim], mij f64[nnz], mi f64[n_relevant],
 *     mi_inv f64[n_relevant], b_i u32[n_bdry], b_normal f64[n_bdry*dim], b_id u8[n_bdry],
 *     p_i, p_col, p_j u32[n_pairs], initial_precomputed f64[n_relevant*n_initial_precomputed],
 *     nbr_rank i32[n_nbr], send_off u32[n_nbr+1], send_idx u32[send_off[n_nbr]], recv_off u32[n_nbr+1],
 *     row_send_off u32[n_nbr+1], row_send_row, row_send_col u32[row_send_off[n_nbr]],
 *     positions f64[n_relevant*dim] (flag bit 0), b_positions f64[n_bdry*dim] (flag bit 1),
 *     incidence f64[nnz], mass_matrix_inverse f64[nnz] (flag bit 2)
 *   uint64   FNV-1a checksum of everything before it
 * One file per rank. The reader validates sizes, index ranges, the diagonal-first convention and the
 * checksum; a file that fails any check is rejected (NULL + message), never partially loaded.
 */
#ifndef RYUJIN_OFFLINE_IO_H
#define RYUJIN_OFFLINE_IO_H

#include "ryujin_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ryujin_offline_file ryujin_offline_file;

/* positions [n_relevant*dim] / b_positions [n_bdry*dim] may be NULL (the host then has to evaluate
 * initial and Dirichlet data itself). Returns RYUJIN_OK or RYUJIN_ERR_ARG. */
int ryujin_offline_write(const char *path, const ryujin_hip_offline *offline, int dim,
                         int n_initial_precomputed, const double *positions, const double *b_positions);

/* returns NULL on error (message via ryujin_offline_io_last_error) */
ryujin_offline_file *ryujin_offline_read(const char *path);
void ryujin_offline_file_free(ryujin_offline_file *f);
const char *ryujin_offline_io_last_error(void);

/* views valid until ryujin_offline_file_free */
const ryujin_hip_offline *ryujin_offline_file_view(const ryujin_offline_file *f);
int ryujin_offline_file_dim(const ryujin_offline_file *f);
int ryujin_offline_file_n_initial_precomputed(const ryujin_offline_file *f);
uint64_t ryujin_offline_file_nnz(const ryujin_offline_file *f);
const double *ryujin_offline_file_positions(const ryujin_offline_file *f);   /* NULL if absent */
const double *ryujin_offline_file_b_positions(const ryujin_offline_file *f); /* NULL if absent */

#ifdef __cplusplus
}
#endif
#endif

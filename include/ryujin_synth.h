/*
 * ryujin_synth.h -- synthetic OfflineData on structured Q1 meshes.
 *
 * The reference builds OfflineData with deal.II (source/offline_data.template.h,
 * out of scope: SURVEY.md section 2 row 9). This generator produces the same
 * arrays -- in the reference's conventions -- for uniform Cartesian Q1 meshes
 * with optional cell cut-outs (forward-facing step, staircase cylinder), so that
 * the hot path can be driven and measured without deal.II:
 *   m_ij = int phi_i phi_j, c_ij = int phi_i grad phi_j  (offline_data.template.h:566-576)
 *   m_i  = sum_j m_ij                                     (offline_data.template.h:790-802)
 *   boundary map with merged/normalised normals           (offline_data.template.h:1246-1361)
 *   coupling boundary pairs                               (offline_data.template.h:1364-1461)
 *   export-first local numbering, ghosts sorted by owner  (offline_data.template.h:210-249)
 * Closed-form element matrices are derived in SURVEY.md Appendix D.
 */
#ifndef RYUJIN_SYNTH_H
#define RYUJIN_SYNTH_H

#include "ryujin_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { RYUJIN_CUT_NONE = 0, RYUJIN_CUT_BOX = 1, RYUJIN_CUT_CYLINDER = 2 };

typedef struct ryujin_synth_spec {
  int dim;                /* 1, 2, 3 */
  uint32_t n_cells[3];    /* cells per direction (global) */
  double lower[3], upper[3];
  int bc[6];              /* RYUJIN_BC_* on faces -x,+x,-y,+y,-z,+z */
  int cut_kind;           /* RYUJIN_CUT_*: cells whose CENTRE lies inside are removed */
  double cut_lo[3], cut_hi[3];          /* box */
  double cyl_center[2], cyl_radius;     /* cylinder axis along z through (cx,cy) */
  int cut_bc;             /* RYUJIN_BC_* on faces exposed by the cut-out */
  int n_ranks, rank;      /* slab partition of the node planes along x */
} ryujin_synth_spec;

typedef struct ryujin_synth ryujin_synth;

/* returns NULL on error (message via ryujin_synth_last_error) */
ryujin_synth *ryujin_synth_build(const ryujin_synth_spec *spec);
void ryujin_synth_free(ryujin_synth *s);
const char *ryujin_synth_last_error(void);

/* view valid until ryujin_synth_free */
const ryujin_hip_offline *ryujin_synth_offline(const ryujin_synth *s);
uint64_t ryujin_synth_nnz(const ryujin_synth *s);           /* incl. ghost rows */
uint64_t ryujin_synth_n_global(const ryujin_synth *s);      /* sum of n_owned over ranks */
const double *ryujin_synth_positions(const ryujin_synth *s);   /* [n_relevant*dim] */
const uint64_t *ryujin_synth_global_ids(const ryujin_synth *s); /* [n_relevant] lexicographic id of the node in the full grid */
const double *ryujin_synth_bdry_positions(const ryujin_synth *s); /* [n_bdry*dim] */

/* exported instance of ryujin_ghost_row_send_entries() (include/ryujin_exchange_lists.h): the ghost-row
 * send-list rule of sparse_matrix_simd.template.h:196-264, the function the generator itself calls; bound by
 * the test partitioner and pinned against the reference's 4-rank baseline (tests/test_send_lists_golden.py) */
size_t ryujin_synth_ghost_row_send_entries(const uint64_t *row_starts, const uint32_t *columns,
                                           const uint32_t *exported_rows, size_t n_exported,
                                           uint32_t ghost_begin, uint32_t ghost_end, uint32_t *out_row,
                                           uint32_t *out_col);

#ifdef __cplusplus
}
#endif
#endif

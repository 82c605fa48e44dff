"""Test helpers: build the reference's SparsityPatternSIMD storage (SIMD-interleaved internal rows +
CSR remainder; source/sparse_matrix_simd.template.h:96-127, sparse_matrix_simd.h:403-418) from a
plain CSR, and renumber a synthetic OfflineData so that rows of equal stencil size come first (what
the reference's binning does, source/local_index_handling.h:314-375)."""
import ctypes as C

import numpy as np

from ryujin_amd import capi


def simd_layout_from_rows(rows, n_internal, sl):
    """rows: list of lists of column indices (diagonal first). Returns (row_starts[u64, n+1], columns)
    in the reference layout with `n_internal` SIMD rows of width `sl`."""
    n = len(rows)
    assert n_internal % sl == 0
    row_starts = np.zeros(n + 1, dtype=np.uint64)
    cols = []
    for g in range(n_internal // sl):
        grp = rows[g * sl:(g + 1) * sl]
        length = len(grp[0])
        assert all(len(r) == length for r in grp), "rows of one SIMD group must have equal length"
        for c in range(length):
            for k in range(sl):
                cols.append(grp[k][c])
        row_starts[g + 1] = len(cols)
    row_starts[n_internal] = row_starts[n_internal // sl]
    for i in range(n_internal, n):
        cols.extend(rows[i])
        row_starts[i + 1] = len(cols)
    return row_starts, np.array(cols, dtype=np.uint32)


def data_pos(row_starts, n_internal, sl, row, col_idx, n_comp=1, d=0):
    if row < n_internal:
        return int((int(row_starts[row // sl]) + col_idx * sl) * n_comp + d * sl + row % sl)
    return int((int(row_starts[row]) + col_idx) * n_comp + d)


class OfflineView:
    """Minimal stand-in for ryujin_amd.offline.SyntheticOffline backed by numpy arrays."""

    def __init__(self, dim, n_export, n_internal, n_owned, n_relevant, sl, row_starts, columns, cij, mij,
                 mi, mi_inv, measure, b_i, b_normal, b_id, p_i, p_col, p_j):
        self.dim = dim
        self._keep = dict(row_starts=np.ascontiguousarray(row_starts, dtype=np.uint64),
                          columns=np.ascontiguousarray(columns, dtype=np.uint32),
                          cij=np.ascontiguousarray(cij, dtype=np.float64),
                          mij=np.ascontiguousarray(mij, dtype=np.float64),
                          mi=np.ascontiguousarray(mi, dtype=np.float64),
                          mi_inv=np.ascontiguousarray(mi_inv, dtype=np.float64),
                          b_i=np.ascontiguousarray(b_i, dtype=np.uint32),
                          b_normal=np.ascontiguousarray(b_normal, dtype=np.float64),
                          b_id=np.ascontiguousarray(b_id, dtype=np.uint8),
                          p_i=np.ascontiguousarray(p_i, dtype=np.uint32),
                          p_col=np.ascontiguousarray(p_col, dtype=np.uint32),
                          p_j=np.ascontiguousarray(p_j, dtype=np.uint32))
        k = self._keep
        o = capi.Offline()
        o.n_export, o.n_internal, o.n_owned, o.n_relevant, o.simd_length = n_export, n_internal, n_owned, n_relevant, sl
        o.row_starts = capi.as_ptr(k["row_starts"], capi.c_u64_p)
        o.columns = capi.as_ptr(k["columns"], capi.c_u32_p)
        o.cij = capi.as_ptr(k["cij"], capi.c_double_p)
        o.mij = capi.as_ptr(k["mij"], capi.c_double_p)
        o.mi = capi.as_ptr(k["mi"], capi.c_double_p)
        o.mi_inv = capi.as_ptr(k["mi_inv"], capi.c_double_p)
        o.measure_of_omega = measure
        o.n_bdry = len(k["b_i"])
        o.b_i = capi.as_ptr(k["b_i"], capi.c_u32_p)
        o.b_normal = capi.as_ptr(k["b_normal"], capi.c_double_p)
        o.b_id = capi.as_ptr(k["b_id"], capi.c_u8_p)
        o.n_pairs = len(k["p_i"])
        o.p_i = capi.as_ptr(k["p_i"], capi.c_u32_p)
        o.p_col = capi.as_ptr(k["p_col"], capi.c_u32_p)
        o.p_j = capi.as_ptr(k["p_j"], capi.c_u32_p)
        o.n_nbr = 0
        self._o = o
        self.c = C.pointer(o)
        self.n_export, self.n_internal, self.n_owned, self.n_relevant = n_export, n_internal, n_owned, n_relevant
        self.n_bdry, self.n_pairs = o.n_bdry, o.n_pairs
        self.measure_of_omega = measure
        self.row_starts = k["row_starts"]
        self.mi = k["mi"]

    def set_initial_precomputed(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._keep["initial_precomputed"] = v
        self._o.initial_precomputed = capi.as_ptr(v, capi.c_double_p)


def offline_data_numbering(rows, sl):
    """The numbering OfflineData::setup() gives a single rank's degrees of freedom
    (source/offline_data.template.h:204-232): Cuthill-McKee on the stencil graph (scipy's, un-reversed -- deal.II's
    picks its starting nodes differently, the shape of the result is the same: neighbours close together), then
    DoFRenumbering::internal_range (source/local_index_handling.h:314-368) restated: walk the rows in that order,
    collect them in bins by stencil size, and whenever a bin holds `sl` rows hand out the next `sl` numbers to it
    (ascending within the bin, it is a std::set); what is left in the bins at the end follows, by stencil size and
    index (std::map of std::set). Returns (order[new] = old, n_internal)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    n = len(rows)
    indptr = np.cumsum([0] + [len(r) for r in rows])
    indices = np.concatenate([np.asarray(r, dtype=np.int64) for r in rows])
    graph = csr_matrix((np.ones(len(indices)), indices, indptr), shape=(n, n))
    cm = np.asarray(reverse_cuthill_mckee(graph, symmetric_mode=True))[::-1]      # cm[new] = old
    order, bins = [], {}
    for idx in range(n):
        length = len(rows[cm[idx]])
        bins.setdefault(length, []).append(idx)
        if len(bins[length]) == sl:
            order.extend(bins.pop(length))
    n_internal = len(order)
    for length in sorted(bins):
        order.extend(bins[length])
    assert sorted(order) == list(range(n)) and n_internal % sl == 0
    return cm[np.asarray(order, dtype=np.int64)], n_internal


def to_simd_layout(off, sl, order=None, n_internal=None):
    """Renumber a single-rank OfflineData (SyntheticOffline, or the P1 view of tests/helpers_unstructured.py) and
    store it in the reference's SIMD-interleaved layout. order[new] = old; default: rows sorted by (descending)
    stencil size, n_internal = the longest prefix of full groups of equal length. Returns an OfflineView with
    .new_index[old] = new and .order."""
    assert off.n_relevant == off.n_owned
    n, dim = off.n_owned, off.dim
    rs = off.row_starts.astype(np.int64)
    cols = off.columns.astype(np.int64)
    cij = off.cij_csr if hasattr(off, "cij_csr") else off.cij
    mij = off.mij_csr if hasattr(off, "mij_csr") else off.mij
    keep = getattr(off, "_keep", None)
    b_i_old = (keep["b_i"] if keep is not None and "b_i" in keep else off.b_i).astype(np.int64)
    b_normal = keep["b_normal"] if keep is not None and "b_normal" in keep else off.b_normal
    b_id = keep["b_id"] if keep is not None and "b_id" in keep else off.b_id
    if keep is not None and "p_i" in keep:
        p_i_old, p_j_old = keep["p_i"], keep["p_j"]
    else:
        p_i_old, _, p_j_old = off.pairs
    lengths = np.diff(rs)
    if order is None:
        order = np.argsort(-lengths, kind="stable")       # new -> old
    order = np.asarray(order, dtype=np.int64)
    new_index = np.empty(n, dtype=np.int64)
    new_index[order] = np.arange(n)
    rows, row_c, row_m = [], [], []
    for new in range(n):
        old = order[new]
        e = np.arange(rs[old], rs[old + 1])
        jn = new_index[cols[e]]
        key = np.where(jn == new, -1, jn)
        srt = np.argsort(key, kind="stable")
        rows.append(jn[srt].tolist())
        row_c.append(cij[e][srt])
        row_m.append(mij[e][srt])
    new_len = lengths[order]
    if n_internal is None:
        # largest prefix of full groups with equal lengths
        n_internal = 0
        while n_internal + sl <= n and len(set(new_len[n_internal:n_internal + sl].tolist())) == 1:
            n_internal += sl
    row_starts, columns = simd_layout_from_rows(rows, n_internal, sl)
    nnz = len(columns)
    cdata = np.zeros(nnz * dim)
    mdata = np.zeros(nnz)
    for i in range(n):
        for c in range(len(rows[i])):
            mdata[data_pos(row_starts, n_internal, sl, i, c)] = row_m[i][c]
            for d in range(dim):
                cdata[data_pos(row_starts, n_internal, sl, i, c, dim, d)] = row_c[i][c][d]
    mi = off.mi[order]
    b_i = new_index[b_i_old]
    p_i = new_index[np.asarray(p_i_old).astype(np.int64)]
    p_j = new_index[np.asarray(p_j_old).astype(np.int64)]
    p_col = np.array([rows[i].index(j) for i, j in zip(p_i.tolist(), p_j.tolist())], dtype=np.uint32)
    view = OfflineView(dim, 0, n_internal, n, n, sl, row_starts, columns, cdata, mdata, mi, 1.0 / mi,
                       off.measure_of_omega, b_i, b_normal, b_id, p_i, p_col, p_j)
    view.new_index = new_index
    view.order = order
    view.new_lengths = new_len
    return view

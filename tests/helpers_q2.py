"""Continuous Q2 elements on a periodic Cartesian mesh, assembled for the tests the way the reference assembles its
matrices (source/offline_data.template.h:560-674: c_ij = int phi_i grad phi_j, m_ij = int phi_i phi_j, lumped masses
m_i = sum_j m_ij): rows of 27 ... 125 entries in 3-D (9 ... 25 in 2-D) -- the stencil sizes of the reference's higher
order ansatz spaces (source/discretization.h:131-151), wider than one SELL-64 slice has lanes. Tensor products of the
1-D element matrices; periodic in every direction (no boundary map, c_ij antisymmetric everywhere)."""
import numpy as np

from helpers_layout import OfflineView


def _q2_1d(n_elements, h):
    """Assembled periodic 1-D Q2 matrices on n_elements elements of size h: nodes 2e (vertex), 2e + 1 (mid point)."""
    M_loc = h / 30.0 * np.array([[4.0, 2.0, -1.0], [2.0, 16.0, 2.0], [-1.0, 2.0, 4.0]])
    D_loc = np.array([[-3.0, 4.0, -1.0], [-4.0, 0.0, 4.0], [1.0, -4.0, 3.0]]) / 6.0   # int phi_a phi_b'
    n = 2 * n_elements
    M, D = np.zeros((n, n)), np.zeros((n, n))
    for e in range(n_elements):
        idx = [2 * e, 2 * e + 1, (2 * e + 2) % n]
        for a in range(3):
            for b in range(3):
                M[idx[a], idx[b]] += M_loc[a, b]
                D[idx[a], idx[b]] += D_loc[a, b]
    return M, D


def q2_periodic_offline(dim, n_elements, length=1.0):
    """(OfflineView, positions): periodic Q2 mesh of n_elements^dim elements on [0, length)^dim."""
    assert n_elements >= 3, "five distinct nodes per direction in a vertex row"
    h = length / n_elements
    M1, D1 = _q2_1d(n_elements, h)
    n1 = M1.shape[0]
    pattern1 = M1 != 0.0
    n = n1 ** dim

    def kron(mats):
        out = mats[0]
        for m in mats[1:]:
            out = np.kron(out, m)
        return out
    M = kron([M1] * dim)
    C = [kron([D1 if e == d else M1 for e in range(dim)]) for d in range(dim)]
    pattern = kron([pattern1.astype(float)] * dim) != 0.0
    rows = []
    for i in range(n):
        js = np.flatnonzero(pattern[i])
        rows.append([i] + [int(j) for j in js if j != i])          # diagonal first, then ascending
    row_starts = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    columns = np.concatenate([np.array(r, dtype=np.uint32) for r in rows])
    ii = np.repeat(np.arange(n), [len(r) for r in rows])
    cij = np.stack([Cd[ii, columns] for Cd in C], axis=1)
    mij = M[ii, columns]
    mi = M.sum(axis=1)
    assert (mi > 0).all() and np.abs(sum(Cd.sum(axis=1) for Cd in C)).max() < 1e-13
    off = OfflineView(dim, 0, 0, n, n, 1, row_starts, columns, cij, mij, mi, 1.0 / mi, mi.sum(), [],
                      np.zeros((0, dim)), [], [], [], [])
    # node coordinates, first index slowest (np.kron ordering)
    x1 = 0.5 * h * np.arange(n1)
    grids = np.meshgrid(*([x1] * dim), indexing="ij")
    positions = np.stack([g.reshape(-1) for g in grids], axis=1)
    off.positions = positions
    off.row_starts, off.columns, off.mi = row_starts, columns, mi
    off.max_row_len = max(len(r) for r in rows)
    return off, positions

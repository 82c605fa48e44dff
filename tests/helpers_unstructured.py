"""Test helper: OfflineData of a genuinely unstructured mesh -- continuous P1 elements on a Delaunay
triangulation of a disk -- assembled with numpy exactly as the reference assembles its matrices
(source/offline_data.template.h:566-576 `c_ij = int phi_i grad phi_j`, `m_ij = int phi_i phi_j`;
:790-802 lumped mass; :1246-1361 boundary normals = normalised sum of the face integrals of phi_i n;
:1369-1463 coupling boundary pairs). Nothing on the hot path is specific to Q1 quadrilaterals: it sees
row lengths between 4 and ~10, c_ij that are antisymmetric only in the interior, boundary normals in every
direction and varying m_i."""
import numpy as np
from scipy.spatial import Delaunay

from helpers_layout import OfflineView
from ryujin_amd import capi


def disk_points(n_rings, seed=7, jitter=0.25):
    """Concentric rings of points (radius k / n_rings, 6 k points each), jittered inside, exact on the
    boundary ring, so that the hull of the triangulation is a regular polygon inscribed in the unit circle."""
    rng = np.random.default_rng(seed)
    pts = [np.zeros((1, 2))]
    for k in range(1, n_rings + 1):
        m = 6 * k
        phi = 2.0 * np.pi * (np.arange(m) + 0.5 * (k % 2)) / m
        r = np.full(m, k / n_rings)
        if k < n_rings:
            r = r + jitter / n_rings * rng.uniform(-1.0, 1.0, m)
            phi = phi + jitter * 2.0 * np.pi / m * rng.uniform(-1.0, 1.0, m)
        pts.append(np.column_stack([r * np.cos(phi), r * np.sin(phi)]))
    return np.concatenate(pts)


def p1_offline(points, boundary_id=capi.BC_SLIP):
    """Assemble the OfflineData arrays of continuous P1 elements on the Delaunay triangulation of `points`.
    Returns (OfflineView, dict with triangles, boundary edges and the plain-CSR arrays)."""
    tri = Delaunay(points)
    T = tri.simplices
    n = len(points)
    x = points
    c_acc, m_acc = {}, {}
    area_total = 0.0
    for t in T:
        p = x[t]
        d1, d2 = p[1] - p[0], p[2] - p[0]
        det = d1[0] * d2[1] - d1[1] * d2[0]
        if det < 0:  # orient counter-clockwise
            t = t[[0, 2, 1]]
            p = x[t]
            det = -det
        A = 0.5 * det
        area_total += A
        # grad phi_a = rot90(edge opposite to a) / (2 A)
        grads = np.empty((3, 2))
        for a in range(3):
            e = p[(a + 2) % 3] - p[(a + 1) % 3]
            grads[a] = np.array([-e[1], e[0]]) / (2.0 * A)
        for a in range(3):
            for b in range(3):
                key = (int(t[a]), int(t[b]))
                c_acc[key] = c_acc.get(key, 0.0) + A / 3.0 * grads[b]
                m_acc[key] = m_acc.get(key, 0.0) + A / 12.0 * (2.0 if a == b else 1.0)
    # boundary edges: those of the convex hull; outward normal integral of phi_i over the edge = |e| n / 2
    hull = tri.convex_hull
    centre = x.mean(axis=0)
    nrm = np.zeros((n, 2))
    is_bdry = np.zeros(n, dtype=bool)
    for e in hull:
        a, b = x[e[0]], x[e[1]]
        t = b - a
        nu = np.array([t[1], -t[0]])
        if np.dot(nu, 0.5 * (a + b) - centre) < 0:
            nu = -nu
        for v in e:  # |e|/2 * unit normal = nu / 2
            nrm[v] += 0.5 * nu
            is_bdry[v] = True
    rows = [[i] for i in range(n)]
    for (i, j) in c_acc:
        if i != j:
            rows[i].append(j)
    rows = [[r[0]] + sorted(r[1:]) for r in rows]
    row_starts = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    columns = np.concatenate([np.array(r, dtype=np.uint32) for r in rows])
    cij = np.array([c_acc[(i, j)] for i, r in enumerate(rows) for j in r])
    mij = np.array([m_acc[(i, j)] for i, r in enumerate(rows) for j in r])
    mi = np.add.reduceat(mij, row_starts[:-1].astype(np.int64))
    b_i = np.flatnonzero(is_bdry).astype(np.uint32)
    b_normal = nrm[b_i] / np.linalg.norm(nrm[b_i], axis=1)[:, None]
    p_i, p_col, p_j = [], [], []
    for i in b_i:
        for col_idx, j in enumerate(rows[i]):
            if col_idx > 0 and is_bdry[j]:
                p_i.append(i), p_col.append(col_idx), p_j.append(j)
    off = OfflineView(2, 0, 0, n, n, 1, row_starts, columns, cij, mij, mi, 1.0 / mi, mi.sum(), b_i, b_normal,
                      np.full(len(b_i), boundary_id, dtype=np.uint8), p_i, p_col, p_j)
    off.positions = x
    off.row_starts, off.columns, off.cij_csr, off.mij_csr, off.mi = row_starts, columns, cij, mij, mi
    return off, dict(triangles=T, area=area_total, rows=rows, is_bdry=is_bdry)

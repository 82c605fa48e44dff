"""Test helper: OfflineData of continuous Q1 elements on a mesh of GENERAL (bilinear, non-rectangular) quadrilaterals
-- an annulus between two circles, the geometry family of the reference's check-mass-conservation_02 and of its
cylinder benchmarks, where deal.II's cells are no longer axis-parallel -- assembled with numpy as the reference
assembles its matrices (source/offline_data.template.h:566-576 `c_ij = int phi_i grad phi_j`, `m_ij = int phi_i
phi_j` with the cell's Jacobian; :790-802 lumped mass; :1246-1361 boundary normals = normalised sum of the face
integrals of phi_i n; :1369-1463 coupling boundary pairs; 2 x 2 Gauss points, exact for the bilinear products on
parallelograms and what QGauss(2) gives deal.II on a general cell).

This is NOT deal.II's annulus mesh (geometry_annulus.h builds that from hyper_ball_balanced, hyper_shell, a
transfinite manifold and merge_triangulations): it is a polar mesh of the same domain. What it shares with a real
ryujin mesh and the Cartesian generator lacks: Jacobians that vary inside a cell, c_ij without any symmetry in the
coordinate directions, curved slip walls whose normals turn from node to node, two walls with opposite curvature."""
import numpy as np

from helpers_layout import OfflineView
from ryujin_amd import capi

_G = 1.0 / np.sqrt(3.0)
_GAUSS = [(-_G, -_G), (_G, -_G), (_G, _G), (-_G, _G)]  # weights 1


def _shape(xi, eta):
    """values and reference gradients of the four bilinear shape functions (counter-clockwise vertices)"""
    sx = np.array([-1.0, 1.0, 1.0, -1.0])
    sy = np.array([-1.0, -1.0, 1.0, 1.0])
    phi = 0.25 * (1.0 + sx * xi) * (1.0 + sy * eta)
    dphi = np.stack([0.25 * sx * (1.0 + sy * eta), 0.25 * sy * (1.0 + sx * xi)], axis=1)  # [4, 2]
    return phi, dphi


def q1_quads_offline(points, quads, boundary_edges, boundary_id=capi.BC_SLIP):
    """points [n, 2]; quads [n_cells, 4] counter-clockwise; boundary_edges [n_edges, 2] with the domain on the LEFT of
    the edge (a -> b): the outward normal is (t_y, -t_x). Returns (OfflineView, dict)."""
    x = np.asarray(points, dtype=np.float64)
    n = len(x)
    c_acc, m_acc = {}, {}
    area = 0.0
    for q in quads:
        p = x[q]                                          # [4, 2]
        for xi, eta in _GAUSS:
            phi, dphi = _shape(xi, eta)
            J = dphi.T @ p                                # J[a, b] = d x_b / d xi_a
            det = np.linalg.det(J)
            assert det > 0.0, "cells must be counter-clockwise and convex"
            grad = dphi @ np.linalg.inv(J).T              # physical gradients [4, 2]
            area += det
            for a in range(4):
                for b in range(4):
                    key = (int(q[a]), int(q[b]))
                    c_acc[key] = c_acc.get(key, 0.0) + phi[a] * grad[b] * det
                    m_acc[key] = m_acc.get(key, 0.0) + phi[a] * phi[b] * det
    nrm = np.zeros((n, 2))
    is_bdry = np.zeros(n, dtype=bool)
    for a, b in boundary_edges:
        t = x[b] - x[a]
        nu = np.array([t[1], -t[0]])                      # |e| n, outward when the domain is on the left of a -> b
        for v in (a, b):
            nrm[v] += 0.5 * nu                            # int_e phi_v n dS
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
    return off, dict(rows=rows, is_bdry=is_bdry, area=area, boundary_normals_raw=nrm)


def annulus_mesh(n_r, n_theta, r_inner=0.4, r_outer=1.0, skew=0.15):
    """A polar mesh of the annulus r_inner <= r <= r_outer: (n_r + 1) x n_theta nodes, periodic in theta, the interior
    rings twisted against each other (skew, in units of the angular spacing) and graded towards the inner wall so
    that no cell is a rectangle or a parallelogram. Returns (points, quads, boundary_edges)."""
    s = np.linspace(0.0, 1.0, n_r + 1)
    radii = r_inner + (r_outer - r_inner) * s ** 1.3
    dth = 2.0 * np.pi / n_theta
    pts = np.zeros(((n_r + 1) * n_theta, 2))
    for a in range(n_r + 1):
        twist = skew * dth * np.sin(np.pi * s[a]) * (1 if a % 2 else -1)
        th = dth * np.arange(n_theta) + twist
        pts[a * n_theta:(a + 1) * n_theta, 0] = radii[a] * np.cos(th)
        pts[a * n_theta:(a + 1) * n_theta, 1] = radii[a] * np.sin(th)
    node = lambda a, b: a * n_theta + b % n_theta  # noqa: E731
    # counter-clockwise: (r, th) -> (r+, th) -> (r+, th+) -> (r, th+)
    quads = np.array([[node(a, b), node(a + 1, b), node(a + 1, b + 1), node(a, b + 1)]
                      for a in range(n_r) for b in range(n_theta)])
    # the domain on the left: the outer circle counter-clockwise, the inner circle clockwise
    edges = [(node(n_r, b), node(n_r, b + 1)) for b in range(n_theta)] + \
            [(node(0, b + 1), node(0, b)) for b in range(n_theta)]
    return pts, quads, np.array(edges)


# ---------------------------------------------------------------------------------------------------- 3-D: hexahedra

def _gauss(npts):
    x, w = np.polynomial.legendre.leggauss(npts)
    return list(zip(x.tolist(), w.tolist()))


def _shape_hex(xi, eta, zeta):
    """trilinear shape functions, vertex v = ix + 2 iy + 4 iz at (+-1, +-1, +-1)"""
    s = np.array([[(-1.0, 1.0)[v & 1], (-1.0, 1.0)[(v >> 1) & 1], (-1.0, 1.0)[(v >> 2) & 1]] for v in range(8)])
    f = np.stack([1.0 + s[:, 0] * xi, 1.0 + s[:, 1] * eta, 1.0 + s[:, 2] * zeta], axis=1)
    phi = 0.125 * f[:, 0] * f[:, 1] * f[:, 2]
    dphi = 0.125 * np.stack([s[:, 0] * f[:, 1] * f[:, 2], f[:, 0] * s[:, 1] * f[:, 2], f[:, 0] * f[:, 1] * s[:, 2]],
                            axis=1)
    return phi, dphi


def q1_hexes_offline(points, hexes, boundary_faces, boundary_id=capi.BC_SLIP):
    """Continuous Q1 on general (trilinear, non-planar-faced) hexahedra. hexes [n_cells, 8] in the vertex order of
    _shape_hex with a right-handed mapping; boundary_faces [n_faces, 4] counter-clockwise seen from OUTSIDE.
    3 x 3 x 3 Gauss points: c_ij + c_ji = int cof(J) grad(phi_i phi_j) has degree 4 per variable on a skewed cell,
    and the scheme's conservation needs it to vanish in the interior to round-off (2 x 2 x 2 points integrate degree 3)."""
    x = np.asarray(points, dtype=np.float64)
    n = len(x)
    c_acc, m_acc = {}, {}
    volume = 0.0
    g3 = _gauss(3)
    for q in hexes:
        p = x[q]
        for xi, wx in g3:
            for eta, wy in g3:
                for zeta, wz in g3:
                    phi, dphi = _shape_hex(xi, eta, zeta)
                    J = dphi.T @ p
                    det = np.linalg.det(J)
                    assert det > 0.0
                    w = wx * wy * wz * det
                    grad = dphi @ np.linalg.inv(J).T
                    volume += w
                    for a in range(8):
                        for b in range(8):
                            key = (int(q[a]), int(q[b]))
                            c_acc[key] = c_acc.get(key, 0.0) + phi[a] * grad[b] * w
                            m_acc[key] = m_acc.get(key, 0.0) + phi[a] * phi[b] * w
    nrm = np.zeros((n, 3))
    is_bdry = np.zeros(n, dtype=bool)
    for f in boundary_faces:
        p = x[f]
        for xi, eta in _GAUSS:
            psi, dpsi = _shape(xi, eta)
            t1, t2 = dpsi[:, 0] @ p, dpsi[:, 1] @ p
            nds = np.cross(t1, t2)                        # n dS
            for k in range(4):
                nrm[f[k]] += psi[k] * nds
        is_bdry[f] = True
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
    off = OfflineView(3, 0, 0, n, n, 1, row_starts, columns, cij, mij, mi, 1.0 / mi, mi.sum(), b_i, b_normal,
                      np.full(len(b_i), boundary_id, dtype=np.uint8), p_i, p_col, p_j)
    off.positions = x
    off.row_starts, off.columns, off.cij_csr, off.mij_csr, off.mi = row_starts, columns, cij, mij, mi
    return off, dict(rows=rows, is_bdry=is_bdry, volume=volume, boundary_normals_raw=nrm)


def annulus_mesh_3d(n_r, n_theta, n_z, r_inner=0.4, r_outer=1.0, height=0.5, skew=0.15):
    """The annulus extruded in z between two flat lids, every layer twisted against the next and the interior radii
    breathing with z, so that the cell faces are not planar. Returns (points, hexes, boundary_faces)."""
    s = np.linspace(0.0, 1.0, n_r + 1)
    dth = 2.0 * np.pi / n_theta
    node = lambda a, b, c: (c * (n_r + 1) + a) * n_theta + b % n_theta  # noqa: E731
    pts = np.zeros(((n_r + 1) * n_theta * (n_z + 1), 3))
    for c in range(n_z + 1):
        zc = c / n_z
        for a in range(n_r + 1):
            interior = np.sin(np.pi * s[a])
            radius = r_inner + (r_outer - r_inner) * (s[a] ** 1.3 + 0.04 * interior * np.sin(2.0 * np.pi * zc))
            twist = skew * dth * (interior * (1 if a % 2 else -1) + 0.6 * np.sin(np.pi * zc) * (1 if c % 2 else -1))
            th = dth * np.arange(n_theta) + twist
            idx = [node(a, b, c) for b in range(n_theta)]
            pts[idx, 0], pts[idx, 1], pts[idx, 2] = radius * np.cos(th), radius * np.sin(th), height * zc
    hexes = np.array([[node(a + ix, b + iy, c + iz) for iz in (0, 1) for iy in (0, 1) for ix in (0, 1)]
                      for c in range(n_z) for a in range(n_r) for b in range(n_theta)])
    faces = []
    for c in range(n_z):
        for b in range(n_theta):
            faces.append([node(n_r, b, c), node(n_r, b + 1, c), node(n_r, b + 1, c + 1), node(n_r, b, c + 1)])  # outer
            faces.append([node(0, b, c), node(0, b, c + 1), node(0, b + 1, c + 1), node(0, b + 1, c)])          # inner
    for a in range(n_r):
        for b in range(n_theta):
            faces.append([node(a, b, n_z), node(a + 1, b, n_z), node(a + 1, b + 1, n_z), node(a, b + 1, n_z)])  # top
            faces.append([node(a, b, 0), node(a, b + 1, 0), node(a + 1, b + 1, 0), node(a + 1, b, 0)])          # bottom
    return pts, hexes, np.array(faces)

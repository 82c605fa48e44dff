"""The hot path on an unstructured mesh: continuous P1 triangles on a Delaunay triangulation of a disk
(tests/helpers_unstructured.py), slip walls all around. Checks the mesh data themselves (partition of unity,
antisymmetry of c_ij in the interior, boundary mass on the rim) and that the oracle conserves mass and energy
to round-off there -- which holds exactly when the boundary normals are consistent with the c_ij, the
property the reference's slip condition relies on (source/euler/hyperbolic_system.h:1108-1112)."""
import numpy as np

from helpers_unstructured import disk_points, p1_offline
from ryujin_amd import HyperbolicModule, TimeIntegrator, capi
from ryujin_amd.initial_states import euler_radial_contrast


def test_p1_disk_mesh_data():
    off, info = p1_offline(disk_points(12))
    n = off.n_owned
    rs, cols, cij, mij = off.row_starts.astype(np.int64), off.columns, off.cij_csr, off.mij_csr
    lengths = np.diff(rs)
    assert lengths.min() >= 4 and lengths.max() >= 8 and len(set(lengths.tolist())) >= 4   # ragged
    # polygon inscribed in the unit circle with 72 vertices
    m = 72
    assert abs(info["area"] - 0.5 * m * np.sin(2 * np.pi / m)) < 1e-12
    assert abs(off.mi.sum() - info["area"]) < 1e-13
    # partition of unity: sum_j c_ij = 0 in every row, boundary rows included
    row_sum = np.add.reduceat(cij, rs[:-1], axis=0)
    assert np.abs(row_sum).max() < 1e-15
    lookup = {(i, int(cols[e])): e for i in range(n) for e in range(rs[i], rs[i + 1])}
    is_bdry = info["is_bdry"]
    n_interior_pairs = n_rim_pairs = 0
    for (i, j), e in lookup.items():
        if i >= j:
            continue
        s = cij[e] + cij[lookup[(j, i)]]            # = int_boundary phi_i phi_j n dS
        if is_bdry[i] and is_bdry[j]:
            n_rim_pairs += 1
            if np.abs(s).max() > 1e-15:             # a rim edge: |e|/6 n_e ... (P1 boundary mass, off-diagonal)
                edge = off.positions[j] - off.positions[i]
                assert abs(np.linalg.norm(s) - np.linalg.norm(edge) / 6.0) < 1e-14
        else:
            n_interior_pairs += 1
            assert np.abs(s).max() < 1e-15
    assert n_rim_pairs >= m and n_interior_pairs > 1000
    # diagonal of a rim row: c_ii = 1/2 int phi_i^2-type boundary term, points outward like the normal
    bi = np.flatnonzero(is_bdry)
    assert np.all(np.einsum("ij,ij->i", cij[rs[bi]], off._keep["b_normal"]) > 0)
    assert off.n_pairs == 2 * n_rim_pairs


def test_oracle_conserves_on_unstructured_mesh(oracle):
    off, info = p1_offline(disk_points(16))
    U0 = euler_radial_contrast(off.positions, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.35)
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend=oracle.backend())
    sv = m.new_state_vector(U0)
    ti = TimeIntegrator(m, "ssprk 33", cfl_min=0.5, cfl_max=0.5, cfl_recovery_strategy="none")
    mi = off.mi
    before = (mi[:, None] * U0).sum(0)
    t = 0.0
    for _ in range(200):                             # the blast reaches the wall and reflects
        sv, tau = ti.step(sv, t)
        t += tau
    U = sv.download()
    after = (mi[:, None] * U).sum(0)
    assert t > 0.2 and m.n_warnings() == 0
    assert abs(after[0] - before[0]) < 1e-13 * before[0]          # mass
    assert abs(after[3] - before[3]) < 1e-13 * before[3]          # total energy
    rho, mom, E = U[:, 0], U[:, 1:3], U[:, 3]
    assert rho.min() > 0 and (E - 0.5 * (mom ** 2).sum(1) / rho).min() > 0
    # the reflected flow is really there: momentum at rim nodes is tangential after prepare_state_vector
    m.prepare_state_vector(sv, t)
    U = sv.download()
    bi = off._keep["b_i"]
    assert np.abs(U[bi, 1:3]).max() > 1e-3
    assert np.abs(np.einsum("ij,ij->i", U[bi, 1:3], off._keep["b_normal"])).max() < 1e-15

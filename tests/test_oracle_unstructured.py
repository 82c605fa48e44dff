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


def sector_owner(points, n_ranks):
    """Angular sectors around a point off the centre: every rank touches every other one near that point,
    sector boundaries cut the triangulation obliquely, and rim nodes of different ranks are neighbours."""
    phi = np.arctan2(points[:, 1] - 0.07, points[:, 0] + 0.11)
    return ((phi + np.pi) / (2.0 * np.pi) * n_ranks).astype(int) % n_ranks


def test_partition_helper_and_partitioned_oracle(oracle):
    """An arbitrary 4-way partition of the unstructured mesh (up to 3 neighbours per rank, nodes exported to
    several ranks) through the oracle's exchange hooks reproduces the single-rank run."""
    from helpers_unstructured import partition, run_partitioned_oracle
    off, info = p1_offline(disk_points(14))
    owner = sector_owner(off.positions, 4)
    views = partition(off, info, owner)
    assert sum(v.n_owned for v in views) == off.n_owned
    assert max(v._o.n_nbr for v in views) == 3
    multi = 0
    for v in views:
        k = v._keep
        assert k["recv_off"][0] == v.n_owned and k["recv_off"][-1] == v.n_relevant
        idx = k["send_idx"].tolist()
        multi += len(idx) - len(set(idx))
        assert max(idx) < v.n_export
    assert multi > 0                                            # some node goes to two neighbours
    U0 = euler_radial_contrast(off.positions, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.35,
                               center=(0.1, -0.05))
    p = oracle.default_params(capi.EQ_EULER, 2)
    p.cfl = 0.5
    m = HyperbolicModule(off, p, backend=oracle.backend())
    a, b = m.new_state_vector(U0), m.new_state_vector()
    taus = []
    for _ in range(25):
        m.prepare_state_vector(a, 0.0)
        taus.append(m.step(a, [], [], b))
        a, b = b, a
    U_ref = a.download()
    U, taus_p = run_partitioned_oracle(oracle, views, p, U0, 25)
    for tp in taus_p:
        np.testing.assert_allclose(tp, taus, rtol=1e-14)
    # the local numbering changes the summation order within a stencil: round-off, amplified by 25 updates of
    # a blast wave (observed 3e-12 of the component's magnitude)
    assert (np.abs(U - U_ref).max(axis=0) / np.abs(U_ref).max(axis=0)).max() < 1e-10


def test_p1_ball_mesh_and_oracle_conservation_3d(oracle):
    """P1 tetrahedra on a Delaunay tetrahedralisation of a ball (random interior points, Fibonacci lattice
    on the sphere): rows of 6 .. 30 entries, lumped masses spread over two orders of magnitude."""
    from helpers_unstructured import ball_points
    off, info = p1_offline(ball_points(1500, 700))
    rs = off.row_starts.astype(np.int64)
    lengths = np.diff(rs)
    assert lengths.min() >= 5 and 24 <= lengths.max() <= 64 and off.mi.max() / off.mi.min() > 50
    assert 0.98 * 4.0 / 3.0 * np.pi < info["area"] < 4.0 / 3.0 * np.pi
    assert abs(off.mi.sum() - info["area"]) < 1e-12
    assert np.abs(np.add.reduceat(off.cij_csr, rs[:-1], axis=0)).max() < 1e-16
    lookup = {(i, int(off.columns[e])): e for i in range(off.n_owned) for e in range(rs[i], rs[i + 1])}
    is_bdry = info["is_bdry"]
    for (i, j), e in lookup.items():
        if i < j and not (is_bdry[i] and is_bdry[j]):
            assert np.abs(off.cij_csr[e] + off.cij_csr[lookup[(j, i)]]).max() < 1e-16
    U0 = euler_radial_contrast(off.positions, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.5)
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend=oracle.backend())
    sv = m.new_state_vector(U0)
    # (SSPRK: with stage weights the limited fluxes P_ij of rim-rim pairs contain c_ij (f_i + f_j) terms that
    # are not antisymmetric there, c_ij + c_ji != 0, so ERK33 conserves only up to (1 - l_ij) on those pairs
    # once the wave reaches the wall -- 5e-9 here, in 2-D and 3-D alike; a property of the scheme as the
    # reference states it (hyperbolic_module.template.h:797-845), not of the mesh)
    ti = TimeIntegrator(m, "ssprk 33", cfl_min=0.5, cfl_max=0.5, cfl_recovery_strategy="none")
    before = (off.mi[:, None] * U0).sum(0)
    t = 0.0
    for _ in range(40):
        sv, tau = ti.step(sv, t)
        t += tau
    U = sv.download()
    after = (off.mi[:, None] * U).sum(0)
    assert m.n_warnings() == 0
    assert abs(after[0] - before[0]) < 1e-13 * before[0] and abs(after[4] - before[4]) < 1e-13 * before[4]
    assert U[:, 0].min() > 0 and (U[:, 4] - 0.5 * (U[:, 1:4] ** 2).sum(1) / U[:, 0]).min() > 0
    assert np.abs(U[:, 1:4]).max() > 0.1                            # the flow has started


def test_oracle_conserves_on_a_q2_stencil(oracle):
    """Rows of 27 ... 125 entries (continuous Q2, periodic; tests/helpers_q2.py): the oracle's update conserves mass,
    momentum and energy to round-off over 20 SSPRK33 steps of a smooth wave -- c_ij = -c_ji, d_ij = d_ji and a
    symmetric l_ij are all it needs, whatever the stencil."""
    from helpers_q2 import q2_periodic_offline
    off, x = q2_periodic_offline(3, 3)
    assert off.max_row_len == 125 and off.n_owned == 216
    w = np.sin(2.0 * np.pi * x[:, 0]) * np.cos(2.0 * np.pi * x[:, 1])
    rho, p = 1.0 + 0.4 * w, 1.0 + 0.3 * w
    v = np.zeros((off.n_owned, 3))
    v[:, 0], v[:, 1] = 0.5, -0.25
    U0 = np.concatenate([rho[:, None], rho[:, None] * v, (p / 0.4 + 0.5 * rho * (v ** 2).sum(1))[:, None]], axis=1)
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend=oracle.backend())
    sv = m.new_state_vector(U0)
    ti = TimeIntegrator(m, "ssprk 33", cfl_min=0.5, cfl_max=0.5, cfl_recovery_strategy="none")
    before = (off.mi[:, None] * U0).sum(0)
    t = 0.0
    for _ in range(20):
        sv, tau = ti.step(sv, t)
        t += tau
    U = sv.download()
    after = (off.mi[:, None] * U).sum(0)
    assert np.abs(after - before).max() <= 1e-13 * np.abs(before).max()
    assert (U[:, 0] > 0).all() and m.n_warnings() == 0


def test_oracle_on_the_numbering_and_layout_of_offline_data(oracle):
    """The reader of the reference's SIMD-interleaved storage (oracle/csr.hpp) and the numbering helper
    (tests/helpers_layout.py::offline_data_numbering: Cuthill-McKee + DoFRenumbering::internal_range restated) on the
    unstructured P1 disk: 20 updates of a blast wave (round-off through the limiter grows with the updates: 1e-9) on the renumbered, interleaved mesh equal those on the mesh as
    generated up to summation order, and conserve."""
    from helpers_layout import offline_data_numbering, to_simd_layout
    from helpers_unstructured import disk_points, p1_offline
    from ryujin_amd import capi
    from ryujin_amd.initial_states import euler_radial_contrast
    from ryujin_amd.module import HyperbolicModule
    off0, info = p1_offline(disk_points(16))
    order, n_internal = offline_data_numbering(info["rows"], 8)
    off = to_simd_layout(off0, 8, order, n_internal)
    lengths = off.new_lengths
    assert n_internal % 8 == 0 and 0 < n_internal < off.n_owned
    assert all(len(set(lengths[g:g + 8].tolist())) == 1 for g in range(0, n_internal, 8))
    U0 = euler_radial_contrast(off0.positions, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.35)
    out = []
    for view, start in ((off0, U0), (off, U0[order])):
        p = oracle.default_params(capi.EQ_EULER, 2)
        p.cfl = 0.5
        m = HyperbolicModule(view, p, backend=oracle.backend())
        a, b = m.new_state_vector(start), m.new_state_vector()
        for _ in range(20):
            m.prepare_state_vector(a, 0.0)
            m.step(a, [], [], b)
            a, b = b, a
        out.append(a.download())
    ref, got = out[0], out[1][off.new_index]
    assert (np.abs(got - ref) / np.abs(ref).max(axis=0)).max() < 1e-9
    mass0, mass1 = (off0.mi * U0[:, 0]).sum(), (off.mi * out[1][:, 0]).sum()
    assert abs(mass1 - mass0) < 1e-13 * mass0


def test_q1_annulus_mesh_data_and_conservation(oracle):
    """Continuous Q1 on general quadrilaterals (tests/helpers_q1_quads.py): an annulus between two curved slip walls,
    no cell a parallelogram -- the geometry family of the reference's check-mass-conservation_02 (whose deal.II mesh
    itself cannot be rebuilt here). The mesh data have the properties the scheme relies on (rows of 9 and 6 entries,
    partition of unity, c_ij antisymmetric in the interior, boundary integral on wall pairs, consistent normals), and
    the oracle conserves mass and energy to round-off through a blast that reflects off both walls
    (check-mass-conservation_02's property on its geometry)."""
    from helpers_q1_quads import annulus_mesh, q1_quads_offline
    pts, quads, edges = annulus_mesh(14, 72)
    off, info = q1_quads_offline(pts, quads, edges)
    n = off.n_owned
    rs, cols, cij = off.row_starts.astype(np.int64), off.columns, off.cij_csr
    lengths = np.diff(rs)
    assert set(lengths.tolist()) == {6, 9}
    # the area of the two inscribed polygons' difference, and the lumped masses sum to it
    m = 72
    assert abs(info["area"] - 0.5 * m * np.sin(2 * np.pi / m) * (1.0 - 0.4 ** 2)) < 1e-12
    assert abs(off.mi.sum() - info["area"]) < 1e-13
    assert np.abs(np.add.reduceat(cij, rs[:-1], axis=0)).max() < 1e-15          # partition of unity
    lookup = {(i, int(cols[e])): e for i in range(n) for e in range(rs[i], rs[i + 1])}
    is_bdry = info["is_bdry"]
    skew_cells = 0
    for q in quads:
        d = pts[q[0]] - pts[q[1]] + pts[q[2]] - pts[q[3]]
        skew_cells += np.linalg.norm(d) > 1e-3 * np.linalg.norm(pts[q[2]] - pts[q[0]])
    assert skew_cells == len(quads)                                             # no parallelogram anywhere
    for (i, j), e in lookup.items():
        if i < j and not (is_bdry[i] and is_bdry[j]):
            assert np.abs(cij[e] + cij[lookup[(j, i)]]).max() < 1e-15           # antisymmetric off the walls
    # sum_j c_ji = int grad phi_i = int_boundary phi_i n: the raw (unnormalised) normal of the boundary map, on both
    # walls -- the consistency between c_ij and the normals that makes the slip condition conservative
    bi = np.flatnonzero(is_bdry)
    for i in bi:
        s = sum(cij[lookup[(j, i)]] for j in info["rows"][i])
        assert np.abs(s - info["boundary_normals_raw"][i]).max() < 1e-14
    r = np.linalg.norm(off.positions[bi], axis=1)
    outward = np.einsum("ij,ij->i", off._keep["b_normal"], off.positions[bi]) / r
    assert np.all(outward[r > 0.7] > 0.99) and np.all(outward[r < 0.7] < -0.99)  # the inner wall's normals point inwards

    U0 = euler_radial_contrast(off.positions, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.2,
                               center=(0.7, 0.0))
    mod = HyperbolicModule(off, equation=capi.EQ_EULER, backend=oracle.backend())
    sv = mod.new_state_vector(U0)
    ti = TimeIntegrator(mod, "ssprk 33", cfl_min=0.5, cfl_max=0.5, cfl_recovery_strategy="none")
    mi = off.mi
    before = (mi[:, None] * U0).sum(0)
    t = 0.0
    for _ in range(300):
        sv, tau = ti.step(sv, t)
        t += tau
    U = sv.download()
    after = (mi[:, None] * U).sum(0)
    assert t > 0.2 and mod.n_warnings() == 0
    assert abs(after[0] - before[0]) < 1e-13 * before[0]
    assert abs(after[3] - before[3]) < 1e-13 * before[3]
    rho, mom, E = U[:, 0], U[:, 1:3], U[:, 3]
    assert rho.min() > 0 and (E - 0.5 * (mom ** 2).sum(1) / rho).min() > 0
    mod.prepare_state_vector(sv, t)
    U = sv.download()
    assert np.abs(U[bi, 1:3]).max() > 1e-3                                      # the blast has reached both walls
    assert np.abs(U[bi[r < 0.7], 1:3]).max() > 1e-4
    assert np.abs(np.einsum("ij,ij->i", U[bi, 1:3], off._keep["b_normal"])).max() < 1e-15


def test_q1_hexahedra_between_curved_walls_conserve(oracle):
    """The 3-D counterpart: trilinear Q1 on skewed hexahedra with non-planar faces (the annulus extruded between two
    flat lids, every layer twisted and breathing; tests/helpers_q1_quads.py) -- the cell shape of the reference's
    3-D cylinder benchmark near the cylinder. Rows of 12 / 18 / 27 entries, the identities of the 2-D test, and mass
    and energy conserved to round-off by the oracle through a blast between four slip walls."""
    from helpers_q1_quads import annulus_mesh_3d, q1_hexes_offline
    pts, hexes, faces = annulus_mesh_3d(5, 24, 4)
    off, info = q1_hexes_offline(pts, hexes, faces)
    n = off.n_owned
    rs, cols, cij = off.row_starts.astype(np.int64), off.columns, off.cij_csr
    assert set(np.diff(rs).tolist()) == {12, 18, 27}
    assert abs(off.mi.sum() - info["volume"]) < 1e-13
    assert np.abs(np.add.reduceat(cij, rs[:-1], axis=0)).max() < 1e-15
    lookup = {(i, int(cols[e])): e for i in range(n) for e in range(rs[i], rs[i + 1])}
    is_bdry = info["is_bdry"]
    for (i, j), e in lookup.items():
        if i < j and not (is_bdry[i] and is_bdry[j]):
            assert np.abs(cij[e] + cij[lookup[(j, i)]]).max() < 1e-15
    for i in np.flatnonzero(is_bdry):
        s = sum(cij[lookup[(j, i)]] for j in info["rows"][i])
        assert np.abs(s - info["boundary_normals_raw"][i]).max() < 1e-14
    U0 = euler_radial_contrast(off.positions, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.25,
                               center=(0.7, 0.0, 0.25))
    mod = HyperbolicModule(off, equation=capi.EQ_EULER, backend=oracle.backend())
    sv = mod.new_state_vector(U0)
    ti = TimeIntegrator(mod, "ssprk 33", cfl_min=0.5, cfl_max=0.5, cfl_recovery_strategy="none")
    before = (off.mi[:, None] * U0).sum(0)
    t = 0.0
    for _ in range(120):
        sv, tau = ti.step(sv, t)
        t += tau
    U = sv.download()
    after = (off.mi[:, None] * U).sum(0)
    assert t > 0.15 and mod.n_warnings() == 0
    assert abs(after[0] - before[0]) < 1e-13 * before[0]
    assert abs(after[4] - before[4]) < 1e-13 * before[4]
    rho, mom, E = U[:, 0], U[:, 1:4], U[:, 4]
    assert rho.min() > 0 and (E - 0.5 * (mom ** 2).sum(1) / rho).min() > 0

"""Invariants of the synthetic OfflineData generator -- the reference's own DEBUG checks
(source/offline_data.template.h:1002-1104): sum_j c_ij = 0, c_ij = -c_ji in the interior,
sum_j m_ij = m_i, sum_i m_i = |Omega|, diagonal-first rows, sorted columns."""
import numpy as np
import pytest

from ryujin_amd import capi, offline


def _rows(off):
    rs = off.row_starts.astype(np.int64)
    return rs, off.columns.astype(np.int64)


def _check_invariants(off, interior_antisymmetry=True):
    rs, cols = _rows(off)
    c, m, mi = off.cij, off.mij, off.mi
    n = off.n_owned
    row_of = np.repeat(np.arange(off.n_relevant), np.diff(rs))
    owned = row_of < n
    # diagonal first, remaining columns ascending
    assert np.all(cols[rs[:-1]] == np.arange(off.n_relevant))
    for i in np.random.default_rng(0).integers(0, n, size=min(n, 200)):
        r = cols[rs[i] + 1:rs[i + 1]]
        assert np.all(np.diff(r) > 0)
    # sum_j c_ij = 0 (partition of unity) and sum_j m_ij = m_i on owned rows
    csum = np.zeros((off.n_relevant, off.dim))
    np.add.at(csum, row_of[owned], c[owned])
    msum = np.zeros(off.n_relevant)
    np.add.at(msum, row_of[owned], m[owned])
    scale = np.abs(c).max()
    assert np.abs(csum[:n]).max() <= 1e-14 * scale * 30
    np.testing.assert_allclose(msum[:n], mi[:n], rtol=1e-14)
    # symmetric mass matrix, c_ij = -c_ji for interior pairs; c_ij + c_ji = boundary term otherwise
    is_b = np.zeros(off.n_relevant, dtype=bool)
    is_b[off.b_i] = True
    lookup = {}
    for i in range(n):
        for e in range(rs[i], rs[i + 1]):
            lookup[(i, cols[e])] = e
    bad = 0
    for (i, j), e in list(lookup.items())[:: max(1, len(lookup) // 5000)]:
        if j >= n or (j, i) not in lookup:
            continue
        et = lookup[(j, i)]
        assert abs(m[e] - m[et]) <= 1e-15 * abs(m[e]) + 1e-300
        if not (is_b[i] and is_b[j]):
            if np.abs(c[e] + c[et]).max() > 1e-15 * scale:
                bad += 1
    assert bad == 0


def test_rectangle_2d_invariants():
    off = offline.SyntheticOffline(offline.rectangle_2d(16, (0.0, 0.0), (20.0, 20.0)))
    assert off.n_owned == 17 * 17 and off.n_relevant == off.n_owned
    np.testing.assert_allclose(off.mi[: off.n_owned].sum(), 400.0, rtol=1e-13)
    assert abs(off.measure_of_omega - 400.0) < 1e-10
    _check_invariants(off)
    # interior stencil: 9 entries with the closed-form Q1 values
    rs, cols = _rows(off)
    i = 17 * 8 + 8
    assert rs[i + 1] - rs[i] == 9
    h = 20.0 / 16
    np.testing.assert_allclose(off.mij[rs[i]], (2 * h / 3) ** 2, rtol=1e-14)
    np.testing.assert_allclose(off.mi[i], h * h, rtol=1e-14)
    e = rs[i] + list(cols[rs[i]:rs[i + 1]]).index(i + 1)
    np.testing.assert_allclose(off.cij[e], [0.5 * 2 * h / 3, 0.0], atol=1e-15)
    e = rs[i] + list(cols[rs[i]:rs[i + 1]]).index(i + 17 + 1)
    np.testing.assert_allclose(off.cij[e], [0.5 * h / 6, 0.5 * h / 6], atol=1e-15)


def test_boundary_map_rectangle_slip():
    """Straight walls: one merged entry with the outward unit normal. 2-D corners with two
    non-collinear slip normals become no_slip (offline_data.template.h:1313-1336)."""
    off = offline.SyntheticOffline(offline.rectangle_2d(8))
    b_i, b_id, b_n, b_x = off.b_i, off.b_id, off.b_normal, off.b_positions
    # 4*(8-1) wall nodes with one entry + 4 corners with two entries each
    assert len(b_i) == 4 * 7 + 8
    for i, bid, nrm, x in zip(b_i, b_id, b_n, b_x):
        on = [abs(x[0]) < 1e-14, abs(x[0] - 1) < 1e-14, abs(x[1]) < 1e-14, abs(x[1] - 1) < 1e-14]
        if sum(on) == 2:
            assert bid == capi.BC_NO_SLIP
        else:
            assert bid == capi.BC_SLIP
            expect = [(-1, 0), (1, 0), (0, -1), (0, 1)][on.index(True)]
            np.testing.assert_allclose(nrm, expect, atol=1e-15)
    # coupling pairs: both ends are boundary nodes and stencil neighbours
    p_i, p_col, p_j = off.pairs
    rs, cols = _rows(off)
    bset = set(b_i.tolist())
    assert len(p_i) > 0
    for i, c, j in zip(p_i, p_col, p_j):
        assert cols[rs[i] + c] == j and i in bset and j in bset and c >= 1


def test_step_geometry_counts():
    spec = offline.mach3_step_2d(20)
    off = offline.SyntheticOffline(spec)
    # [0,3]x[0,1] minus [0.6,3]x[0,0.2]: area 3 - 2.4*0.2
    assert abs(off.measure_of_omega - (3.0 - 2.4 * 0.2)) < 1e-12
    np.testing.assert_allclose(off.mi[: off.n_owned].sum(), off.measure_of_omega, rtol=1e-13)
    assert off.n_global == off.n_owned
    _check_invariants(off)
    ids = set(off.b_id.tolist())
    assert capi.BC_DIRICHLET in ids and capi.BC_SLIP in ids and capi.BC_DO_NOTHING in ids


def test_box_3d_invariants():
    off = offline.SyntheticOffline(offline.box_3d(6))
    assert off.n_owned == 7 ** 3
    rs, _ = _rows(off)
    i = (3 * 7 + 3) * 7 + 3
    assert rs[i + 1] - rs[i] == 27
    np.testing.assert_allclose(off.mi[: off.n_owned].sum(), 8.0, rtol=1e-13)
    _check_invariants(off)


def test_cylinder_3d_builds():
    off = offline.SyntheticOffline(offline.cylinder_channel_3d(8, length_units=2))
    np.testing.assert_allclose(off.mi[: off.n_owned].sum(), off.measure_of_omega, rtol=1e-13)
    assert off.measure_of_omega < 2 * 2 * 2
    _check_invariants(off)


@pytest.mark.parametrize("n_ranks", [2, 3])
def test_partition_consistency(n_ranks):
    """Slab partition: ownership is a partition of the global node set, ghost rows hold exactly
    the transposes of owned entries, send/recv lists match pairwise."""
    full = offline.SyntheticOffline(offline.mach3_step_2d(10))
    parts = [offline.SyntheticOffline(offline.mach3_step_2d(10, n_ranks=n_ranks, rank=r))
             for r in range(n_ranks)]
    assert sum(p.n_owned for p in parts) == full.n_owned
    gid_full = {g: i for i, g in enumerate(full.global_ids.tolist())}
    all_owned = np.concatenate([p.global_ids[: p.n_owned] for p in parts])
    assert len(set(all_owned.tolist())) == full.n_owned
    for p in parts:
        assert abs(p.measure_of_omega - full.measure_of_omega) < 1e-12
        o = p.c.contents
        assert p.n_export <= p.n_internal <= p.n_owned <= p.n_relevant
        # lumped mass of owned and ghost nodes equals the global one
        for i, g in enumerate(p.global_ids.tolist()):
            assert abs(p.mi[i] - full.mi[gid_full[g]]) <= 1e-15 * p.mi[i]
        rs, cols = _rows(p)
        # owned rows: same stencil as the serial mesh (as global ids)
        frs, fcols = _rows(full)
        for i in range(0, p.n_owned, 7):
            mine = sorted(p.global_ids[cols[rs[i]:rs[i + 1]]].tolist())
            fi = gid_full[int(p.global_ids[i])]
            ref = sorted(full.global_ids[fcols[frs[fi]:frs[fi + 1]]].tolist())
            assert mine == ref
        # ghost rows: diagonal + owned columns only, each the transpose of an owned entry
        for i in range(p.n_owned, p.n_relevant):
            r = cols[rs[i]:rs[i + 1]]
            assert r[0] == i and np.all(r[1:] < p.n_owned)
            for j in r[1:]:
                assert i in cols[rs[j]:rs[j + 1]]
        n_nbr = o.n_nbr
        assert n_nbr == (1 if p.spec.rank in (0, n_ranks - 1) else 2)
    # pairwise matching of exchange lists
    for r in range(n_ranks - 1):
        a, b = parts[r], parts[r + 1]
        oa, ob = a.c.contents, b.c.contents
        ia = [oa.nbr_rank[q] for q in range(oa.n_nbr)].index(r + 1)
        ib = [ob.nbr_rank[q] for q in range(ob.n_nbr)].index(r)
        send_a = [oa.send_idx[q] for q in range(oa.send_off[ia], oa.send_off[ia + 1])]
        ghosts_b = list(range(ob.recv_off[ib], ob.recv_off[ib + 1]))
        assert a.global_ids[send_a].tolist() == b.global_ids[ghosts_b].tolist()
        send_b = [ob.send_idx[q] for q in range(ob.send_off[ib], ob.send_off[ib + 1])]
        ghosts_a = list(range(oa.recv_off[ia], oa.recv_off[ia + 1]))
        assert b.global_ids[send_b].tolist() == a.global_ids[ghosts_a].tolist()
        # matrix rows: entries sent by a == entries of b's ghost rows, in storage order
        rsb, colsb = _rows(b)
        sent = [(int(a.global_ids[oa.row_send_row[q]]),
                 int(a.global_ids[_rows(a)[1][_rows(a)[0][oa.row_send_row[q]] + oa.row_send_col[q]]]))
                for q in range(oa.row_send_off[ia], oa.row_send_off[ia + 1])]
        recv = [(int(b.global_ids[i]), int(b.global_ids[colsb[e]]))
                for i in ghosts_b for e in range(rsb[i], rsb[i + 1])]
        assert sent == recv

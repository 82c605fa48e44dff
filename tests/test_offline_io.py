"""OfflineData dumps (include/ryujin_offline_io.h): the wire format carries exactly the
arrays of `ryujin_hip_offline`, so a dump written on the deal.II side drives the hot path unchanged.
Checked here: lossless round trip (single rank, partitioned, shallow water with bathymetry), rejection of
damaged files, and -- through the oracle -- that a step on the imported data is bit-identical."""
import os

import numpy as np
import pytest

from ryujin_amd import HyperbolicModule, capi, offline
from ryujin_amd.initial_states import euler_uniform

ARRAYS = ("row_starts", "columns", "cij", "mij", "mi", "b_i", "b_id", "b_normal", "positions", "b_positions")


def _assert_same(a, b):
    for name in ("dim", "n_export", "n_internal", "n_owned", "n_relevant", "n_bdry", "n_pairs", "nnz",
                 "measure_of_omega"):
        assert getattr(a, name) == getattr(b, name), name
    for name in ARRAYS:
        x, y = getattr(a, name), getattr(b, name)
        assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), name
    for x, y in zip(a.pairs, b.pairs):
        assert np.array_equal(x, y)
    ca, cb = a.c.contents, b.c.contents
    assert ca.simd_length == cb.simd_length and ca.n_nbr == cb.n_nbr
    n_nbr = ca.n_nbr
    if n_nbr:
        for name, n in (("nbr_rank", n_nbr), ("send_off", n_nbr + 1), ("recv_off", n_nbr + 1),
                        ("row_send_off", n_nbr + 1)):
            assert [getattr(ca, name)[i] for i in range(n)] == [getattr(cb, name)[i] for i in range(n)], name
        n_send, n_row = ca.send_off[n_nbr], ca.row_send_off[n_nbr]
        assert [ca.send_idx[i] for i in range(n_send)] == [cb.send_idx[i] for i in range(n_send)]
        for name in ("row_send_row", "row_send_col"):
            assert [getattr(ca, name)[i] for i in range(n_row)] == [getattr(cb, name)[i] for i in range(n_row)]


def test_round_trip_single_rank_step_mesh(tmp_path):
    off = offline.SyntheticOffline(offline.mach3_step_2d(20))
    assert off.n_pairs > 0 and off.n_bdry > 0
    path = str(tmp_path / "step.ryjoffl")
    off.save(path)
    imp = offline.ImportedOffline(path)
    _assert_same(off, imp)
    assert imp.n_initial_precomputed == 0 and not imp.c.contents.initial_precomputed


@pytest.mark.parametrize("rank", [0, 1, 2])
def test_round_trip_partitioned(tmp_path, rank):
    off = offline.SyntheticOffline(offline.mach3_step_2d(10, n_ranks=3, rank=rank))
    assert off.c.contents.n_nbr >= 1 and off.n_relevant > off.n_owned
    path = str(tmp_path / f"rank{rank}.ryjoffl")
    off.save(path)
    _assert_same(off, offline.ImportedOffline(path))


def test_round_trip_3d_and_1d(tmp_path):
    for spec in (offline.box_3d(6), offline.MeshSpec(1, (33,), (0.0,), (1.0,), (capi.BC_DIRICHLET, capi.BC_DO_NOTHING))):
        off = offline.SyntheticOffline(spec)
        path = str(tmp_path / f"m{spec.dim}.ryjoffl")
        off.save(path)
        _assert_same(off, offline.ImportedOffline(path))


def test_round_trip_shallow_water_bathymetry(tmp_path):
    off = offline.SyntheticOffline(offline.rectangle_2d(12, (-1.0, -1.0), (1.0, 1.0)))
    Z = 0.1 * np.cos(3.0 * off.positions[:, 0]) * np.sin(2.0 * off.positions[:, 1])
    off.set_initial_precomputed(Z)
    path = str(tmp_path / "sw.ryjoffl")
    off.save(path)
    imp = offline.ImportedOffline(path)
    _assert_same(off, imp)
    assert imp.n_initial_precomputed == 1
    got = capi.np_from_ptr(imp.c.contents.initial_precomputed, imp.n_relevant, np.float64)
    assert np.array_equal(got, Z)


def test_damaged_files_are_rejected(tmp_path):
    off = offline.SyntheticOffline(offline.mach3_step_2d(10))
    path = str(tmp_path / "ok.ryjoffl")
    off.save(path)
    blob = open(path, "rb").read()

    def expect(data, what):
        p = str(tmp_path / "bad.ryjoffl")
        with open(p, "wb") as f:
            f.write(data)
        with pytest.raises(RuntimeError, match=what):
            offline.ImportedOffline(p)

    expect(b"", "truncated")
    expect(b"NOTADUMP" + blob[8:], "bad magic")
    expect(blob[:8] + (7).to_bytes(4, "little") + blob[12:], "unsupported version")
    expect(blob[: len(blob) // 2], "truncated|bytes, expected|exceed")
    expect(blob[:-8], "checksum missing")
    expect(blob + b"\0", "trailing bytes")
    flipped = bytearray(blob)
    flipped[len(blob) // 3] ^= 0x10  # inside a payload: sizes stay consistent, the checksum does not
    expect(bytes(flipped), "checksum mismatch|invalid OfflineData|bytes, expected")
    with pytest.raises(RuntimeError, match="cannot open"):
        offline.ImportedOffline(str(tmp_path / "does-not-exist"))


def test_structural_validation(tmp_path):
    """A dump with a valid checksum but inconsistent contents (a column index out of range, a row that
    does not start with its diagonal) is rejected by the reader's structural checks."""
    off = offline.SyntheticOffline(offline.rectangle_2d(6))
    lib = capi.load_synth()
    cols = off.columns.copy()
    o = off.c.contents

    def write_with(columns):
        import ctypes as C
        shadow = capi.Offline()
        C.memmove(C.byref(shadow), C.byref(o), C.sizeof(capi.Offline))
        keep = np.ascontiguousarray(columns, dtype=np.uint32)
        shadow.columns = capi.as_ptr(keep, capi.c_u32_p)
        p = str(tmp_path / "s.ryjoffl")
        assert lib.ryujin_offline_write(p.encode(), C.byref(shadow), 2, 0, None, None) == 0
        return p

    bad = cols.copy()
    bad[1] = off.n_relevant + 5
    with pytest.raises(RuntimeError, match="out of range"):
        offline.ImportedOffline(write_with(bad))
    bad = cols.copy()
    rs = off.row_starts
    bad[rs[3]], bad[rs[3] + 1] = cols[rs[3] + 1], cols[rs[3]]
    with pytest.raises(RuntimeError, match="diagonal"):
        offline.ImportedOffline(write_with(bad))
    imp = offline.ImportedOffline(write_with(cols))  # without positions
    with pytest.raises(ValueError):
        imp.positions


def test_oracle_step_on_imported_data_is_bit_identical(oracle, tmp_path):
    spec = offline.mach3_step_2d(20)
    off = offline.SyntheticOffline(spec)
    path = str(tmp_path / "step.ryjoffl")
    off.save(path)
    imp = offline.ImportedOffline(path)
    out = []
    for o in (off, imp):
        m = HyperbolicModule(o, equation=capi.EQ_EULER, backend=oracle.backend())
        m.cfl = 0.9
        U0 = euler_uniform(o.positions)
        U0 *= 1.0 + 1e-3 * np.sin(7.0 * o.positions[:, :1] + 3.0 * o.positions[:, 1:2])
        dirichlet = euler_uniform(o.b_positions)
        a, b = m.new_state_vector(U0), m.new_state_vector()
        taus = []
        for _ in range(3):
            m.prepare_state_vector(a, 0.0, dirichlet)
            taus.append(m.step(a, [], [], b))
            a, b = b, a
        out.append((a.download(), taus))
    assert out[0][1] == out[1][1]
    assert np.array_equal(out[0][0], out[1][0])
    assert os.path.getsize(path) > 0


def test_round_trip_unstructured_arbitrary_partition(tmp_path):
    """Dumps of a mesh the generator cannot produce: P1 triangles on a disk, cut into 4 sectors (ranks with
    three neighbours, nodes exported to several ranks, ragged rows): every array of every rank survives the
    round trip through the file format bit for bit."""
    from helpers_unstructured import disk_points, p1_offline, partition
    from test_oracle_unstructured import sector_owner
    off, info = p1_offline(disk_points(10))
    views = partition(off, info, sector_owner(off.positions, 4))
    lib = capi.load_synth()
    for r, v in enumerate(views):
        path = str(tmp_path / f"disk{r}.ryjoffl")
        pos = np.ascontiguousarray(v.positions)
        bpos = np.ascontiguousarray(v.b_positions)
        rc = lib.ryujin_offline_write(path.encode(), v.c, 2, 0, capi.as_ptr(pos, capi.c_double_p),
                                      capi.as_ptr(bpos, capi.c_double_p))
        assert rc == 0, lib.ryujin_offline_io_last_error()
        imp = offline.ImportedOffline(path)
        a, b = v._o, imp.c.contents
        for name in ("n_export", "n_internal", "n_owned", "n_relevant", "simd_length", "n_bdry", "n_pairs",
                     "n_nbr", "measure_of_omega"):
            assert getattr(a, name) == getattr(b, name), name
        nnz = int(v._keep["row_starts"][-1])
        n_nbr = a.n_nbr
        assert n_nbr >= 2
        sizes = dict(row_starts=v.n_relevant + 1, columns=nnz, cij=2 * nnz, mij=nnz, mi=v.n_relevant,
                     mi_inv=v.n_relevant, b_i=v.n_bdry, b_normal=2 * v.n_bdry, b_id=v.n_bdry, p_i=v.n_pairs,
                     p_col=v.n_pairs, p_j=v.n_pairs, nbr_rank=n_nbr, send_off=n_nbr + 1,
                     send_idx=int(a.send_off[n_nbr]), recv_off=n_nbr + 1, row_send_off=n_nbr + 1,
                     row_send_row=int(a.row_send_off[n_nbr]), row_send_col=int(a.row_send_off[n_nbr]))
        for name, n in sizes.items():
            x, y = getattr(a, name), getattr(b, name)
            assert [x[i] for i in range(n)] == [y[i] for i in range(n)], (r, name)
        assert np.array_equal(imp.positions, pos) and np.array_equal(imp.b_positions, bpos)

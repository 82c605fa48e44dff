"""SURVEY.md section 8 f-2, the exporter EXECUTED: contrib/ryujin_export_offline.h walks the reference's own
SparsityPatternSIMD / SparseMatrixSIMD (rows [0, n_internal) SIMD-interleaved, read back through get_entry / get_tensor)
and the OfflineData accessors inside tests/cpp/time_integrator_run.cc (the reference's classes from a patched temporary
copy of its tree; deal.II and the assembly of OfflineData are stand-ins fed by the synthetic generator) and writes a
dump; the dump is imported (include/ryujin_offline_io.h) and

  CPU: compared array by array with what the generator produced, and the ORACLE run on the dump reproduces the oracle
       run on the generator's arrays;
  GPU: one update of the HIP path on the imported dump against the oracle on the same dump, sweep by sweep
       (tests/helpers_parity.py) -- f-2's GPU tests no longer compare HIP with HIP.

Round 5 had the exporter type checked only. What this still cannot show is deal.II's own OfflineData (curved
boundaries, Cuthill-McKee numbering): not buildable here."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers_reference_tree as reftree  # noqa: E402
import test_binding_run as tbr  # noqa: E402

from ryujin_amd import HyperbolicModule, capi, offline  # noqa: E402
from ryujin_amd.initial_states import euler_uniform  # noqa: E402


def _export(exe_name, tmp_path, mesh):
    prefix = str(tmp_path / "exported")
    out = subprocess.run([tbr.EXE[exe_name], "export", prefix, mesh], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "exported" in out.stdout
    return prefix + "-0.ryjoffl"


def _rows(o):
    """{row: {column: (m_ij, c_ij tuple)}} of offline data in plain CSR"""
    rs, cols = o.row_starts.astype(np.int64), o.columns.astype(np.int64)
    cij = np.asarray(o.cij).reshape(-1, o.dim)
    mij = np.asarray(o.mij)
    return [{int(cols[e]): (float(mij[e]), tuple(cij[e])) for e in range(rs[i], rs[i + 1])} for i in range(o.n_owned)]


@pytest.mark.skipif(not reftree.available(), reason="needs the reference tree, g++ and patch (the build container)")
def test_exporter_runs_on_the_reference_containers_and_the_oracle_reads_its_dump(tmp_path, oracle):
    tbr.build_binaries(str(tmp_path), which=("oracle_double",))
    n = 20
    path = _export("oracle_double", tmp_path, f"step:{n}")
    imp = offline.ImportedOffline(path)
    gen = offline.SyntheticOffline(offline.mach3_step_2d(n))
    # ---- the dump against the generator's arrays: same numbers, the rows in the reference's column order
    for name in ("n_export", "n_internal", "n_owned", "n_relevant", "n_bdry", "dim"):
        assert getattr(imp, name) == getattr(gen, name), name
    assert imp.c.contents.simd_length in (0, 1)   # written as plain CSR: get_entry() undid the SIMD interleave
    rows_i, rows_g = _rows(imp), _rows(gen)
    assert rows_i == rows_g                      # bitwise: every (row, column) entry of m_ij and c_ij
    rs, cols = imp.row_starts.astype(np.int64), imp.columns.astype(np.int64)
    for i in range(imp.n_owned):                 # diagonal first, then ascending (dealii::SparsityPattern)
        assert cols[rs[i]] == i and (np.diff(cols[rs[i] + 1: rs[i + 1]]) > 0).all()
    assert np.array_equal(imp.mi, gen.mi)
    mi_inv = lambda o: np.ctypeslib.as_array(o.c.contents.mi_inv, shape=(o.n_relevant,))  # noqa: E731
    assert np.array_equal(mi_inv(imp), mi_inv(gen))
    assert np.array_equal(imp.b_i, gen.b_i) and np.array_equal(imp.b_id, gen.b_id)
    assert np.array_equal(imp.b_normal, gen.b_normal) and np.array_equal(imp.b_positions, gen.b_positions)
    assert np.array_equal(imp.positions, gen.positions)
    assert sorted(zip(*[x.tolist() for x in imp.pairs])) == sorted(zip(*[x.tolist() for x in gen.pairs]))
    assert imp.measure_of_omega == gen.measure_of_omega
    # ---- the oracle on the dump == the oracle on the generator's arrays (another column order: round-off)
    rng = np.random.default_rng(3)
    U0 = euler_uniform(gen.positions) * (1.0 + 1e-3 * rng.uniform(-1, 1, size=(gen.n_relevant, 4)))
    dirichlet = euler_uniform(gen.b_positions)
    res = []
    for o in (gen, imp):
        m = HyperbolicModule(o, equation=capi.EQ_EULER, backend=oracle.backend())
        m.cfl = 0.9
        a, b = m.new_state_vector(U0), m.new_state_vector()
        taus = []
        for _ in range(5):
            m.prepare_state_vector(a, 0.0, dirichlet)
            taus.append(m.step(a, [], [], b))
            a, b = b, a
        res.append((np.array(taus), a.download()[: o.n_owned]))
    assert np.abs(res[0][0] / res[1][0] - 1.0).max() < 1e-13
    assert (np.abs(res[0][1] - res[1][1]) / np.abs(res[0][1]).max(axis=0)).max() < 1e-12


@pytest.mark.gpu
def test_hip_on_an_exported_dump_against_the_oracle(tmp_path, oracle):
    """the dump written by the exporter (run HERE, by the shipped test binary) through the kernels, against the oracle"""
    from test_gpu_parity import _both, _compare_step, _perturbed
    if not os.path.exists(tbr.EXE["unmodified"]):
        if not reftree.available():
            pytest.skip("tests/cpp/time_integrator_run_* are built where the reference tree is (__graft_entry__.build())")
        tbr.build_binaries(str(tmp_path), which=("unmodified",))
    for mesh, n_warm in (("step:40", 12), ("box:24", 3)):
        path = _export("unmodified", tmp_path, mesh)
        imp = offline.ImportedOffline(path)
        U0 = _perturbed(euler_uniform(imp.positions))
        dirichlet = euler_uniform(imp.b_positions) if imp.n_bdry else None
        _, mods = _both(None, U0, oracle, n_warm=n_warm, dirichlet=dirichlet, off=imp)
        g, c = _compare_step(imp, mods, dirichlet)
        assert g["status"] == 0

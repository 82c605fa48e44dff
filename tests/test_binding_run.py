"""The drop-in boundary EXECUTED: ryujin's own TimeIntegrator -- step(), step_ssprk_33(),
step_erk_33() ..., the host-side sadd() and StateVector::swap() -- runs on top of contrib/hyperbolic_module_hip.h and
reproduces the reference's tests/euler/check-mass-conservation_01 baseline.

tests/cpp/time_integrator_run.cc says what is real (the reference's time integrator, state vectors, SIMD sparsity
pattern and matrices from a patched temporary COPY of the reference tree; the adapter; the library) and what is a
stand-in (deal.II: tests/cpp/dealii_mock with one-rank behaviour; the three collaborators that need deal.II's grid/FE
stack). Two builds of the same source:

  time_integrator_run_unmodified   contrib/hyperbolic_module_hip.patch ALONE: nothing but hyperbolic_module.h is
                                   touched, TimeIntegrator::step calls prepare_state_vector/step<s>/sadd/swap;
  time_integrator_run_patched      + contrib/time_integrator_hip.patch: TimeIntegrator::step -> time_step().

CPU leg (this container, no GPU): the unmodified build linked against tests/cpp/hip_abi_on_oracle.cc, a test double
that forwards the ABI subset the adapter uses to the CPU oracle -- the adapter's host logic (twins that follow the
storage through swap(), what is uploaded and written back when) against the golden.
GPU leg: both builds linked against libryujin_hip.so; every combination of {unmodified, patched} x {hip device
resident state vectors = false, true} reproduces the golden and all four agree bit for bit.

The binaries are built where the reference tree is (this container; __graft_entry__.build() builds them) and travel
to the GPU box with the snapshot like the other in-tree artefacts."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers_reference_tree as reftree  # noqa: E402

from ryujin_amd import _build  # noqa: E402

CPP = os.path.join(ROOT, "tests", "cpp")
EXE = {name: os.path.join(CPP, "time_integrator_run_" + name) for name in ("unmodified", "patched", "oracle_double")}


def build_binaries(tmp_root: str, which=("unmodified", "patched", "oracle_double")) -> None:
    """g++ on tests/cpp/time_integrator_run.cc + the reference's multicomponent_vector.cc inside patched copies of the
    reference tree. Needs /root/reference (this container)."""
    _build.build_synth()
    trees = {}
    for name in which:
        patches = reftree.PATCHES if name == "patched" else ("hyperbolic_module_hip.patch",)
        key = "all" if name == "patched" else "one"
        if key not in trees:
            trees[key] = reftree.make_patched_tree(os.path.join(tmp_root, key), patches)
        src = trees[key]
        cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-DRYUJIN_WITH_HIP", "-I" + reftree.MOCK, "-I" + src,
               "-I" + os.path.join(src, "euler"), "-I" + _build.INCLUDE,
               os.path.join(CPP, "time_integrator_run.cc"), os.path.join(src, "multicomponent_vector.cc")]
        if name == "patched":
            cmd += ["-DRYUJIN_TEST_PATCHED_TIME_LOOP"]
        if name == "oracle_double":
            from build_oracle import build_oracle
            oracle_so = build_oracle()
            cmd += [os.path.join(CPP, "hip_abi_on_oracle.cc"), "-L" + os.path.dirname(oracle_so), "-lryujin_oracle",
                    "-Wl,-rpath," + os.path.dirname(oracle_so)]
        else:
            cmd += ["-L" + _build.LIBDIR, "-lryujin_hip", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
        cmd += ["-L" + _build.LIBDIR, "-lryujin_synth", "-Wl,-rpath," + _build.LIBDIR, "-o", EXE[name]]
        res = subprocess.run(cmd, capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-6000:]


def run(name: str, scheme: str, n_steps: int, device_resident: int, env=None):
    out = subprocess.run([EXE[name], scheme, str(n_steps), str(device_resident)], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, **(env or {})))
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    series = np.array([[float(x) for x in ln.split()] for ln in lines if ln[0].isdigit()])
    return series, out.stdout


def golden(golden_dir):
    from test_oracle_golden_integration import golden_mass_conservation
    return golden_mass_conservation(golden_dir)


@pytest.mark.skipif(not reftree.available(), reason="needs the reference tree, g++ and patch (the build container)")
def test_unmodified_time_integrator_on_the_adapter_cpu(tmp_path, golden_dir):
    """the reference's unmodified step_ssprk_33 (host sadd + swap) on the adapter, oracle-backed test double"""
    build_binaries(str(tmp_path), which=("oracle_double",))
    gold = golden(golden_dir)
    series, text = run("oracle_double", "ssprk33", 6, 0)
    np.testing.assert_allclose(series[:, 0], gold[1:7, 0], rtol=0, atol=5e-14)
    np.testing.assert_allclose(series[:, 1], gold[1:7, 1], rtol=0, atol=1e-13)
    assert "n_restarts 0 n_warnings 0" in text
    # the run-time parameter that used to break the unmodified caller (twins keyed by the address of the StateVector
    # object, uploads skipped): it changes NOTHING for calls that do not come through time_step()
    for scheme in ("ssprk33", "erk11", "erk33", "erk54"):
        a, b = run("oracle_double", scheme, 3, 0)[1], run("oracle_double", scheme, 3, 1)[1]
        assert a == b, scheme
    # erk 11 ends with state_vector.swap(temp_[0]) and nothing else: a twin that stays with the object instead of
    # the storage recomputes the first step forever -- the time axis has to advance by different tau
    series, _ = run("oracle_double", "erk11", 4, 1)
    assert len(set(np.round(np.diff(series[:, 0]), 12))) > 1
    assert np.all(np.abs(series[:, 1] - 1.4) < 1e-13)


@pytest.mark.skipif(not reftree.available(), reason="needs the reference tree, g++ and patch (the build container)")
def test_binaries_for_the_gpu_leg_build(tmp_path):
    build_binaries(str(tmp_path), which=("unmodified", "patched"))
    for name in ("unmodified", "patched"):
        assert os.path.exists(EXE[name])
    # the patched build routes TimeIntegrator::step to the adapter's time_step(); the unmodified one cannot
    syms = {name: subprocess.run(["nm", "-C", EXE[name]], capture_output=True, text=True).stdout for name in EXE
            if name != "oracle_double"}
    assert "ryujin_hip_time_step_fn" in syms["patched"]


@pytest.mark.gpu
def test_reference_time_integrator_runs_on_the_adapter_gpu(golden_dir):
    for name in ("unmodified", "patched"):
        if not os.path.exists(EXE[name]):
            if not reftree.available():
                pytest.skip("tests/cpp/time_integrator_run_* are built where the reference tree is "
                            "(__graft_entry__.build()); they are missing from this snapshot")
            import tempfile
            build_binaries(tempfile.mkdtemp(), which=("unmodified", "patched"))
    gold = golden(golden_dir)
    texts = {}
    for name in ("unmodified", "patched"):
        for device_resident in (0, 1):
            series, text = run(name, "ssprk33", 6, device_resident)
            np.testing.assert_allclose(series[:, 0], gold[1:7, 0], rtol=0, atol=1e-12, err_msg=f"{name} {device_resident}")
            np.testing.assert_allclose(series[:, 1], gold[1:7, 1], rtol=0, atol=1e-12, err_msg=f"{name} {device_resident}")
            assert "n_restarts 0 n_warnings 0" in text
            texts[name, device_resident] = text
    # host-mirrored stage by stage (host sadd, swap) = one call of the device-resident driver, bit for bit (time axis,
    # mean density and the checksum of the final U as it arrives in the HOST vector)
    assert len(set(texts.values())) == 1, texts
    # the caller's vectors pinned in place (hip pin host vectors), and without the derived vectors
    assert run("unmodified", "ssprk33", 6, 0, {"RYUJIN_TEST_PIN": "1"})[1] == texts["unmodified", 0]
    assert run("patched", "ssprk33", 6, 1, {"RYUJIN_TEST_PIN": "1", "RYUJIN_TEST_NO_DERIVED": "1"})[1] == texts["unmodified", 0]
    # the multi-stage schemes: stage vectors and weights through step<1>, step<2> against the library's driver
    for scheme in ("erk33", "erk11", "erk54", "ssprk22"):
        outs = {(name, dr): run(name, scheme, 3, dr)[1] for name in ("unmodified", "patched") for dr in (0, 1)}
        assert len(set(outs.values())) == 1, (scheme, outs)


@pytest.mark.gpu
def test_host_mirroring_entry_points_write_back_only_what_the_call_changes(oracle):
    """ryujin_hip_state_download_prepared / _owned / host_register against full downloads"""
    from ryujin_amd import HyperbolicModule, capi, offline
    from ryujin_amd.initial_states import euler_uniform
    from ryujin_amd.module import HostMirroredModule, HostStateVector
    spec = offline.mach3_step_2d(25)
    off = offline.SyntheticOffline(spec)
    rng = np.random.default_rng(7)
    U0 = euler_uniform(off.positions) * (1.0 + 1e-3 * rng.uniform(-1, 1, size=(off.n_relevant, 4)))
    dirichlet = euler_uniform(off.b_positions)
    p = oracle.default_params(capi.EQ_EULER, 2)
    p.cfl = 0.9
    m = HyperbolicModule(off, p, backend="hip")
    # reference: device-resident handles
    old, new = m.new_state_vector(U0), m.new_state_vector()
    m.prepare_state_vector(old, 0.0, dirichlet)
    U_prepared, prec = old.download(), old.download_precomputed()
    tau_ref = m.step(old, [], [], new)
    U_new = new.download()
    for pin in (True, False):
        hm = HostMirroredModule(m, pin=pin)
        a, b = HostStateVector(m, U0), HostStateVector(m)
        b.U[:] = -7.0  # sentinel
        hm.prepare_state_vector(a, 0.0, dirichlet)
        assert np.array_equal(a.U, U_prepared) and np.array_equal(a.precomputed, prec)
        changed = np.flatnonzero((U_prepared != U0).any(axis=1))
        assert set(changed) <= set(off.b_i.tolist())  # only boundary rows differ from what was uploaded
        assert hm.step(a, [], [], b) == tau_ref
        assert np.array_equal(b.U[: off.n_owned], U_new[: off.n_owned])
        assert np.array_equal(hm.alpha, m.alpha())
        # write-back of prepare touches nothing but the boundary rows (and the ghost range: none on one rank)
        a.U[:] = U0
        a.U[~np.isin(np.arange(off.n_relevant), off.b_i)] = 123.0
        m._check(m._lib.ryujin_hip_state_download_prepared(m._ctx, hm.twin_of(a).handle,
                                                           capi.as_ptr(a.U, capi.c_double_p)))
        interior = ~np.isin(np.arange(off.n_relevant), off.b_i)
        assert (a.U[interior] == 123.0).all() and np.array_equal(a.U[~interior], U_prepared[~interior])
        # SSPRK33 as the unmodified caller drives it (host sadd + swap) against the device-resident driver
        state, temp = HostStateVector(m, U0), [HostStateVector(m), HostStateVector(m)]
        t = 0.0
        for _ in range(2):
            hm.prepare_state_vector(state, t, dirichlet)
            tau = hm.step(state, [], [], temp[0])
            hm.prepare_state_vector(temp[0], t + tau, dirichlet)
            hm.step(temp[0], [], [], temp[1], tau)
            temp[1].U[: off.n_owned] = 0.25 * temp[1].U[: off.n_owned] + 0.75 * state.U[: off.n_owned]
            hm.prepare_state_vector(temp[1], t + 0.5 * tau, dirichlet)
            hm.step(temp[1], [], [], temp[0], tau)
            temp[0].U[: off.n_owned] = 2.0 / 3.0 * temp[0].U[: off.n_owned] + 1.0 / 3.0 * state.U[: off.n_owned]
            state.swap(temp[0])
            t += tau
        sv, T3 = m.new_state_vector(U0), [m.new_state_vector() for _ in range(3)]
        t_dev = 0.0
        for _ in range(2):
            t_dev += m.time_step("ssprk 33", sv, T3, dirichlet)
        assert t == t_dev
        assert np.array_equal(state.U[: off.n_owned], sv.download()[: off.n_owned])
        hm.close()
    # registering the same range twice is fine, unregistering an unknown one is a warning, not an error
    x = np.zeros(4096)
    assert m._lib.ryujin_hip_host_register(m._ctx, x.ctypes.data, x.nbytes) == capi.RYUJIN_OK
    assert m._lib.ryujin_hip_host_register(m._ctx, x.ctypes.data, x.nbytes) == capi.RYUJIN_OK
    assert m._lib.ryujin_hip_host_unregister(m._ctx, x.ctypes.data) == capi.RYUJIN_OK
    assert m._lib.ryujin_hip_host_unregister(m._ctx, x.ctypes.data) == capi.RYUJIN_WARN


@pytest.mark.gpu
def test_download_prepared_on_a_middle_rank_sees_the_boundary_conditions_of_export_rows(oracle):
    """with neighbours and a mesh small enough for the boundary conditions to ride on the pre-pass
    (fold_bc), the boundary rows inside EXPORT slices get their boundary values from the export part of the pre-pass on
    the exchange stream. ryujin_hip_state_download_prepared packs the boundary rows on the compute stream: it has to
    join the exchange stream first. A middle rank of a slab partition (loopback communicator), repeated so that a
    missing join would show as a stale row sooner or later: the packed write-back equals the full download."""
    import ctypes as C

    from ryujin_amd import HyperbolicModule, capi, offline
    from ryujin_amd.initial_states import euler_uniform
    spec = offline.mach3_step_2d(60, n_ranks=3, rank=1)
    off = offline.SyntheticOffline(spec)
    assert off.n_bdry > 0 and off.n_relevant > off.n_owned
    rng = np.random.default_rng(11)
    dirichlet = euler_uniform(off.b_positions)
    p = oracle.default_params(capi.EQ_EULER, 2)
    p.cfl = 0.9
    comm = C.c_void_p()
    assert capi.load_hip().ryujin_hip_comm_init_loopback(C.byref(comm), 1, 3, 0) == 0
    m = HyperbolicModule(off, p, backend="hip", comm=comm)
    sv = m.new_state_vector()
    boundary = np.isin(np.arange(off.n_relevant), off.b_i)
    ghost = np.arange(off.n_relevant) >= off.n_owned
    for rep in range(40):
        # momentum with a wall-normal part on every boundary row: slip rows MUST change in prepare_state_vector
        U0 = euler_uniform(off.positions) * (1.0 + 1e-2 * rng.uniform(-1, 1, size=(off.n_relevant, 4)))
        U0[:, 2] += 0.3
        sv.upload(U0)
        m.prepare_state_vector(sv, 0.0, dirichlet)
        got = np.full_like(U0, 123.0)
        m._check(m._lib.ryujin_hip_state_download_prepared(m._ctx, sv.handle, capi.as_ptr(got, capi.c_double_p)))
        full = sv.download()
        assert (full[boundary & ~ghost] != U0[boundary & ~ghost]).any()
        touched = boundary | ghost
        assert np.array_equal(got[touched], full[touched]), rep
        assert (got[~touched] == 123.0).all()
    m.close()

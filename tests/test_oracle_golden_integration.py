"""Pin the WHOLE hot path of the CPU oracle (prepare_state_vector + step + SSPRK33/ERK33 driver on
a synthetic Cartesian Q1 mesh) against the reference's integration-test baselines:

 * tests/euler/check-mass-conservation_01.{prm,output}: 64^2 cells on [0,20]^2, slip walls, uniform
   Mach-3 state, SSPRK33 at cfl 0.9; the reference prints the mass-weighted mean primitive state and
   its second moments after every step with 15 digits.
 * tests/euler/verification-isentropic_vortex-2d-{ssprk33,erk33}-l5.{prm,output}: 32^2 cells on
   [-5,5]^2, Dirichlet data = exact solution, cfl 0.2, final time and normalised error norms.

The meshes are Cartesian, so our closed-form Q1 stencils coincide with deal.II's assembly; the only
differences are the local numbering (stencil summation order) and std::pow vs vcl::pow.
"""
import os
import re

import numpy as np
import pytest

from ryujin_amd import HyperbolicModule, TimeIntegrator, capi, offline
from ryujin_amd.initial_states import euler_isentropic_vortex, euler_uniform


def _space_average(U, mi, gamma=1.4):
    """Quantities::internal_accumulate (source/quantities.template.h:370-420)."""
    rho = U[:, 0]
    v = U[:, 1:3] / rho[:, None]
    p = (gamma - 1.0) * (U[:, 3] - 0.5 * (U[:, 1] ** 2 + U[:, 2] ** 2) / rho)
    prim = np.column_stack([rho, v, p])
    w = mi / mi.sum()
    return (w[:, None] * prim).sum(0), (w[:, None] * prim ** 2).sum(0)


def run_mass_conservation(backend, n_steps=None):
    off = offline.SyntheticOffline(offline.rectangle_2d(64, (0.0, 0.0), (20.0, 20.0)))
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend=backend)
    sv = m.new_state_vector(euler_uniform(off.positions))
    ti = TimeIntegrator(m, "ssprk 33", cfl_min=0.9, cfl_max=0.9, cfl_recovery_strategy="none")
    rows, t = [], 0.0
    mi = off.mi[: off.n_owned]
    n_steps = 18 if n_steps is None else n_steps
    for _ in range(n_steps + 1):
        a, b = _space_average(sv.download()[: off.n_owned], mi)
        rows.append(np.concatenate([[t], a, b]))
        sv, tau = ti.step(sv, t)
        t += tau
    return np.array(rows), m


def golden_mass_conservation(golden_dir):
    path = os.path.join(golden_dir, "euler_check-mass-conservation_01.output")
    return np.array([[float(x) for x in line.split()] for line in open(path) if line[0].isdigit()])


def test_mass_conservation_01_golden(oracle, golden_dir):
    gold = golden_mass_conservation(golden_dir)
    assert gold.shape == (19, 9)
    got, m = run_mass_conservation(oracle.backend())
    assert m.n_warnings() == 0 and m.n_restarts() == 0
    # time axis: pins every tau_max (d_ij diagonal, CFL) of 18 x 3 Euler steps
    np.testing.assert_allclose(got[:, 0], gold[:, 0], rtol=0, atol=5e-14)
    # mean primitive state (printed with 15 significant digits)
    np.testing.assert_allclose(got[:, [1, 2, 4]], gold[:, [1, 2, 4]], rtol=0, atol=1e-13)
    np.testing.assert_allclose(got[:, 3], gold[:, 3], rtol=0, atol=1e-15)  # mean v_2 ~ 1e-17
    # second moments
    np.testing.assert_allclose(got[:, [5, 6, 8]], gold[:, [5, 6, 8]], rtol=2e-14, atol=0)
    np.testing.assert_allclose(got[:, 7], gold[:, 7], rtol=1e-7, atol=1e-18)  # 1e-9 .. 1e-5 values
    # the conservation statement itself: mean density stays 1.4 to round-off
    assert np.abs(got[:, 1] - 1.4).max() < 1e-13


def _gauss3():
    x = np.array([-np.sqrt(3.0 / 5.0), 0.0, np.sqrt(3.0 / 5.0)]) * 0.5 + 0.5
    w = np.array([5.0, 8.0, 5.0]) / 18.0
    return x, w


def _cell_norms(values_grid, h):
    """L1 and L2 norm of the Q1 interpolant of nodal values on a uniform grid, QGauss<2>(3) per cell
    (VectorTools::integrate_difference as used in time_loop.template.h:741-795)."""
    x, w = _gauss3()
    v00, v10 = values_grid[:-1, :-1], values_grid[1:, :-1]
    v01, v11 = values_grid[:-1, 1:], values_grid[1:, 1:]
    l1 = 0.0
    l2 = 0.0
    for a, wa in zip(x, w):
        for b, wb in zip(x, w):
            val = v00 * (1 - a) * (1 - b) + v10 * a * (1 - b) + v01 * (1 - a) * b + v11 * a * b
            l1 += wa * wb * np.abs(val).sum()
            l2 += wa * wb * (val ** 2).sum()
    return l1 * h * h, np.sqrt(l2 * h * h)


def run_isentropic_vortex(backend, scheme, refinement=5, t_final=2.0, equation=capi.EQ_EULER):
    n = 2 ** refinement
    off = offline.SyntheticOffline(offline.rectangle_2d(n, (-5.0, -5.0), (5.0, 5.0), bc=capi.BC_DIRICHLET))
    m = HyperbolicModule(off, equation=equation, backend=backend)
    exact = lambda pos, t: euler_isentropic_vortex(pos, t, mach=1.0, beta=5.0)  # noqa: E731
    sv = m.new_state_vector(exact(off.positions, 0.0))
    bpos = off.b_positions
    ti = TimeIntegrator(m, scheme, cfl_min=0.2, cfl_max=0.2, cfl_recovery_strategy="none",
                        dirichlet_fn=lambda t: exact(bpos, t))
    t = 0.0
    while t < t_final:
        sv, tau = ti.step(sv, t)
        t += tau
    # compute_error (time_loop.template.h:694-833)
    m.prepare_state_vector(sv, t, exact(bpos, t))
    U = sv.download()
    A = exact(off.positions, t)
    h = 10.0 / n
    order = np.lexsort((off.positions[:, 0], off.positions[:, 1]))
    linf = l1 = l2 = 0.0
    for c in range(4):
        a = A[order, c].reshape(n + 1, n + 1).T
        e = (U[order, c] - A[order, c]).reshape(n + 1, n + 1).T
        l1a, l2a = _cell_norms(a, h)
        l1e, l2e = _cell_norms(e, h)
        linf += np.abs(e).max() / np.abs(a).max()
        l1 += l1e / l1a
        l2 += l2e / l2a
    return t, linf, l1, l2, off.n_owned


def _golden_vortex(golden_dir, scheme, level, prefix="euler_verification-isentropic_vortex-2d"):
    # the l7 baselines of the reference are MPI runs (mpirun=4 and mpirun=8, which differ by 8e-11 in Linf)
    name = f"{prefix}-{scheme.replace(' ', '')}-l{level}" + (".mpirun4.output" if level == 7 else ".output")
    text = open(os.path.join(golden_dir, name)).read()
    g = lambda k: float(re.search(k + r"\s*=\s*([0-9.e+-]+)", text).group(1))  # noqa: E731
    return int(g("#dofs")), g("t    "), g("Linf "), g("L1   "), g("L2   ")


@pytest.mark.parametrize("scheme", ["ssprk 33", "erk 33"])
def test_isentropic_vortex_l5_golden(oracle, golden_dir, scheme):
    """Exercises Dirichlet BCs and, for ERK33, the multi-stage step<1>/step<2> path with stage
    weights {-1} and {0.75,-2} (time_integrator.template.h:373-403)."""
    dofs, t_ref, linf_ref, l1_ref, l2_ref = _golden_vortex(golden_dir, scheme, 5)
    t, linf, l1, l2, n = run_isentropic_vortex(oracle.backend(), scheme, 5)
    assert n == dofs == 1089
    assert abs(t - t_ref) < 1e-11           # final time pins all time-step sizes
    assert abs(linf - linf_ref) < 1e-9 * linf_ref + 1e-12
    assert abs(l1 - l1_ref) < 1e-9 * l1_ref + 1e-12
    assert abs(l2 - l2_ref) < 1e-9 * l2_ref + 1e-12


@pytest.mark.parametrize("scheme", ["ssprk 33", "erk 33"])
def test_aeos_isentropic_vortex_l5_golden(oracle, golden_dir, scheme):
    """tests/euler_aeos/verification-isentropic_vortex-pge-2d-{ssprk33,erk33}-l5: the EulerAEOS
    Description (polytropic gas EOS, compute strict bounds = true) on the same configuration: pins the
    two precomputation cycles, the surrogate-gamma Riemann solver, indicator and limiter of
    source/euler_aeos/ through a whole run."""
    prefix = "euler_aeos_verification-isentropic_vortex-pge-2d"
    dofs, t_ref, linf_ref, l1_ref, l2_ref = _golden_vortex(golden_dir, scheme, 5, prefix)
    t, linf, l1, l2, n = run_isentropic_vortex(oracle.backend(), scheme, 5, equation=capi.EQ_EULER_AEOS)
    assert n == dofs == 1089
    assert abs(t - t_ref) < 1e-11
    assert abs(linf - linf_ref) < 1e-9 * linf_ref + 1e-12
    assert abs(l1 - l1_ref) < 1e-9 * l1_ref + 1e-12
    assert abs(l2 - l2_ref) < 1e-9 * l2_ref + 1e-12


FINE_VORTEX_CASES = [("euler", "ssprk 33", 6), ("euler", "erk 33", 6), ("euler", "erk 33", 7),
                     ("euler_aeos", "ssprk 33", 6), ("euler_aeos", "erk 33", 6), ("euler_aeos", "erk 33", 7)]


def check_fine_vortex(backend, golden_dir, description, scheme, level, rtol=2e-10):
    """Levels 6 and 7 (64^2 and 128^2 cells; observed: t to 1e-15, norms to 2.4e-12 and 3e-11; the
    reference's own mpirun=4 and mpirun=8 baselines at l7 differ by 8e-11)."""
    eq, prefix = {"euler": (capi.EQ_EULER, "euler_verification-isentropic_vortex-2d"),
                  "euler_aeos": (capi.EQ_EULER_AEOS, "euler_aeos_verification-isentropic_vortex-pge-2d")}[description]
    dofs, t_ref, linf_ref, l1_ref, l2_ref = _golden_vortex(golden_dir, scheme, level, prefix)
    t, linf, l1, l2, n = run_isentropic_vortex(backend, scheme, level, equation=eq)
    assert n == dofs == (2 ** level + 1) ** 2
    assert abs(t - t_ref) < 1e-13 * t_ref
    assert abs(linf - linf_ref) < rtol * linf_ref
    assert abs(l1 - l1_ref) < rtol * l1_ref
    assert abs(l2 - l2_ref) < rtol * l2_ref


@pytest.mark.parametrize("description,scheme,level", FINE_VORTEX_CASES)
def test_isentropic_vortex_fine_golden(oracle, golden_dir, description, scheme, level):
    check_fine_vortex(oracle.backend(), golden_dir, description, scheme, level)

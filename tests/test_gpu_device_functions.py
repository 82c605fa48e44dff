"""The DEVICE implementations of the Riemann solvers and the convex limiter, run on the reference's own unit-test
inputs and compared with the reference-held baselines directly:

  tests/euler/riemann_solver.cc:79-98 + riemann_solver{,-iterated-2,-iterated-10}.output  (10 states x 3 Newton settings)
  tests/euler/limiter.cc:61-139 + limiter.output                                          (12 cases)
  tests/shallow_water/riemann_solver.cc:75-77 + riemann_solver.output                     (3 states, incl. dry)

through ryujin_hip_debug_function (one thread per item, the same inlined device functions the sweeps call).
Tolerance: 1e-13 relative at function level (the reference's own std::pow / vcl::pow
builds differ in the last digits); the iterated Riemann baselines are printed with 16 decimals.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ryujin_amd import capi
from test_oracle_golden_euler import LIMITER_CASES, RIEMANN_CASES, _blocks, _grab, _limiter_blocks, _riemann_data

pytestmark = pytest.mark.gpu


def _device(params, which, items, n_out):
    lib = capi.load_hip()
    a = np.ascontiguousarray(np.array(items, dtype=np.float64))
    out = np.zeros((a.shape[0], n_out))
    rc = lib.ryujin_hip_debug_function(0, C.byref(params), which, capi.as_ptr(a, capi.c_double_p),
                                       capi.as_ptr(out, capi.c_double_p), a.shape[0])
    assert rc == 0, lib.ryujin_hip_last_error()
    return out


@pytest.mark.parametrize("n_newton,golden", [(0, "euler_riemann_solver.output"),
                                             (2, "euler_riemann_solver-iterated-2.output"),
                                             (10, "euler_riemann_solver-iterated-10.output")])
def test_device_riemann_solver_against_the_reference_baselines(oracle, golden_dir, n_newton, golden):
    params = oracle.default_params(capi.EQ_EULER, 1)
    params.riemann_newton_max_iterations = n_newton
    blocks = _blocks(os.path.join(golden_dir, golden))
    assert len(blocks) == len(RIEMANN_CASES) == 10
    items = [np.concatenate([_riemann_data(left, params.gamma), _riemann_data(right, params.gamma)])
             for left, right in RIEMANN_CASES]
    lam = _device(params, capi.DEBUG_EULER_RIEMANN, items, 1)[:, 0]
    for got, block in zip(lam, blocks):
        ref = _grab(block, "-> lambda_max =")[0]
        if n_newton == 0:
            assert abs(got - ref) <= 1e-13 * abs(ref), (got, ref)
        else:   # printed with 16 decimals (not 17 significant digits)
            assert abs(got - ref) <= 1e-12 * abs(ref) + 1e-15, (got, ref)
    if n_newton == 0:   # the values SURVEY.md 8c quotes
        expected = [1.5084890784907763, 1.7620896140769147, 2.6335650740600323, 34.018686867258801,
                    12.617757915202823, 11.832159566199232, 10.832159566199232, 9.7758781271580943,
                    6.6963146691962327, 9.7758781271580943]
        np.testing.assert_allclose(lam, expected, rtol=1e-13, atol=0)


def test_production_riemann_path_against_the_reference_baselines(oracle, golden_dir):
    """The evaluation path the sweeps RUN by default -- per-node Riemann records, dij_from_records<false>: no pow,
    another operation order than the reference (euler_device.hpp) -- on the reference's own ten states
    (tests/euler/riemann_solver.cc:79-98, incl. the Leblanc state with p_R = 6.7e-11 and pressure ratios of 1e9)
    against riemann_solver.output: 1e-12 relative, the d_ij contract."""
    params = oracle.default_params(capi.EQ_EULER, 1)
    assert params.riemann_newton_max_iterations == 0     # the default configuration = the records fast path
    blocks = _blocks(os.path.join(golden_dir, "euler_riemann_solver.output"))
    items = [np.concatenate([_riemann_data(left, params.gamma), _riemann_data(right, params.gamma)])
             for left, right in RIEMANN_CASES]
    lam = _device(params, capi.DEBUG_EULER_RIEMANN_RECORDS, items, 1)[:, 0]
    for got, block in zip(lam, blocks):
        ref = _grab(block, "-> lambda_max =")[0]
        assert abs(got - ref) <= 1e-12 * abs(ref), (got, ref)
    # and with Newton iterations switched on the records path hands over to the reference's riemann_compute
    for n_newton, golden in ((2, "euler_riemann_solver-iterated-2.output"), (10, "euler_riemann_solver-iterated-10.output")):
        params.riemann_newton_max_iterations = n_newton
        lam = _device(params, capi.DEBUG_EULER_RIEMANN_RECORDS, items, 1)[:, 0]
        for got, block in zip(lam, _blocks(os.path.join(golden_dir, golden))):
            ref = _grab(block, "-> lambda_max =")[0]
            assert abs(got - ref) <= 1e-12 * abs(ref) + 1e-15, (n_newton, got, ref)


def test_device_limiter_against_the_reference_baseline(oracle, golden_dir):
    """limiter.output is the EXPENSIVE_BOUNDS_CHECK build (limiter.cc:10); the device runs the production
    control flow (limiter.template.h:183-217). For the six in-bounds cases both return the same l up to the
    Newton tolerance -- the well-conditioned ones to round-off; for the six
    cases whose LOW-ORDER state violates the bounds both report failure."""
    params = oracle.default_params(capi.EQ_EULER, 1)
    blocks = _limiter_blocks(os.path.join(golden_dir, "euler_limiter.output"))
    assert len(blocks) == len(LIMITER_CASES) == 12
    items = [np.concatenate([bounds, U, P]) for _, U, P, bounds in LIMITER_CASES]
    out = _device(params, capi.DEBUG_EULER_LIMIT_1D, items, 3)
    lib = oracle.load()
    for (label, U, P, bounds), block, (l, success, _) in zip(LIMITER_CASES, blocks, out):
        l_ref = _grab(block, "\nl:")[0]
        low_order_violation = "low-order" in block
        assert bool(success) == (not low_order_violation), label
        if not low_order_violation:
            assert "Success!" in block
            assert abs(l - l_ref) <= 1e-10, (label, l, l_ref)
        # and the oracle's production flow (same control flow as the device): round-off
        lo, so = C.c_double(), C.c_int()
        arr = lambda t: capi.as_ptr(np.array(t, dtype=np.float64), capi.c_double_p)  # noqa: E731
        lib.ryujin_oracle_euler_limit(C.byref(params), arr(bounds), arr(U), arr(P), C.byref(lo), C.byref(so))
        assert abs(l - lo.value) <= 1e-13, (label, l, lo.value)
        assert bool(success) == bool(so.value), label
    # the l values SURVEY.md 8c quotes for the bound cases
    expected = {6: 0.4999999999999993, 7: 0.1999999188484877, 8: 0.4999999999999998, 9: 0.1999999188484877,
                10: 0.0336589067585305, 11: 0.0000000003370627}
    for n, l_ref in expected.items():
        assert abs(out[n, 0] - l_ref) <= 1e-10, (n, out[n, 0], l_ref)


def test_device_limiter_in_the_checked_control_flow_reproduces_all_twelve_rows(oracle, golden_dir):
    """tests/euler/limiter.output was written by an EXPENSIVE_BOUNDS_CHECK build (limiter.cc:10). The device's
    limit_checked() (ryujin_hip_params::debug_expensive_bounds_check, RYUJIN_DEBUG_EULER_LIMIT_CHECKED_1D) runs that
    control flow: every row of the baseline -- the six Failure cases included -- is reproduced, l to 1e-13 against the
    oracle's checked flow (which tests/test_oracle_golden_euler.py pins on the baseline's printed trace) and to the
    printed digits against the baseline itself, Success / Failure exactly."""
    from test_oracle_golden_euler import _run_limit
    params = oracle.default_params(capi.EQ_EULER, 1)
    blocks = _limiter_blocks(os.path.join(golden_dir, "euler_limiter.output"))
    assert len(blocks) == len(LIMITER_CASES) == 12
    items = [np.concatenate([bounds, U, P]) for _, U, P, bounds in LIMITER_CASES]
    out = _device(params, capi.DEBUG_EULER_LIMIT_CHECKED_1D, items, 3)
    n_failures = 0
    for (label, U, P, bounds), block, (l, success, _) in zip(LIMITER_CASES, blocks, out):
        ref, _ = _run_limit(oracle, params, True, U, P, bounds)
        assert abs(l - ref[0]) <= 1e-13, (label, l, ref[0])
        assert bool(success) == bool(ref[1]), label
        assert abs(l - _grab(block, "\nl:")[0]) <= 5e-16 + 1e-13, (label, l)
        assert ("Success!" in block) == bool(success) and ("Failure!" in block) == (not bool(success)), label
        n_failures += not bool(success)
    assert n_failures == 6


def test_device_sw_riemann_solver_against_the_reference_baseline(oracle, golden_dir):
    text = open(os.path.join(golden_dir, "shallow_water_riemann_solver.output")).read()
    lam = [float(x) for x in re.findall(r"lambda_max: ([0-9.e+-]+)", text)]
    hst = [float(x) for x in re.findall(r"h_star: ([0-9.e+-]+)", text)]
    assert len(lam) == len(hst) == 3
    params = oracle.default_params(capi.EQ_SHALLOW_WATER, 1)
    eps = np.finfo(np.float64).eps

    def riemann_data(state):   # tests/shallow_water/riemann_solver.cc:33-44
        h = max(state[0], params.reference_water_depth * params.dry_state_relaxation_small * eps)
        return [h, state[1] / h, np.sqrt(params.gravity * h)]

    cases = [((0.0, 0.0), (0.0, 0.0)), ((1.0, 1.0), (0.0, 0.0)), ((1.8, 0.0), (1.0, 0.0))]
    out = _device(params, capi.DEBUG_SW_RIEMANN, [riemann_data(a) + riemann_data(b) for a, b in cases], 2)
    for (h_star, l), h_ref, l_ref in zip(out, hst, lam):
        assert abs(h_star - h_ref) <= 1e-13 * abs(h_ref), (h_star, h_ref)
        assert abs(l - l_ref) <= 1e-13 * abs(l_ref), (l, l_ref)
    # the evaluation path the sweeps run (per-node records, ShallowWater::dij_from_records) on the same three
    # states: 1e-12 relative
    lam_rec = _device(params, capi.DEBUG_SW_RIEMANN_RECORDS, [riemann_data(a) + riemann_data(b) for a, b in cases], 1)[:, 0]
    for l, l_ref in zip(lam_rec, lam):
        assert abs(l - l_ref) <= 1e-12 * abs(l_ref), (l, l_ref)


@pytest.mark.parametrize("records", [False, True])
@pytest.mark.parametrize("dim,decades", [(2, (4, 6)), (3, (4, 6)), (2, (10, 10))])
def test_device_dij_against_the_oracle_on_random_states(oracle, dim, decades, records):
    """d_ij = |c_ij| lambda_max(U_i, U_j, c_ij/|c_ij|) for 200 k random admissible state pairs and directions,
    Mach numbers up to 5, density / pressure ratios up to 1e4 / 1e6 -- and up to 1e10 / 1e10, beyond the Leblanc
    state of the reference's unit test: 1e-12 relative (the stated d_ij contract) -- through dij_from_states
    (the reference's operation order) and through the per-node Riemann records of the sweep (k_dij_records: a
    different but equivalent evaluation, see euler_device.hpp)."""
    rng = np.random.default_rng(7)
    n = 200_000
    params = oracle.default_params(capi.EQ_EULER, dim)
    k = dim + 2

    def states():
        rho = 10.0 ** rng.uniform(1 - decades[0], 1, n)
        p = 10.0 ** rng.uniform(2 - decades[1], 2, n)
        a = np.sqrt(params.gamma * p / rho)
        v = rng.normal(size=(n, dim))
        v *= (rng.uniform(0, 5, n) * a / np.linalg.norm(v, axis=1))[:, None]
        U = np.empty((n, k))
        U[:, 0] = rho
        U[:, 1:1 + dim] = rho[:, None] * v
        U[:, -1] = p / (params.gamma - 1.0) + 0.5 * rho * (v ** 2).sum(1)
        return U

    U_i, U_j = states(), states()
    c = rng.normal(size=(n, dim)) * 10.0 ** rng.uniform(-4, 0, n)[:, None]
    which = {(2, False): capi.DEBUG_EULER_DIJ_2D, (3, False): capi.DEBUG_EULER_DIJ_3D,
             (2, True): capi.DEBUG_EULER_DIJ_RECORDS_2D, (3, True): capi.DEBUG_EULER_DIJ_RECORDS_3D}[(dim, records)]
    got = _device(params, which, np.hstack([U_i, U_j, c]), 1)[:, 0]
    lib = oracle.load()
    ref = np.empty(n)
    dp = capi.c_double_p
    U_i, U_j, c = (np.ascontiguousarray(x) for x in (U_i, U_j, c))
    lib.ryujin_oracle_euler_dij_batch(C.byref(params), n, capi.as_ptr(U_i, dp), capi.as_ptr(U_j, dp),
                                      capi.as_ptr(c, dp), capi.as_ptr(ref, dp))
    rel = np.abs(got - ref) / np.abs(ref)
    assert rel.max() <= 1e-12, (rel.max(), int(rel.argmax()))


@pytest.mark.parametrize("records", [False, True])
def test_device_sw_dij_against_the_oracle_on_random_states(oracle, records):
    """shallow water d_ij for 200 k random state pairs incl. dry and nearly dry states, Froude numbers up to 4 --
    through dij_from_states (the reference's operation order) and through the per-node records of the sweep."""
    rng = np.random.default_rng(11)
    n = 200_000
    params = oracle.default_params(capi.EQ_SHALLOW_WATER, 2)

    def states():
        h = 10.0 ** rng.uniform(-6, 1, n)
        h[rng.uniform(size=n) < 0.05] = 0.0                      # dry
        a = np.sqrt(params.gravity * np.maximum(h, 1e-300))
        v = rng.normal(size=(n, 2))
        v *= (rng.uniform(0, 4, n) * a / np.linalg.norm(v, axis=1))[:, None]
        return np.column_stack([h, h * v[:, 0], h * v[:, 1]])

    U_i, U_j = states(), states()
    c = rng.normal(size=(n, 2)) * 10.0 ** rng.uniform(-4, 0, n)[:, None]
    got = _device(params, capi.DEBUG_SW_DIJ_RECORDS_2D if records else capi.DEBUG_SW_DIJ_2D,
                  np.hstack([U_i, U_j, c]), 1)[:, 0]
    ref = np.empty(n)
    dp = capi.c_double_p
    U_i, U_j, c = (np.ascontiguousarray(x) for x in (U_i, U_j, c))
    oracle.load().ryujin_oracle_sw_dij_batch(C.byref(params), n, capi.as_ptr(U_i, dp), capi.as_ptr(U_j, dp),
                                             capi.as_ptr(c, dp), capi.as_ptr(ref, dp))
    # the wave speed is a difference u -+ a sqrt(..): scale by |c| (|u| + a) of the faster state
    h = np.maximum(np.maximum(U_i[:, 0], U_j[:, 0]), 1e-300)
    speed = np.maximum(np.linalg.norm(U_i[:, 1:], axis=1) / np.maximum(U_i[:, 0], 1e-300) * (U_i[:, 0] > 0),
                       np.linalg.norm(U_j[:, 1:], axis=1) / np.maximum(U_j[:, 0], 1e-300) * (U_j[:, 0] > 0))
    scale = np.linalg.norm(c, axis=1) * (speed + np.sqrt(params.gravity * h))
    err = np.abs(got - ref) / np.maximum(scale, 1e-300)
    assert err.max() <= 1e-12, (err.max(), int(err.argmax()))


# ----------------------------------------------------------------------------- EulerAEOS

@pytest.mark.parametrize("variant", ["", "-strict", "-strict-NASG"])
def test_device_aeos_riemann_solver_against_the_reference_baselines(oracle, golden_dir, variant):
    """The device Riemann solver of the EulerAEOS Description on the 22 (21) problems of
    tests/euler_aeos/riemann_solver{,-strict,-strict-NASG}.cc -- constant and varying surrogate gamma, covolume,
    NASG reference pressure, gamma up to 118 and down to 1.00005 -- against the lambda_max of the reference's
    baselines. The pressure estimates are powers with exponent 2 gamma / (gamma - 1) (4e4 for gamma = 1.00005):
    their condition number scales the tolerance, as for the oracle (tests/test_oracle_golden_aeos.py)."""
    from test_oracle_golden_aeos import RIEMANN_CASES as AEOS_CASES
    from test_oracle_golden_aeos import _params, _riemann_blocks
    blocks = _riemann_blocks(os.path.join(golden_dir, f"euler_aeos_riemann_solver{variant}.output"))
    cases = list(AEOS_CASES)
    nasg = variant.endswith("NASG")
    if nasg:
        del cases[17]
    assert len(blocks) == len(cases)
    for n, ((left, right, cov), (ins, tr, lam_ref)) in enumerate(zip(cases, blocks)):
        if cov is None:
            p = _params(oracle, strict=(variant != ""))
        elif nasg:
            p = _params(oracle, eos=capi.EOS_NOBLE_ABEL_STIFFENED_GAS, b=cov, pinf=0.5, strict=True)
        else:
            p = _params(oracle, eos=capi.EOS_VAN_DER_WAALS, b=cov, strict=(variant != ""))
        item = []
        for rho, u, pr, gamma in (left, right):
            x = 1. - cov * rho if cov else 1. - 0. * rho
            item += [rho, u, pr, gamma, np.sqrt(gamma * pr / (rho * x))]
        lam = _device(p, capi.DEBUG_AEOS_RIEMANN, [item], 1)[0, 0]
        g_min = min(left[3], right[3])
        rel = 2e-12 + 1e-15 * 2. * g_min / (g_min - 1.)
        scale = max(abs(left[2]), abs(right[2]), abs(left[1]), abs(right[1]), 1e-300)
        assert abs(lam - lam_ref) <= rel * max(abs(lam), abs(lam_ref)) + 1e-13 * scale, (n, lam, lam_ref)


@pytest.mark.parametrize("nasg", [False, True])
def test_device_aeos_limiter_against_the_reference_baselines(oracle, golden_dir, nasg):
    """limiter{,-NASG}.output are EXPENSIVE_BOUNDS_CHECK builds (limiter.cc:10); the device runs the production
    control flow. Cases whose low-order state violates the bounds report failure on both; for the others l agrees
    up to the Newton tolerance with the baseline and to round-off with the oracle's production flow."""
    from test_oracle_golden_aeos import _limiter_cases, _limiter_golden, _params
    if nasg:
        _, eos, components, compress = _limiter_cases(True)
        cases = ([(dict(), c) for c in components(2.0, 1.8, 1.82, None)] +
                 [(eos(1.0e-1), c) for c in components(2.0, 1.5, 1.7448913582358123, None, s0_first=1.7)] +
                 [(eos(0.2), c) for c in compress(0.2)])
        gold = _limiter_golden(os.path.join(golden_dir, "euler_aeos_limiter-NASG.output"))
    else:
        cases = _limiter_cases(False)
        gold = _limiter_golden(os.path.join(golden_dir, "euler_aeos_limiter.output"))
    assert len(cases) == len(gold)
    lib = oracle.lib()
    dbl = lambda *v: (C.c_double * len(v))(*v)  # noqa: E731
    for n, ((eos_kw, (U, P, bounds)), ref) in enumerate(zip(cases, gold)):
        p = _params(oracle, **eos_kw)
        l, success, _ = _device(p, capi.DEBUG_AEOS_LIMIT_1D, [list(bounds) + list(U) + list(P)], 3)[0]
        lo, so = C.c_double(), C.c_int()
        trace = (C.c_double * 40)()
        assert lib.ryujin_oracle_aeos_limit(C.byref(p), 0, dbl(*bounds), dbl(*U), dbl(*P), C.byref(lo), C.byref(so),
                                            trace, 40) == 0    # the oracle's PRODUCTION flow
        assert bool(success) == bool(so.value), n
        assert abs(l - lo.value) <= 1e-13, (n, l, lo.value)
        if ref["success"]:
            assert bool(success), n
            assert abs(l - ref["l"]) <= 1e-10, (n, l, ref["l"])


@pytest.mark.parametrize("records", [False, True])
@pytest.mark.parametrize("eos", ["polytropic", "covolume", "nasg"])
def test_device_aeos_dij_against_the_oracle_on_random_states(oracle, eos, records):
    """EulerAEOS d_ij = |c_ij| lambda_max for 20 k random admissible state pairs and directions (density over 3.6
    decades, internal energy over 2.7, Mach numbers up to 3; polytropic gas, covolume b = 0.1, NASG with b = 0.1 and
    p_infty = 0.5): in the reference's operation order (dij_from_states) and through the per-node Riemann records
    the sweeps use (k_dij_aeos: rho, p, gamma, a, alpha, alpha_hat and the velocity per node, normal velocity
    v . n instead of (m . n) / rho) against the oracle's pipeline from states, 1e-12 relative."""
    from test_oracle_golden_aeos import _params
    kw = {"polytropic": dict(), "covolume": dict(eos=capi.EOS_VAN_DER_WAALS, b=0.1),
          "nasg": dict(eos=capi.EOS_NOBLE_ABEL_STIFFENED_GAS, b=0.1, pinf=0.5)}[eos]
    params = _params(oracle, dim=2, **kw)
    rng = np.random.default_rng(11)
    n = 20_000

    def states():
        rho = 10.0 ** rng.uniform(-3, 0.6, n)
        rho_e = 10.0 ** rng.uniform(0.3, 3, n)
        a = np.sqrt(1.4 * 0.4 * rho_e / rho)
        v = rng.normal(size=(n, 2))
        v *= (rng.uniform(0, 3, n) * a / np.linalg.norm(v, axis=1))[:, None]
        return np.column_stack([rho, rho[:, None] * v, rho_e + 0.5 * rho * (v ** 2).sum(1)])

    U_i, U_j = states(), states()
    c = rng.normal(size=(n, 2)) * 10.0 ** rng.uniform(-4, 0, n)[:, None]
    got = _device(params, capi.DEBUG_AEOS_DIJ_RECORDS_2D if records else capi.DEBUG_AEOS_DIJ_2D,
                  np.hstack([U_i, U_j, c]), 1)[:, 0]
    lib = oracle.lib()
    norm = np.linalg.norm(c, axis=1)
    nrm = np.ascontiguousarray(c / norm[:, None])
    U_i, U_j = np.ascontiguousarray(U_i), np.ascontiguousarray(U_j)
    dp = capi.c_double_p
    ref = np.array([lib.ryujin_oracle_aeos_lambda_max(C.byref(params), capi.as_ptr(U_i[q], dp), capi.as_ptr(U_j[q], dp),
                                                      capi.as_ptr(nrm[q], dp)) for q in range(n)]) * norm
    rel = np.abs(got - ref) / np.abs(ref)
    assert rel.max() <= 1e-12, (rel.max(), int(rel.argmax()))

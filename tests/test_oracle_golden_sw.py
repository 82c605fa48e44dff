"""Pin the shallow-water Riemann solver of the CPU oracle against the reference's golden output
tests/shallow_water/riemann_solver.{cc:75-77,output} (verified against Mathematica by the authors):
dry/dry, wet/dry and wet/wet states -> h_star and lambda_max at 17 digits."""
import ctypes as C
import os
import re

import numpy as np

from ryujin_amd import capi

CASES = [((0.0, 0.0), (0.0, 0.0)), ((1.0, 1.0), (0.0, 0.0)), ((1.8, 0.0), (1.0, 0.0))]


def test_sw_riemann_solver_golden(oracle, golden_dir):
    text = open(os.path.join(golden_dir, "shallow_water_riemann_solver.output")).read()
    lam = [float(x) for x in re.findall(r"lambda_max: ([0-9.e+-]+)", text)]
    hst = [float(x) for x in re.findall(r"h_star: ([0-9.e+-]+)", text)]
    assert len(lam) == len(hst) == 3
    # SURVEY.md 8c quotes these
    assert lam == [4.6671807060735897e-07, 7.2598063846511982, 4.2021423107743505]
    params = oracle.default_params(capi.EQ_SHALLOW_WATER, 1)
    g = params.gravity
    eps = np.finfo(np.float64).eps
    lib = oracle.load()

    def riemann_data(state):
        # tests/shallow_water/riemann_solver.cc:33-44: h = water_depth_sharp, u = q / h, a = sqrt(g h)
        h = max(state[0], params.reference_water_depth * params.dry_state_relaxation_small * eps)
        return np.array([h, state[1] / h, np.sqrt(g * h)])

    for (Ui, Uj), l_ref, h_ref in zip(CASES, lam, hst):
        rd_i, rd_j = riemann_data(Ui), riemann_data(Uj)
        out = np.zeros(2)
        rc = lib.ryujin_oracle_sw_riemann(C.byref(params), capi.as_ptr(rd_i, capi.c_double_p),
                                          capi.as_ptr(rd_j, capi.c_double_p), capi.as_ptr(out, capi.c_double_p))
        assert rc == 0
        assert abs(out[0] - h_ref) <= 1e-13 * abs(h_ref), (out[0], h_ref)
        assert abs(out[1] - l_ref) <= 1e-13 * abs(l_ref), (out[1], l_ref)

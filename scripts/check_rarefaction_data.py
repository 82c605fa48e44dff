#!/usr/bin/env python3
"""The single-rarefaction verification baselines (tests/golden/*rarefaction*) are reproduced only to 5.2e-10 in the
final time, an offset a ~7e-11 asymmetry of the initial / Dirichlet data would explain (DESIGN.md section 9,
tests/test_oracle_golden_verification.py). This script rules OUR evaluation of the data out: the formulas of
source/euler/initial_state_rarefaction.h:66-153 in 60-digit arithmetic on the double constants the reference's code
holds, against ryujin_amd.initial_states.euler_rarefaction at all 1601 nodes and four times. Prints the largest
relative difference (7.9e-16 when this was written). CPU only, ~1 minute."""
import os
import sys
from decimal import Decimal as D
from decimal import getcontext

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ryujin_amd.initial_states import euler_rarefaction  # noqa: E402

getcontext().prec = 60


def exact_state(xf, tf, gamma=1.4, position=0.2):
    g = float(gamma)
    rho_l, p_l = 3.0, 1.0
    c_l = float(np.sqrt(g * p_l / rho_l))
    u_l = c_l
    rho_r = 0.5
    p_r = float(np.power(rho_r / rho_l, g)) * p_l
    c_r = float(np.sqrt(g * p_r / rho_r))
    u_r = u_l + 2.0 * (c_l - c_r) / (g - 1.0)
    k1 = 2.0 / (g + 1.0)
    k2 = (g - 1.0) / ((g + 1.0) * c_l)
    de = 2.0 / (g - 1.0)
    k3 = c_l + ((g - 1.0) / 2.0) * u_l
    pe = 2.0 * g / (g - 1.0)
    x = float(xf - position)          # the reference subtracts the position in double
    t = float(0.2 / (u_r - u_l) + tf)
    if x <= t * (u_l - c_l):
        rho, u, p = D(rho_l), D(u_l), D(p_l)
    elif x <= t * (u_r - c_r):
        chi = D(x) / D(t)
        base = D(k1) + D(k2) * (D(u_l) - chi)
        rho, u, p = D(rho_l) * base ** D(de), D(k1) * (D(k3) + chi), D(p_l) * base ** D(pe)
    else:
        rho, u, p = D(rho_r), D(u_r), D(p_r)
    return rho, rho * u, p / D(g - 1.0) + D("0.5") * rho * u * u


x = np.linspace(0.0, 1.0, 1601)
worst = 0.0
for t in (0.0, 4.2823450e-05, 0.1, 0.30558):
    U = euler_rarefaction(x[:, None], t, gamma=1.4, position=0.2)
    for i in range(x.size):
        ex = exact_state(x[i], t)
        for c in range(3):
            worst = max(worst, float(abs((D(float(U[i, c])) - ex[c]) / ex[c])))
    print(f"t = {t:g}: largest relative difference so far {worst:.3e}")

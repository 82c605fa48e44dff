"""One-off check (not part of the test suite): GPU and CPU-oracle trajectories of the Mach-3 step over many
Runge-Kutta steps -- how fast do round-off differences (pow, limiter branch flips) grow?"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
import oracle_py as oracle
from ryujin_amd import HyperbolicModule, TimeIntegrator, capi, offline
from ryujin_amd.initial_states import euler_uniform

cpu, n_rk = int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 300
off = offline.SyntheticOffline(offline.mach3_step_2d(cpu))
U0 = euler_uniform(off.positions)
dirichlet = euler_uniform(off.b_positions)
res = {}
for name, backend in (("hip", "hip"), ("oracle", oracle.backend())):
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend=backend)
    sv = m.new_state_vector(U0)
    ti = TimeIntegrator(m, "ssprk 33", cfl_min=0.9, cfl_max=0.9, cfl_recovery_strategy="none",
                        dirichlet_fn=lambda t: dirichlet)
    t, snaps = 0.0, {}
    for n in range(1, n_rk + 1):
        sv, tau = ti.step(sv, t)
        t += tau
        if n in (10, 50, 100, 200, n_rk):
            snaps[n] = (t, sv.download()[: off.n_owned].copy())
    res[name] = snaps
    print(name, "t_final", t, "warnings", m.n_warnings(), flush=True)
for n in sorted(res["hip"]):
    tg, Ug = res["hip"][n]; tc, Uc = res["oracle"][n]
    scale = np.abs(Uc).max(axis=0)
    err = np.abs(Ug - Uc) / scale
    mass = lambda U: (off.mi[: off.n_owned] * U[:, 0]).sum()
    print(f"RK step {n:4d}: |dt|/t = {abs(tg - tc) / tc:.2e}, max rel dU = {err.max():.2e}, "
          f"99.9% quantile = {np.quantile(err, 0.999):.2e}, mass diff = {abs(mass(Ug) - mass(Uc)) / mass(Uc):.2e}")

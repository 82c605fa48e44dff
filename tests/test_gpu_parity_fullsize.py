"""HIP vs oracle AT THE SIZES BASELINE.json NAMES: one test per config. The flow is developed on the GPU (device
resident SSPRK33 steps), the developed state is handed to the CPU oracle, and ONE update (prepare_state_vector +
step<0>) is compared array by array with the tolerances and the psi_r branch-flip classification of
tests/helpers_parity.py. This is where 32-bit position arithmetic, the large-grid launch paths and the
slice/ghost layout of multi-million-row meshes are exercised against the oracle (the small-mesh tests cannot).

  C2  2-D Euler Mach-3 forward-facing step, h = 1/995          2.50 M gridpoints (bench.py's workload)
  C3  3-D Euler radial contrast ("Sedov-like"), 200^3 cells     8.12 M gridpoints, 216 M stencil entries
  C4  3-D Euler Mach-3 cylinder in a channel, per-GPU share     4.18 M gridpoints (h = 1/96, 1.25 units long)
  C5  2-D shallow-water circular dam break, 1824^2 cells        3.33 M gridpoints

Host memory: the oracle holds the whole problem (C3: 33 GB) next to the generator's arrays and the fetched
device arrays; a box with less memory runs the largest mesh that fits (and says so)."""
import gc
import os

import numpy as np
import pytest

from helpers_parity import compare_step
from ryujin_amd import HyperbolicModule, capi, offline
from ryujin_amd.initial_states import euler_radial_contrast, euler_uniform, sw_circular_dam_break
from ryujin_amd.workloads import benchmark_workload, developed_state

pytestmark = pytest.mark.gpu


def _available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 64.0


def _fullsize(oracle, spec, initial, equation, n_develop_rk, with_dirichlet, label, fetch_pij):
    off = offline.SyntheticOffline(spec)
    U0 = initial(off.positions)
    dirichlet = euler_uniform(off.b_positions) if with_dirichlet else None
    p = oracle.default_params(equation, off.dim)
    p.cfl = 0.9
    mg = HyperbolicModule(off, p, backend="hip")
    state = mg.new_state_vector(U0)
    temps = [mg.new_state_vector() for _ in range(3)]
    for _ in range(n_develop_rk):
        mg.time_step("ssprk 33", state, temps, dirichlet)
    assert mg.n_warnings() == 0
    U_start = state.download()
    assert np.isfinite(U_start).all()
    mc = HyperbolicModule(off, p, backend=oracle.backend())
    mods = [(mg, state, temps[0]), (mc, mc.new_state_vector(U_start), mc.new_state_vector())]
    del U0
    gc.collect()
    g, c = compare_step(off, mods, dirichlet, oracle=oracle, params=p, label=label, fetch_pij=fetch_pij,
                        keep_matrices=False)
    n = off.n_owned
    # the developed flow exercises the limiter: a non-trivial share of the pairs is limited (l < 1)
    limited = float((c["lij_next"] < 1.0).mean())
    # invariant domain preserved in sign on every DoF, on both backends alike
    rho_g, rho_c = g["U"][:n, 0], c["U"][:n, 0]
    assert (rho_g > 0).all() and (np.sign(rho_g) == np.sign(rho_c)).all()
    if equation == capi.EQ_EULER:
        e_g = g["U"][:n, -1] - 0.5 * (g["U"][:n, 1:-1] ** 2).sum(1) / rho_g
        assert (e_g > 0).all()
    mc.close()
    mg.close()
    return off, g, c, limited


def _bench_state(oracle, key, label, fetch_pij, limited_slices, develop_time=None, warmup=12, **kw):
    """One update on the state bench.py times for workload `key`, against the oracle. limited_slices = (lo, hi): the
    bench lines' limited_slice_fraction for this workload (profiles/r04*_bench*.json, r05*)."""
    wl = benchmark_workload(key, **kw)
    off = offline.SyntheticOffline(wl.make_spec(wl.resolution, 1, 0))
    U0, t_start, info = developed_state(wl, off, develop_time)
    dirichlet = wl.dirichlet_fn(off.b_positions) if (wl.dirichlet_fn is not None and off.n_bdry) else None
    p = oracle.default_params(wl.equation, off.dim)
    p.cfl = 0.9
    mg = HyperbolicModule(off, p, backend="hip")
    state = mg.new_state_vector(U0)
    temps = [mg.new_state_vector() for _ in range(3)]
    del U0
    # bench.py: develop_updates single updates (an SSPRK33 step is three), then --warmup more before the clock starts
    n_rk = (wl.develop_updates + warmup) // 3
    for q in range(n_rk):
        mg.time_step("ssprk 33", state, temps, dirichlet if q == 0 else None)
    assert mg.n_warnings() == 0
    stats = mg.limiter_statistics()
    U_start = state.download()
    assert np.isfinite(U_start).all()
    mc = HyperbolicModule(off, p, backend=oracle.backend())
    mods = [(mg, state, temps[0]), (mc, mc.new_state_vector(U_start), mc.new_state_vector())]
    del U_start
    gc.collect()
    g, c = compare_step(off, mods, dirichlet, oracle=oracle, params=p, label=label, fetch_pij=fetch_pij,
                        keep_matrices=False)
    n = off.n_owned
    rho_g, rho_c = g["U"][:n, 0], c["U"][:n, 0]
    assert (rho_g > 0).all() and (np.sign(rho_g) == np.sign(rho_c)).all()
    if wl.equation == capi.EQ_EULER:
        e_g = g["U"][:n, -1] - 0.5 * (g["U"][:n, 1:-1] ** 2).sum(1) / rho_g
        assert (e_g > 0).all()
    lo, hi = limited_slices
    assert lo <= stats["limited_slice_fraction"] <= hi, (stats, limited_slices)
    mc.close()
    mg.close()
    return off, g, c, stats


def test_c2_bench_state(oracle):
    """the headline: t = 2.0 of the Mach-3 step, 93 % of the slices limited, plain kernels, P_ij stored per tile"""
    off, g, c, stats = _bench_state(oracle, "step2d", "bench_c2", True, (0.89, 0.97))
    assert off.n_owned == 2498844
    assert stats["pij_stored"] == "per tile"


def test_c2_bench_state_t1(oracle):
    """bench.py --develop-time 1.0: 73 % of the slices limited, a third of the tiles stored"""
    off, g, c, stats = _bench_state(oracle, "step2d", "bench_c2_t1", True, (0.68, 0.79), develop_time=1.0)
    assert stats["pij_stored"] == "per tile"


def test_c3_bench_state(oracle):
    """bench.py --workload sedov3d: the blast wave half-way to the walls, 41-44 % limited, per slice + repair"""
    avail = _available_gb()
    if avail <= 75:
        pytest.skip(f"BASELINE configs[2] at 200^3 cells needs ~60 GB of host memory for the oracle, "
                    f"{avail:.0f} GB available")
    # (P_ij as well -- 8.8 GB per backend, 1.1 G matrix entries -- where the host has the room)
    off, g, c, stats = _bench_state(oracle, "sedov3d", "bench_c3", avail > 160, (0.36, 0.50))
    assert off.n_owned == 201 ** 3
    assert stats["pij_stored"] == "per slice"


def test_c4_bench_state(oracle):
    """bench.py --workload cylinder3d: the bow shock stands and has reflected off the walls, every slice limited"""
    # P_ij too where the host has room for it (4.2 M rows x 27 entries x 5 components: 4.5 GB per backend, a few copies
    # in the comparison): 564 M matrix entries, i.e. offsets beyond 2^29 into p_ij, compared entry by entry
    fetch_pij = _available_gb() > 48
    off, g, c, stats = _bench_state(oracle, "cylinder3d", "bench_c4", fetch_pij, (0.95, 1.0))
    assert off.n_owned > 4_000_000


def test_c5_bench_state(oracle):
    """bench.py --workload sw2d: the bore half-way to the walls"""
    off, g, c, stats = _bench_state(oracle, "sw2d", "bench_c5", True, (0.60, 0.95))
    assert off.n_owned == 1825 ** 2


def test_fullsize_c2_step_2d(oracle):
    """BASELINE configs[1], the bench line's mesh: 2 498 844 gridpoints = 9 995 376 DoFs."""
    def initial(pos):
        rng = np.random.default_rng(42)
        return euler_uniform(pos) * (1.0 + 1e-3 * rng.uniform(-1.0, 1.0, size=(len(pos), 4)))
    off, g, c, limited = _fullsize(oracle, offline.mach3_step_2d(995), initial, capi.EQ_EULER, 100, True,
                                   "fullsize_c2", True)
    assert off.n_owned == 2498844
    assert limited > 1e-3, limited


def test_miniature_cylinder_against_the_oracle(oracle):
    """configs[3]'s geometry (3-D Dirichlet + do-nothing + slip + staircase cylinder with 3-D coupling boundary
    pairs) on 10 k gridpoints, every sweep against the ORACLE (the partitioned cylinder tests compare HIP with
    HIP)."""
    spec = offline.cylinder_channel_3d(8, length_units=2)
    off = offline.SyntheticOffline(spec)
    rng = np.random.default_rng(3)
    U0 = euler_uniform(off.positions) * (1.0 + 1e-3 * rng.uniform(-1.0, 1.0, size=(off.n_relevant, 5)))
    dirichlet = euler_uniform(off.b_positions)
    p = oracle.default_params(capi.EQ_EULER, 3)
    p.cfl = 0.9
    mg = HyperbolicModule(off, p, backend="hip")
    a, b = mg.new_state_vector(U0), mg.new_state_vector()
    for _ in range(25):
        mg.prepare_state_vector(a, 0.0, dirichlet)
        mg.step(a, [], [], b)
        a, b = b, a
    mc = HyperbolicModule(off, p, backend=oracle.backend())
    mods = [(mg, a, b), (mc, mc.new_state_vector(a.download()), mc.new_state_vector())]
    assert off.n_pairs > 0 and set(np.unique(off.b_id)) >= {capi.BC_DIRICHLET, capi.BC_SLIP}
    g, c = compare_step(off, mods, dirichlet, oracle=oracle, params=p, label="mini_cylinder")
    assert (g["U"][: off.n_owned, 0] > 0).all()

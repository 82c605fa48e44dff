"""The benchmark workloads of BASELINE.json: mesh recipe, initial and Dirichlet data, and how the
state the timed updates start from is made -- the flow is run to `develop_time` on a coarse mesh of the same domain,
interpolated to the benchmark mesh and re-sharpened by a number of updates there. bench.py times these; the full-size
parity tests (tests/test_gpu_parity_fullsize.py) build the SAME state with the SAME functions and compare one update
with the oracle on it."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np

from . import capi, offline
from .initial_states import euler_radial_contrast, euler_uniform, sw_circular_dam_break
from .module import HyperbolicModule


class Ssprk33Stages:
    """SSPRK33 (time_integrator.template.h:302-328) unrolled into single forward-Euler updates."""

    def __init__(self, module, U0, dirichlet):
        self.m = module
        self.U = module.new_state_vector(U0)
        self.T = [module.new_state_vector(), module.new_state_vector()]
        self.T2 = None
        self.dirichlet = dirichlet
        self.stage = 0
        self.tau = 0.0
        self.t = 0.0
        self.first = True

    def rk_step(self):
        """Three forward-Euler updates = one SSPRK33 step through the device-resident driver
        (ryujin_hip_time_step): one host synchronisation per RK step instead of per update."""
        assert self.stage == 0
        if self.T2 is None:
            self.T2 = self.m.new_state_vector()
        d = self.dirichlet if self.first else None
        self.first = False
        self.tau = self.m.time_step("ssprk 33", self.U, [self.T[0], self.T[1], self.T2], d)
        self.t += self.tau

    def update(self):
        m, U, T = self.m, self.U, self.T
        d = self.dirichlet if self.first else None  # constant Dirichlet data: upload once
        self.first = False
        if self.stage == 0:
            m.prepare_state_vector(U, self.t, d)
            self.tau = m.step(U, [], [], T[0], 0.0)
        elif self.stage == 1:
            m.prepare_state_vector(T[0], self.t + self.tau, d)
            m.step(T[0], [], [], T[1], self.tau)
            m.sadd(T[1], 1.0 / 4.0, 3.0 / 4.0, U)
        else:
            m.prepare_state_vector(T[1], self.t + 0.5 * self.tau, d)
            m.step(T[1], [], [], T[0], self.tau)
            m.sadd(T[0], 2.0 / 3.0, 1.0 / 3.0, U)
            self.U, self.T[0] = T[0], U
            self.t += self.tau
        self.stage = (self.stage + 1) % 3



@dataclass
class Workload:
    key: str
    name: str
    equation: int
    make_spec: Callable  # (resolution, n_ranks, rank) -> MeshSpec
    U0_fn: Callable  # positions -> U
    dirichlet_fn: Optional[Callable]  # boundary positions -> Dirichlet states, or None
    resolution: int
    coarse_resolution: int
    default_develop_time: float
    develop_updates: int = 600  # updates on the benchmark mesh behind the interpolation (bench.py --develop)


def benchmark_workload(key: str, n_gpus: int = 1, cells_per_unit: int = 995, size: int = 0,
                       coarse_factor: int = 4) -> Workload:
    """`key`: step2d (BASELINE.json configs[1], the headline), step2d_aeos, sedov3d (configs[2]), cylinder3d
    (configs[3], one GPU's share) or sw2d (configs[4]). N GPUs: the domain is lengthened so that every GPU keeps
    the gridpoint count of the single-GPU mesh (weak scaling)."""
    equation = capi.EQ_EULER
    dirichlet_fn = None
    if key in ("step2d", "step2d_aeos"):
        # area 0.8 L + 0.12 with the step cut out: 2.52 per GPU
        length = 3.0 if n_gpus == 1 else (2.52 * n_gpus - 0.12) / 0.8
        resolution = cells_per_unit
        coarse_resolution = max(20, int(round(resolution / coarse_factor / 5.0)) * 5)

        def make_spec(n, n_ranks, r):
            return offline.mach3_step_2d(n, length_units=length, n_ranks=n_ranks, rank=r)
        U0_fn = dirichlet_fn = euler_uniform  # prm/benchmarks/euler-mach3-forward-facing-step.prm:55-66
        default_develop_time = 2.0            # of the prm's final time 4.0 (:30)
        name = ("2D Euler Mach-3 forward-facing step, Q1, SSPRK33 stage sequence "
                "(BASELINE.json configs[1])")
        if key == "step2d_aeos":  # same problem through the EulerAEOS Description (f-3)
            equation = capi.EQ_EULER_AEOS
            name = ("2D Euler-AEOS (polytropic gas EOS, strict bounds) Mach-3 forward-facing step, "
                    "Q1, SSPRK33 stage sequence")
    elif key == "cylinder3d":
        # BASELINE.json configs[3]: h = 1/126 on [0,4]x[-1,1]^2 over 8 GPUs is 4M gridpoints per GPU; fewer
        # GPUs keep that per-GPU count with a shorter channel (half a unit of length per GPU, at least 1.25)
        # (the cylinder needs 1.25 units of channel: one or two GPUs run h = 1/96 and 1/120 instead)
        resolution = size or {1: 96, 2: 120}.get(n_gpus, 126)
        coarse_resolution = max(8, resolution // coarse_factor)
        length = max(1.25, 0.5 * n_gpus)

        def make_spec(n, n_ranks, r):
            return offline.cylinder_channel_3d(n, length_units=length, n_ranks=n_ranks, rank=r)
        U0_fn = dirichlet_fn = euler_uniform  # prm/benchmarks/euler-mach3-cylinder-3d.prm:49-91
        default_develop_time = 1.0  # the bow shock stands and has reflected off the channel walls (final time 5.0, :41)
        name = "3D Euler Mach-3 cylinder in a channel, Q1 (BASELINE.json configs[3])"
    elif key == "sedov3d":
        resolution = size or 200
        coarse_resolution = max(8, resolution // coarse_factor)

        def make_spec(n, n_ranks, r):
            return offline.box_3d(n, nx=n * n_gpus, upper=(2.0 * n_gpus - 1.0, 1.0, 1.0), n_ranks=n_ranks, rank=r)

        def U0_fn(positions):
            return euler_radial_contrast(positions, inner=(1.0, 0.0, 100.0), outer=(1.0, 0.0, 0.1), radius=0.1)
        default_develop_time = 0.25           # the blast wave is half-way to the walls
        name = "3D Euler Sedov-like radial contrast, rectangular domain, Q1 (BASELINE.json configs[2])"
    elif key == "sw2d":
        equation = capi.EQ_SHALLOW_WATER
        resolution = size or 1824
        coarse_resolution = max(8, resolution // coarse_factor)

        def make_spec(n, n_ranks, r):
            return offline.rectangle_2d(n * n_gpus, (-5.0, -5.0), (10.0 * n_gpus - 5.0, 5.0), ny=n, n_ranks=n_ranks,
                                        rank=r)
        U0_fn = sw_circular_dam_break
        default_develop_time = 0.5            # the bore has travelled half-way to the walls
        name = "2D shallow-water circular dam break, Q1 (BASELINE.json configs[4])"
    else:
        raise ValueError(f"unknown workload {key!r}")
    return Workload(key, name, equation, make_spec, U0_fn, dirichlet_fn, resolution, coarse_resolution,
                    default_develop_time)


def interpolate_from_lattice(spec_c, pos_c, U_c, pos_f, n_fill: int = 3):
    """Multilinear interpolation of nodal values given on the Cartesian lattice of a coarse synthetic mesh
    (nodes pos_c [n, dim] of spec_c, values U_c [n, k]) to the points pos_f. Lattice nodes the mesh does not have
    (cut-outs: the step, the staircase cylinder) are filled from their nearest existing axis neighbours, n_fill
    layers deep -- a fine fluid point next to a coarser staircase may sit in a cell with such a corner. A convex
    combination of admissible states is admissible (the invariant set is convex)."""
    dim = spec_c.dim
    lower = np.array(spec_c.lower[:dim], dtype=np.float64)
    upper = np.array(spec_c.upper[:dim], dtype=np.float64)
    n_cells = np.array(spec_c.n_cells[:dim], dtype=np.int64)
    h = (upper - lower) / n_cells
    k = U_c.shape[1]
    grid = np.full(tuple(n_cells + 1) + (k,), np.nan)
    idx = np.rint((pos_c - lower) / h).astype(np.int64)
    grid[tuple(idx.T)] = U_c
    for _ in range(n_fill):
        hole = np.isnan(grid[..., 0])
        if not hole.any():
            break
        acc = np.zeros_like(grid)
        cnt = np.zeros(grid.shape[:-1])
        for ax in range(dim):
            for shift in (1, -1):
                nb = np.roll(grid, shift, axis=ax)
                edge = [slice(None)] * dim
                edge[ax] = 0 if shift == 1 else -1
                nb[tuple(edge)] = np.nan  # no wrap-around
                ok = ~np.isnan(nb[..., 0])
                acc[ok] += nb[ok]
                cnt[ok] += 1.0
        fill = hole & (cnt > 0)
        grid[fill] = acc[fill] / cnt[fill][:, None]
    x = np.clip((pos_f - lower) / h, 0.0, n_cells.astype(np.float64))
    i0 = np.minimum(np.floor(x).astype(np.int64), n_cells - 1)
    t = x - i0
    out = np.zeros((pos_f.shape[0], k))
    for corner in range(1 << dim):
        w = np.ones(pos_f.shape[0])
        ii = []
        for ax in range(dim):
            bit = (corner >> ax) & 1
            w = w * (t[:, ax] if bit else 1.0 - t[:, ax])
            ii.append(i0[:, ax] + bit)
        v = grid[tuple(ii)]
        # a corner that does not exist (deep inside a cut-out) must carry no weight for a fluid point
        bad = np.isnan(v[:, 0])
        assert not (bad & (w > 1e-9)).any(), "fine point inside a coarse cut-out: raise n_fill"
        v = np.where(bad[:, None], 0.0, v)
        out += w[:, None] * v
    return out


def develop_on_coarse_mesh(spec_c, U0_fn, dirichlet_fn, equation, t_final: float, device: int = 0):
    """Run the flow to t_final with SSPRK33 at cfl 0.9 on a coarse mesh of the same domain (one rank, no
    communicator, device-resident driver): (coarse offline data, state at t_final, number of RK steps)."""
    off_c = offline.SyntheticOffline(spec_c)
    m = HyperbolicModule(off_c, equation=equation, backend="hip", device=device)
    m.cfl = 0.9
    U = m.new_state_vector(U0_fn(off_c.positions))
    temps = [m.new_state_vector() for _ in range(3)]
    dirichlet = dirichlet_fn(off_c.b_positions) if (dirichlet_fn is not None and off_c.n_bdry) else None
    t, n = 0.0, 0
    while t < t_final * (1.0 - 1e-12):
        t += m.time_step("ssprk 33", U, temps, dirichlet if n == 0 else None, tau_max=t_final - t)
        n += 1
    U_c = U.download()
    m.close()
    return off_c, U_c, n




def developed_state(wl: Workload, off, develop_time: float | None = None, device: int = 0):
    """The state bench.py's updates start from, before the `develop_updates` updates on the benchmark mesh: the
    coarse run to develop_time interpolated to `off` (the mesh of wl.make_spec(wl.resolution, ...)). Returns
    (U0, t_start, info); develop_time 0: the initial state."""
    develop_time = wl.default_develop_time if develop_time is None else develop_time
    if develop_time <= 0.0:
        return wl.U0_fn(off.positions), 0.0, None
    spec_c = wl.make_spec(wl.coarse_resolution, 1, 0)
    off_c, U_c, n_rk = develop_on_coarse_mesh(spec_c, wl.U0_fn, wl.dirichlet_fn, wl.equation, develop_time, device)
    U0 = interpolate_from_lattice(spec_c, off_c.positions[: off_c.n_owned], U_c[: off_c.n_owned], off.positions)
    info = {"resolution": wl.coarse_resolution, "gridpoints": off_c.n_owned, "ssprk33_steps": n_rk}
    off_c.close()
    return U0, develop_time, info

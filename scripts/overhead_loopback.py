#!/usr/bin/env python3
"""What the multi-rank choreography costs PER RANK, measured on one unshared GPU: the middle rank of a 3-rank
x-slab partition of a channel of uniform cross-section is run alone with the loopback communicator
(ryujin_hip_comm_init_loopback: every neighbour is the rank itself, i.e. a periodic channel) -- split sweeps
(export + interior launches), pack kernels, device-to-device copies, events, the all-reduces, exactly as a
middle rank of a real run issues them -- against the same slab run as an ordinary single-rank mesh. No
RCCL/xGMI latency in it (there is no second GPU on this box); everything else that separates `--gpus 8` from
`--gpus 1` is. usage: overhead_loopback.py [--dim 2|3] [--n cells per direction of one slab] [--rk-steps K]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ryujin_amd import HyperbolicModule, capi, offline  # noqa: E402
from ryujin_amd.initial_states import euler_uniform  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=2)
ap.add_argument("--n", type=int, default=0)
ap.add_argument("--rk-steps", type=int, default=20)
ap.add_argument("--develop", type=int, default=30, help="untimed SSPRK33 steps")
ap.add_argument("--timers", action="store_true", help="also print the per-sweep device times (event pairs)")
args = ap.parse_args()
lib = capi.load_hip()
n = args.n or (1580 if args.dim == 2 else 160)
DN, SL = capi.BC_DO_NOTHING, capi.BC_SLIP


def spec(n_ranks, rank):
    if args.dim == 2:
        return offline.rectangle_2d(n * n_ranks, (0.0, 0.0), (float(n_ranks), 1.0), bc=(DN, DN, SL, SL), ny=n,
                                    n_ranks=n_ranks, rank=rank)
    return offline.box_3d(n, lower=(0.0, 0.0, 0.0), upper=(float(n_ranks), 1.0, 1.0), bc=(DN, DN, SL, SL, SL, SL),
                          nx=n * n_ranks, n_ranks=n_ranks, rank=rank)


def initial(off):
    x = off.positions
    U = euler_uniform(x)
    s = 1.0 + 1e-3 * np.sin(2.0 * np.pi * x[:, 0]) * np.cos(2.0 * np.pi * x[:, 1])   # period 1 = one slab
    return U * s[:, None]


def run(off, comm):
    # the kernels of steps 5 - 7 depend on where the flow is limited, and the two runs differ in that (walls at the
    # ends of the single-rank slab, a periodic channel through the loopback): both run the plain kernels (P_ij stored
    # everywhere), so that the difference is the choreography alone
    HyperbolicModule.library_switches = {"debug_pij_storage": -1}
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend="hip", comm=comm)
    m.cfl = 0.9
    state = m.new_state_vector(initial(off))
    temps = [m.new_state_vector() for _ in range(3)]
    for _ in range(args.develop):
        m.time_step("ssprk 33", state, temps, None)
    lib.ryujin_hip_synchronize(m._ctx)
    t0 = time.perf_counter()
    for _ in range(args.rk_steps):
        m.time_step("ssprk 33", state, temps, None)
    lib.ryujin_hip_synchronize(m._ctx)
    ms = (time.perf_counter() - t0) / (3 * args.rk_steps) * 1e3
    assert np.isfinite(state.download()).all() and m.n_warnings() == 0
    if args.timers:
        lib.ryujin_hip_set_timers(m._ctx, 1)
        acc, cnt = (C.c_double * 8)(), C.c_uint()
        lib.ryujin_hip_get_timers_accum(m._ctx, acc, C.byref(cnt), 1)
        for _ in range(args.rk_steps):
            m.time_step("ssprk 33", state, temps, None)
        lib.ryujin_hip_synchronize(m._ctx)
        lib.ryujin_hip_get_timers_accum(m._ctx, acc, C.byref(cnt), 1)
        lib.ryujin_hip_set_timers(m._ctx, 0)
        print("  sweeps [ms]: " + " ".join(f"{a / max(cnt.value, 1):.4f}" for a in acc)
              + f"  sum {sum(acc) / max(cnt.value, 1):.4f} over {cnt.value} updates", flush=True)
    return ms, off.n_owned


single = offline.SyntheticOffline(spec(1, 0))
t_s, n_s = run(single, None)
print(f"single-rank slab      : {t_s:.4f} ms/update, {n_s} gridpoints", flush=True)
middle = offline.SyntheticOffline(spec(3, 1))
o = middle.c.contents
assert o.n_nbr == 2 and o.send_off[1] - o.send_off[0] == o.send_off[2] - o.send_off[1]
comm = C.c_void_p()
assert lib.ryujin_hip_comm_init_loopback(C.byref(comm), 1, 3, 0) == 0
t_l, n_l = run(middle, comm)
per_dof = (t_l / n_l) / (t_s / n_s)
print(f"middle rank, loopback : {t_l:.4f} ms/update, {n_l} gridpoints + {middle.n_relevant - n_l} ghosts, "
      f"{o.send_off[2]} rows exported per exchange", flush=True)
print(f"per gridpoint: {(per_dof - 1) * 100:+.1f} % against the single-rank run "
      f"(choreography of a middle rank without network latency)", flush=True)
lib.ryujin_hip_comm_destroy(comm)

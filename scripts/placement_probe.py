#!/usr/bin/env python3
"""Placement study: the same library, the same mesh, the same developed state in N contexts of
one process -- only WHERE the allocator put the stencil streams differs. Prints, per context, the device addresses
of the streams (modulo 2 MiB / 1 GiB, and their pairwise distances modulo 4 KiB * 2^k) and the per-sweep device times.
usage: placement_probe.py [--dim 3] [--cells-per-unit 96] [--contexts 5] [--steps 9]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import Ssprk33Stages  # noqa: E402
from ryujin_amd import HyperbolicModule, capi, offline  # noqa: E402
from ryujin_amd.initial_states import euler_uniform  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=3)
ap.add_argument("--cells-per-unit", type=int, default=96)
ap.add_argument("--contexts", type=int, default=5)
ap.add_argument("--develop", type=int, default=150)
ap.add_argument("--steps", type=int, default=9)
ap.add_argument("--rounds", type=int, default=2)
args = ap.parse_args()
lib = capi.load_hip()
spec = (offline.mach3_step_2d(args.cells_per_unit) if args.dim == 2
        else offline.cylinder_channel_3d(args.cells_per_unit, length_units=1.25))
off = offline.SyntheticOffline(spec)
U0, dirichlet = euler_uniform(off.positions), euler_uniform(off.b_positions)
print(f"n_q={off.n_owned}", flush=True)
names = ["cols", "cij", "mij", "dij", "lij", "lij_next", "pij", "idx_t"]
ctxs, U_dev = [], None
for q in range(args.contexts):
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend="hip")
    m.cfl = 0.9
    if U_dev is None:
        d = Ssprk33Stages(m, U0, dirichlet)
        for _ in range(args.develop):
            d.update()
        U_dev = d.U.download()
    drv = Ssprk33Stages(m, U_dev, dirichlet)
    for _ in range(3):
        drv.update()
    lib.ryujin_hip_set_timers(m._ctx, 1)
    addr = (C.c_uint64 * 8)()
    lib.ryujin_hip_debug_addresses(m._ctx, addr)
    ctxs.append((m, drv, np.zeros(8), [0], list(addr)))
tmp = (C.c_double * 8)()
for r in range(args.rounds):
    for m, drv, acc, cnt, addr in ctxs:
        for _ in range(args.steps):
            drv.update()
            lib.ryujin_hip_get_timers(m._ctx, tmp)
            acc += np.array(tmp[:])
            cnt[0] += 1
print("ctx  " + " ".join("%9s" % n for n in ["dij_alpha", "diag", "low_order", "pij_lij", "ho_next", "ho_last"]) + "   total")
for q, (m, drv, acc, cnt, addr) in enumerate(ctxs):
    ms = acc[1:7] / cnt[0]
    print("%-4d " % q + " ".join("%9.4f" % x for x in ms) + "  %8.4f" % ms.sum())
print("\naddresses: offset within 2 MiB in units of 4 KiB | GiB index")
for q, (m, drv, acc, cnt, addr) in enumerate(ctxs):
    print("%-4d " % q + "  ".join("%s %3d|%3d" % (n, (a % (2 << 20)) >> 12, a >> 30) for n, a in zip(names, addr)))
print("\ndistance pij - {dij, mij, lij, cols} modulo 64 KiB, in units of 4 KiB (HBM channel interleave candidates)")
for q, (m, drv, acc, cnt, addr) in enumerate(ctxs):
    a = dict(zip(names, addr))
    print("%-4d " % q + "  ".join("%s %2d" % (n, ((a["pij"] - a[n]) % (64 << 10)) >> 12) for n in ("dij", "mij", "lij", "cols")))

#!/usr/bin/env python3
"""Within-process A/B of library build variants (ryujin_amd/lib/variants/*.so): same mesh, same
developed state, interleaved rounds; prints the mean per-sweep device time of each variant.
usage: ab_variants.py [--cells-per-unit N] [--develop D] [--steps K] [--rounds R] name=path[:switch=value,...] ...
(switch: a run-time field of ryujin_hip_params, e.g. debug_pij_storage=-1 -- the same library may appear several times)
--load-state FILE: the developed state bench.py --save-state wrote (same mesh); --perturbation P with
--perturbed-fractions f1,f2,...: the whole comparison once per fraction of perturbed rows (a sweep over the
fraction of limited slices)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ryujin_amd.workloads import Ssprk33Stages  # noqa: E402
from ryujin_amd import HyperbolicModule, capi, offline  # noqa: E402
from ryujin_amd.initial_states import euler_uniform  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells-per-unit", type=int, default=995)
ap.add_argument("--develop", type=int, default=300)
ap.add_argument("--steps", type=int, default=15)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--dim", type=int, default=2)
ap.add_argument("--length", type=float, default=1.25, help="3-D: channel length in units")
ap.add_argument("--workload", default="", help="sw2d: the shallow-water dam break of bench.py (--cells-per-unit = "
                                               "cells per direction, default 1824); sedov3d: the radial-contrast box "
                                               "(default 160 cells per direction); step2d_aeos: step2d through EulerAEOS; "
                                               "default: step2d / cylinder3d by --dim")
ap.add_argument("--load-state", default=None)
ap.add_argument("--perturbation", type=float, default=0.0)
ap.add_argument("--perturbed-fractions", default="")
ap.add_argument("variants", nargs="+")
args = ap.parse_args()

equation = capi.EQ_EULER
if args.workload == "sw2d":
    from ryujin_amd.initial_states import sw_circular_dam_break
    n = args.cells_per_unit if args.cells_per_unit != 995 else 1824
    spec = offline.rectangle_2d(n, (-5.0, -5.0), (5.0, 5.0), ny=n)
    equation = capi.EQ_SHALLOW_WATER
elif args.workload == "sedov3d":
    from ryujin_amd.initial_states import euler_radial_contrast
    n = args.cells_per_unit if args.cells_per_unit != 995 else 160
    spec = offline.box_3d(n)
elif args.dim == 2:
    spec = offline.mach3_step_2d(args.cells_per_unit)
    if args.workload == "step2d_aeos":  # the same problem through the EulerAEOS Description
        equation = capi.EQ_EULER_AEOS
else:
    spec = offline.cylinder_channel_3d(args.cells_per_unit, length_units=args.length)
off = offline.SyntheticOffline(spec)
if args.workload == "sw2d":
    U0, dirichlet = sw_circular_dam_break(off.positions), None
elif args.workload == "sedov3d":
    U0, dirichlet = euler_radial_contrast(off.positions, inner=(1.0, 0.0, 100.0), outer=(1.0, 0.0, 0.1), radius=0.1), None
else:
    U0 = euler_uniform(off.positions)
    dirichlet = euler_uniform(off.b_positions)
print(f"n_q={off.n_owned}", flush=True)

mods = {}
U_dev = None
if args.load_state:
    U_dev = np.load(args.load_state)["U"]
    assert U_dev.shape[0] == off.n_relevant, "the state file belongs to another mesh"
for v in args.variants:
    name, spec_v = v.split("=", 1)
    path, _, switches = spec_v.partition(":")
    lib = C.CDLL(path)
    capi._declare_module_api(lib, "ryujin_hip_")
    lib.ryujin_hip_set_timers.argtypes = [C.c_void_p, C.c_int]
    lib.ryujin_hip_get_timers.argtypes = [C.c_void_p, capi.c_double_p]
    lib.ryujin_hip_synchronize.argtypes = [C.c_void_p]
    lib.ryujin_hip_event_record.argtypes = [C.c_void_p, C.c_int]
    lib.ryujin_hip_event_elapsed_ms.argtypes = [C.c_void_p, capi.c_double_p]
    p = capi.Params()
    lib.ryujin_hip_default_params(C.byref(p), equation, off.dim)
    env = {}
    for kv in filter(None, switches.split(",")):
        key, value = kv.split("=")
        if key.startswith("env."):  # an environment variable the library reads when the context is created
            env[key[4:]] = value
        else:
            setattr(p, key, int(value))
    os.environ.update(env)
    m = HyperbolicModule(off, p, backend=(lib, "ryujin_hip_"))
    for key in env:
        del os.environ[key]
    m.cfl = 0.9
    if U_dev is None:
        d = Ssprk33Stages(m, U0, dirichlet)
        for _ in range(args.develop):
            d.update()
        U_dev = d.U.download()
    mods[name] = (lib, m)

tmp = (C.c_double * 8)()
names = ["dij_alpha", "diag", "low_order", "pij_lij", "ho_next", "ho_last"]
fractions = [float(x) for x in args.perturbed_fractions.split(",") if x] or [1.0 if args.perturbation else 0.0]
rng = np.random.default_rng(42)
noise = rng.uniform(-1.0, 1.0, size=U_dev.shape)
for frac in fractions:
    U_start = U_dev.copy()
    n_p = int(round(frac * U_start.shape[0])) if args.perturbation else 0
    U_start[:n_p] *= 1.0 + args.perturbation * noise[:n_p]
    runs = {}
    for name, (lib, m) in mods.items():
        drv = Ssprk33Stages(m, U_start, dirichlet)
        for _ in range(9):
            drv.update()
        lib.ryujin_hip_set_timers(m._ctx, 1)
        runs[name] = (lib, m, drv, np.zeros(8), [0, 0.0])
    for r in range(args.rounds):
        for name, (lib, m, drv, acc, cnt) in runs.items():
            lib.ryujin_hip_event_record(m._ctx, 0)
            for _ in range(args.steps):
                drv.update()
                lib.ryujin_hip_get_timers(m._ctx, tmp)
                acc += np.array(tmp[:])
                cnt[0] += 1
            lib.ryujin_hip_event_record(m._ctx, 1)
            ms = C.c_double()
            lib.ryujin_hip_event_elapsed_ms(m._ctx, C.byref(ms))
            cnt[1] += ms.value
    if args.perturbation:
        print(f"--- perturbation {args.perturbation:g} on {frac:.0%} of the rows", flush=True)
    print("%-14s " % "variant" + " ".join("%9s" % n for n in names) + "     total(2-7)  update(events, incl. step 1 and host syncs)")
    for name, (lib, m, drv, acc, cnt) in runs.items():
        ms = acc[1:7] / cnt[0]
        print("%-14s " % name + " ".join("%9.4f" % x for x in ms) + "  %9.4f  %9.4f" % (ms.sum(), cnt[1] / cnt[0]), flush=True)
    for name, (lib, m, drv, acc, cnt) in runs.items():
        if hasattr(lib, "ryujin_hip_limiter_statistics"):
            print("%-14s " % name + "limiter statistics: %s" % m.limiter_statistics(), flush=True)
        for sv in [drv.U, *drv.T] + ([drv.T2] if drv.T2 is not None else []):
            sv.free()

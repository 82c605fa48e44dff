#!/usr/bin/env python3
"""What the multi-rank choreography costs, measured on ONE GPU: the same mesh run by 1 context and by N
in-process ranks (x-slabs, one host thread each, stream-ordered in-process transport = the RCCL leg's event
graph with device-to-device copies in place of ncclSend/ncclRecv). All ranks share the GPU, so the wall time per
update of the N-rank run against the single-context run isolates the fixed costs a rank adds: split sweeps
(export + interior launches), pack kernels, events, the all-reduce, the host-side rendezvous. It does NOT contain
RCCL/xGMI latency. usage: overhead_partitioned.py [--cells-per-unit 995] [--ranks 2 4 8] [--rk-steps 20]"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ryujin_amd import HyperbolicModule, capi, offline  # noqa: E402
from ryujin_amd.initial_states import euler_uniform  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cells-per-unit", type=int, default=995)
ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--rk-steps", type=int, default=20)
ap.add_argument("--develop", type=int, default=100, help="untimed SSPRK33 steps")
args = ap.parse_args()
lib = capi.load_hip()


def run(off, comm, out, key, barrier):
    try:
        m = HyperbolicModule(off, equation=capi.EQ_EULER, backend="hip", comm=comm)
        m.cfl = 0.9
        d = euler_uniform(off.b_positions) if off.n_bdry else None
        state = m.new_state_vector(euler_uniform(off.positions))
        temps = [m.new_state_vector() for _ in range(3)]
        for k in range(args.develop):
            m.time_step("ssprk 33", state, temps, d if k == 0 else None)
        lib.ryujin_hip_synchronize(m._ctx)
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(args.rk_steps):
            m.time_step("ssprk 33", state, temps, None)
        lib.ryujin_hip_synchronize(m._ctx)
        barrier.wait()
        out[key] = (time.perf_counter() - t0) / (3 * args.rk_steps) * 1e3, off.n_owned, m.n_warnings()
    except Exception as e:  # noqa: BLE001
        out[key] = e
        barrier.abort()


ref = {}
run(offline.SyntheticOffline(offline.mach3_step_2d(args.cells_per_unit)), None, ref, 0, threading.Barrier(1))
if isinstance(ref[0], Exception):
    raise ref[0]
t1, n1, _ = ref[0]
print(f"1 context : {t1:.4f} ms/update, {n1} gridpoints", flush=True)
for n_ranks in args.ranks:
    comms = (C.c_void_p * n_ranks)()
    assert lib.ryujin_hip_comm_init_local(comms, n_ranks, 0) == 0
    parts = [offline.SyntheticOffline(offline.mach3_step_2d(args.cells_per_unit, n_ranks=n_ranks, rank=r))
             for r in range(n_ranks)]
    out, barrier = {}, threading.Barrier(n_ranks)
    th = [threading.Thread(target=run, args=(parts[r], C.c_void_p(comms[r]), out, r, barrier)) for r in range(n_ranks)]
    [t.start() for t in th]
    [t.join(timeout=900) for t in th]
    for r in range(n_ranks):
        if isinstance(out[r], Exception):
            raise out[r]
    tn = max(out[r][0] for r in range(n_ranks))
    print(f"{n_ranks} ranks   : {tn:.4f} ms/update for the same {sum(out[r][1] for r in range(n_ranks))} gridpoints "
          f"({(tn / t1 - 1) * 100:+.1f} % against one context; {sum(out[r][2] for r in range(n_ranks))} warnings)",
          flush=True)
    for r in range(n_ranks):
        lib.ryujin_hip_comm_destroy(C.c_void_p(comms[r]))

import ctypes as C, threading, sys
import numpy as np
sys.path.insert(0, '.')
from ryujin_amd import HyperbolicModule, capi, offline
from ryujin_amd.initial_states import euler_uniform
lib = capi.load_hip()
cpu, n_steps = 200, 40
def run(off, comm, out, key):
    try:
        m = HyperbolicModule(off, equation=capi.EQ_EULER, backend="hip", comm=comm)
        m.cfl = 0.9
        U0 = euler_uniform(off.positions)
        U0 *= 1.0 + 1e-3 * np.sin(7.0 * off.positions[:, :1] + 3.0 * off.positions[:, 1:2])
        d = euler_uniform(off.b_positions) if off.n_bdry else None
        state = m.new_state_vector(U0); temps = [m.new_state_vector() for _ in range(3)]
        taus = [m.time_step("ssprk 33", state, temps, d if k == 0 else None) for k in range(n_steps)]
        out[key] = (off.global_ids[:off.n_owned].astype(np.int64), state.download()[:off.n_owned], taus)
    except Exception as e:
        out[key] = e
ref = {}
run(offline.SyntheticOffline(offline.mach3_step_2d(cpu)), None, ref, 0)
gid, U, taus = ref[0]
for n_ranks in (2, 4, 7):
    comms = (C.c_void_p * n_ranks)()
    assert lib.ryujin_hip_comm_init_local(comms, n_ranks, 0) == 0
    parts = [offline.SyntheticOffline(offline.mach3_step_2d(cpu, n_ranks=n_ranks, rank=r)) for r in range(n_ranks)]
    out = {}
    th = [threading.Thread(target=run, args=(parts[r], C.c_void_p(comms[r]), out, r)) for r in range(n_ranks)]
    [t.start() for t in th]; [t.join(timeout=600) for t in th]
    for r in range(n_ranks):
        assert not isinstance(out[r], Exception), out[r]
    g = np.concatenate([out[r][0] for r in range(n_ranks)]); Up = np.concatenate([out[r][1] for r in range(n_ranks)])
    o1, o2 = np.argsort(gid), np.argsort(g)
    err = (np.abs(Up[o2] - U[o1]) / np.abs(U).max(axis=0)).max()
    dt = max(abs(a - b) / b for a, b in zip(out[0][2], taus))
    print(n_ranks, "ranks: max rel U err", err, "max rel tau err", dt, flush=True)
    for r in range(n_ranks):
        lib.ryujin_hip_comm_destroy(C.c_void_p(comms[r]))

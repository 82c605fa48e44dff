"""Wall time per forward-Euler update of the device-resident RK driver for different schemes (timers off).
Usage: scheme_timing.py <dim> <size> [aeos]"""
import sys, time

sys.path.insert(0, '.')
from ryujin_amd import HyperbolicModule, capi, offline
from ryujin_amd.initial_states import euler_uniform, euler_radial_contrast
dim, size = int(sys.argv[1]), int(sys.argv[2])
equation = capi.EQ_EULER_AEOS if (len(sys.argv) > 3 and sys.argv[3] == "aeos") else capi.EQ_EULER
if dim == 2:
    off = offline.SyntheticOffline(offline.mach3_step_2d(size))
    U0 = euler_uniform(off.positions); d = euler_uniform(off.b_positions)
else:
    off = offline.SyntheticOffline(offline.box_3d(size))
    U0 = euler_radial_contrast(off.positions, inner=(1.0, 0.0, 100.0), outer=(1.0, 0.0, 0.1), radius=0.1); d = None
lib = capi.load_hip()
for scheme, stages in (("ssprk 33", 3), ("erk 33", 3), ("erk 54", 5)):
    m = HyperbolicModule(off, equation=equation, backend="hip")
    m.cfl = 0.9
    state = m.new_state_vector(U0); temps = [m.new_state_vector() for _ in range(5)]
    m.time_step(scheme, state, temps, d)
    for _ in range(20):
        m.time_step(scheme, state, temps)
    lib.ryujin_hip_synchronize(m._ctx)
    n = 12
    t0 = time.perf_counter()
    for _ in range(n):
        m.time_step(scheme, state, temps)
    lib.ryujin_hip_synchronize(m._ctx)
    dt = time.perf_counter() - t0
    print(f"dim={dim} n_q={off.n_owned} {scheme}: {dt / (stages * n) * 1e3:.4f} ms/update", flush=True)
    del m

"""Host-side mirror of ryujin::HyperbolicModule / ryujin::TimeIntegrator over the C ABI.

Same member names, argument meaning and error behaviour as the reference
(source/hyperbolic_module.h:110-278, source/time_integrator.h:279-302) so that the
parity tests read like the reference's own. The production shim is the C++ header
ryujin_amd/csrc/hyperbolic_module_shim.hpp; this Python twin exists for pytest/bench.py.

The class is backend-agnostic: `backend="hip"` binds libryujin_hip.so (the product);
tests may pass a (lib, prefix) pair for the CPU oracle, which exports the same call
surface. There is NO silent fallback: a missing HIP library raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class Restart(Exception):
    """Mirror of ryujin::Restart (source/hyperbolic_module.h:49-57)."""


class TauError(RuntimeError):
    """tau_max is NaN/inf/<=0 (AssertThrow at hyperbolic_module.template.h:573-576)."""


class StateVector:
    """Device-resident (U, precomputed) pair behind an integer handle (source/state_vector.h:47-51)."""

    def __init__(self, module: "HyperbolicModule"):
        self.module = module
        h = C.c_int(-1)
        module._check(module._f("state_alloc")(module._ctx, C.byref(h)))
        self.handle = h.value

    def upload(self, U: np.ndarray) -> None:
        U = np.ascontiguousarray(U, dtype=np.float64)
        assert U.size == self.module.n_relevant * self.module.k
        self.module._check(self.module._f("state_upload")(
            self.module._ctx, self.handle, capi.as_ptr(U, capi.c_double_p)))

    def download(self) -> np.ndarray:
        U = np.empty((self.module.n_relevant, self.module.k), dtype=np.float64)
        self.module._check(self.module._f("state_download")(
            self.module._ctx, self.handle, capi.as_ptr(U, capi.c_double_p)))
        return U

    def download_precomputed(self) -> np.ndarray:
        P = np.empty((self.module.n_relevant, self.module.n_prec), dtype=np.float64)
        self.module._check(self.module._f("state_download_precomputed")(
            self.module._ctx, self.handle, capi.as_ptr(P, capi.c_double_p)))
        return P

    def free(self) -> None:
        if self.handle >= 0 and self.module._ctx:
            self.module._f("state_free")(self.module._ctx, self.handle)
        self.handle = -1


class HyperbolicModule:
    # run-time switches of the library (ryujin_hip_params::system_scope_events / debug_*) applied to every module
    # constructed while set: the parity suite re-runs whole test functions through the branches a small mesh does
    # not take by itself (monkeypatch.setattr(HyperbolicModule, "library_switches", {...}))
    library_switches: dict = {}

    def __init__(self, offline, params: capi.Params | None = None, *, equation=capi.EQ_EULER,
                 backend="hip", comm=None, device: int = 0):
        if backend == "hip":
            self._lib, self._prefix = capi.load_hip(), "ryujin_hip_"
        else:
            self._lib, self._prefix = backend  # (ctypes lib, prefix): the test oracle
        self.offline = offline
        self.dim = offline.dim
        if params is None:
            params = self.default_params(equation, self.dim)
        for name, value in type(self).library_switches.items():
            setattr(params, name, value)
        self.params = params
        self.equation = params.equation
        # problem_dimension, n_precomputed_values, Limiter::n_bounds of the Description
        self.k, self.n_prec, self.n_bounds = {
            capi.EQ_EULER: (self.dim + 2, 2, 3),            # source/euler/{hyperbolic_system,limiter}.h
            capi.EQ_SHALLOW_WATER: (self.dim + 1, 2, 5),    # source/shallow_water/...
            capi.EQ_EULER_AEOS: (self.dim + 2, 4, 4),       # source/euler_aeos/...
            capi.EQ_SCALAR_CONSERVATION: (1, 2 * self.dim, 2),  # source/scalar_conservation/...
        }[self.equation]
        self.n_owned, self.n_relevant = offline.n_owned, offline.n_relevant
        self._ctx = C.c_void_p()
        self._comm = comm
        rc = self._f("create")(C.byref(self._ctx), offline.c, C.byref(params),
                               comm if comm is not None else None, device)
        self._check(rc)

    # ------------------------------------------------------------------ plumbing
    def _f(self, name):
        return getattr(self._lib, self._prefix + name)

    def _check(self, rc: int) -> int:
        if rc < 0:
            msg = self._f("last_error")()
            raise RuntimeError(f"{self._prefix}* failed with status {rc}: {msg.decode() if msg else ''}")
        return rc

    def default_params(self, equation, dim) -> capi.Params:
        p = capi.Params()
        self._f("default_params")(C.byref(p), equation, dim)
        return p

    def close(self):
        if getattr(self, "_ctx", None):
            self._f("destroy")(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ reference API
    def new_state_vector(self, U: np.ndarray | None = None) -> StateVector:
        sv = StateVector(self)
        if U is not None:
            sv.upload(U)
        return sv

    def prepare_state_vector(self, state: StateVector, t: float, dirichlet: np.ndarray | None = None):
        ptr = None
        if dirichlet is not None:
            dirichlet = np.ascontiguousarray(dirichlet, dtype=np.float64)
            assert dirichlet.size == self.offline.n_bdry * self.k
            ptr = capi.as_ptr(dirichlet, capi.c_double_p)
        self._check(self._f("prepare_state_vector")(self._ctx, state.handle, float(t), ptr))

    def step(self, old: StateVector, stage_state_vectors, stage_weights, new: StateVector,
             tau: float = 0.0, tau_max: float = np.finfo(np.float64).max) -> float:
        """step<stages>(): returns the tau used; raises Restart like the reference."""
        stages = len(stage_state_vectors)
        assert stages == len(stage_weights)
        hs = (C.c_int * max(stages, 1))(*[s.handle for s in stage_state_vectors])
        ws = (C.c_double * max(stages, 1))(*[float(w) for w in stage_weights])
        tau_out = C.c_double(0.0)
        rc = self._f("step")(self._ctx, old.handle, stages, hs, ws, new.handle, float(tau),
                             float(tau_max), C.byref(tau_out))
        if rc == capi.RYUJIN_ERR_TAU:
            raise TauError("I'm sorry, Dave. I'm afraid I can't do that. We crashed.")
        self._check(rc)
        self.last_status = rc
        if rc == capi.RYUJIN_RESTART:
            raise Restart()
        return tau_out.value

    def time_step(self, scheme: str, state: StateVector, temps, dirichlet=None, tau_max=None,
                  cfl_recovery="none", cfl_min=0.45, cfl_max=0.90, t=0.0, dirichlet_fn=None) -> float:
        """Device-resident TimeIntegrator::step (ryujin_hip_time_step): one host synchronisation per
        RK step. `state` names the solution before and after the call. dirichlet_fn(time) -> [n_bdry, k]:
        time-dependent Dirichlet data, evaluated by the library at the stage times t + c_s tau
        (ryujin_hip_time_step_fn)."""
        if dirichlet_fn is not None:
            return self._time_step_fn(scheme, state, temps, t, dirichlet_fn, tau_max, cfl_recovery, cfl_min, cfl_max)
        schemes = {"ssprk 22": capi.SCHEME_SSPRK_22, "ssprk 33": capi.SCHEME_SSPRK_33,
                   "erk 11": capi.SCHEME_ERK_11, "erk 22": capi.SCHEME_ERK_22, "erk 33": capi.SCHEME_ERK_33,
                   "erk 43": capi.SCHEME_ERK_43, "erk 54": capi.SCHEME_ERK_54}
        ptr = None
        if dirichlet is not None:
            dirichlet = np.ascontiguousarray(dirichlet, dtype=np.float64)
            ptr = capi.as_ptr(dirichlet, capi.c_double_p)
        hs = (C.c_int * len(temps))(*[t.handle for t in temps])
        tau = C.c_double(0.0)
        rc = self._f("time_step_n")(self._ctx, schemes[scheme], state.handle, len(temps), hs, ptr,
                                  float(np.finfo(np.float64).max if tau_max is None else tau_max),
                                  capi.CFL_RECOVERY_BANG_BANG if cfl_recovery == "bang bang control"
                                  else capi.CFL_RECOVERY_NONE, float(cfl_min), float(cfl_max), C.byref(tau))
        if rc == capi.RYUJIN_ERR_TAU:
            raise TauError("I'm sorry, Dave. I'm afraid I can't do that. We crashed.")
        self._check(rc)
        self.last_status = rc
        if rc == capi.RYUJIN_RESTART:
            raise Restart()
        return tau.value

    SCHEMES = {"ssprk 22": capi.SCHEME_SSPRK_22, "ssprk 33": capi.SCHEME_SSPRK_33, "erk 11": capi.SCHEME_ERK_11,
               "erk 22": capi.SCHEME_ERK_22, "erk 33": capi.SCHEME_ERK_33, "erk 43": capi.SCHEME_ERK_43,
               "erk 54": capi.SCHEME_ERK_54}

    def _time_step_fn(self, scheme, state, temps, t, dirichlet_fn, tau_max, cfl_recovery, cfl_min, cfl_max):
        n = self.offline.n_bdry * self.k

        failure = []

        def callback(user, time, out):
            # ctypes prints and swallows an exception raised inside a callback; the library has zero-filled the
            # stage's Dirichlet data, so the stage would run on -- silently -- with zeros. Keep the exception, poison
            # the values (NaN: the step cannot pass unnoticed) and re-raise once the call has returned.
            try:
                values = np.ascontiguousarray(dirichlet_fn(time), dtype=np.float64).reshape(-1)
                if values.size != n:
                    raise ValueError(f"dirichlet_fn returned {values.size} values, expected {n}")
                C.memmove(out, values.ctypes.data, n * 8)
            except Exception as e:  # noqa: BLE001
                failure.append(e)
                nan = np.full(n, np.nan)
                C.memmove(out, nan.ctypes.data, n * 8)
            except BaseException as e:  # KeyboardInterrupt, SystemExit: poison the data as well, re-raised first
                failure.insert(0, e)
                nan = np.full(n, np.nan)
                C.memmove(out, nan.ctypes.data, n * 8)
        cb = capi.DIRICHLET_FN(callback)
        hs = (C.c_int * len(temps))(*[x.handle for x in temps])
        tau = C.c_double(0.0)
        rc = self._f("time_step_fn")(self._ctx, self.SCHEMES[scheme], state.handle, len(temps), hs, float(t), cb,
                                     None, float(np.finfo(np.float64).max if tau_max is None else tau_max),
                                     capi.CFL_RECOVERY_BANG_BANG if cfl_recovery == "bang bang control"
                                     else capi.CFL_RECOVERY_NONE, float(cfl_min), float(cfl_max), C.byref(tau))
        if failure:
            raise failure[0]
        if rc == capi.RYUJIN_ERR_TAU:
            raise TauError("I'm sorry, Dave. I'm afraid I can't do that. We crashed.")
        self._check(rc)
        self.last_status = rc
        if rc == capi.RYUJIN_RESTART:
            raise Restart()
        return tau.value

    def integrals(self, state: StateVector) -> np.ndarray:
        """sum_i m_i U_i over the owned DoFs of all ranks, computed on the device with a fixed
        summation order (ryujin_hip_state_integrals; device backend only)."""
        out = np.zeros(self.k, dtype=np.float64)
        self._check(self._f("state_integrals")(self._ctx, state.handle, capi.as_ptr(out, capi.c_double_p)))
        return out

    def sadd(self, dst: StateVector, s: float, b: float, src: StateVector):
        self._check(self._f("sadd")(self._ctx, dst.handle, float(s), float(b), src.handle))

    @property
    def cfl(self) -> float:
        v = C.c_double()
        self._check(self._f("get_cfl")(self._ctx, C.byref(v)))
        return v.value

    @cfl.setter
    def cfl(self, value: float):
        self._check(self._f("set_cfl")(self._ctx, float(value)))

    @property
    def id_violation_strategy(self):
        return self._idv if hasattr(self, "_idv") else self.params.id_violation_strategy

    @id_violation_strategy.setter
    def id_violation_strategy(self, s: int):
        self._idv = s
        self._check(self._f("set_id_violation_strategy")(self._ctx, int(s)))

    def alpha(self) -> np.ndarray:
        a = np.empty(self.n_relevant, dtype=np.float64)
        self._check(self._f("get_alpha")(self._ctx, capi.as_ptr(a, capi.c_double_p)))
        return a

    def _counters(self):
        r, w = C.c_uint(), C.c_uint()
        self._check(self._f("get_counters")(self._ctx, C.byref(r), C.byref(w)))
        return r.value, w.value

    def n_restarts(self) -> int:
        return self._counters()[0]

    def n_warnings(self) -> int:
        return self._counters()[1]

    # ------------------------------------------------------------------ introspection
    def exchange_info(self) -> dict:
        """Neighbour ranks of this context and the number of ghost exchanges / scalar all-reduces it has
        enqueued so far (ryujin_hip_exchange_info; device backend only)."""
        n_nbr, nx, nr = C.c_int(0), C.c_ulonglong(0), C.c_ulonglong(0)
        nbr = (C.c_int * 64)()
        self._check(self._f("exchange_info")(self._ctx, C.byref(n_nbr), nbr, 64, C.byref(nx), C.byref(nr)))
        return dict(neighbours=[nbr[q] for q in range(min(n_nbr.value, 64))], n_exchanges=nx.value,
                    n_allreduces=nr.value)

    def limiter_statistics(self) -> dict:
        """Fraction of the 64-row slices in which the first high-order sweep found a limited pair (between the two
        latest host synchronisations), how the latest step kept P_ij ("everywhere" / "per slice") and the fraction
        of slices it was stored in (ryujin_hip_limiter_statistics; device backend only)."""
        f, stored, fs = C.c_double(1.0), C.c_int(1), C.c_double(1.0)
        fn = self._f("limiter_statistics")
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]
        self._check(fn(self._ctx, C.byref(f), C.byref(stored), C.byref(fs)))
        out = dict(limited_slice_fraction=f.value,
                   pij_stored={1: "everywhere", 2: "per slice", 3: "per tile"}.get(stored.value),
                   pij_stored_slice_fraction=fs.value)
        if stored.value == 3:
            # of the (slice, column) tiles: stored by step 5, read by step 6, formed by step 6 (ryujin_hip_tile_statistics)
            a, b, c = C.c_double(1.0), C.c_double(1.0), C.c_double(0.0)
            self._check(self._lib.ryujin_hip_tile_statistics(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
            out.update(tiles_stored_fraction=a.value, tiles_read_fraction=b.value, tiles_formed_by_step6_fraction=c.value)
            if hasattr(self._lib, "ryujin_hip_deferred_slices"):  # (A/B runs load older builds of the library)
                n = C.c_uint(0)
                self._lib.ryujin_hip_deferred_slices.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
                self._check(self._lib.ryujin_hip_deferred_slices(self._ctx, C.byref(n)))
                out.update(deferred_slices_last_update=n.value)
        return out

    def layout_info(self) -> dict:
        """tiles (64-entry columns of the SELL-64 slices) of the owned rows and how many the tile map serves from a
        16-byte descriptor (ryujin_hip_layout_info; device backend only)"""
        n, r = C.c_ulonglong(0), C.c_ulonglong(0)
        self._check(self._lib.ryujin_hip_layout_info(self._ctx, C.byref(n), C.byref(r)))
        ch, ce = C.c_ulonglong(0), C.c_ulonglong(0)
        if hasattr(self._lib, "ryujin_hip_chain_info"):  # (A/B libraries of earlier trees do not have it)
            self._lib.ryujin_hip_chain_info.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
            self._check(self._lib.ryujin_hip_chain_info(self._ctx, C.byref(ch), C.byref(ce)))
        return dict(n_tiles=n.value, n_regular_tiles=r.value,
                    regular_tile_fraction=(r.value / n.value if n.value else 0.0),
                    n_chained_tiles=ch.value, chained_tile_fraction=(ch.value / n.value if n.value else 0.0),
                    chained_entry_fraction=(ce.value / (64.0 * n.value) if n.value else 0.0))

    def debug_fetch(self, what: str) -> np.ndarray:
        """`*_all`: over all locally relevant rows, i.e. including the ghost rows / ghost range received from
        the neighbour ranks."""
        codes = {"dij": 0, "lij": 1, "pij": 2, "bounds": 3, "r": 4, "lij_next": 5,
                 "dij_all": 6, "lij_all": 7, "lij_next_all": 8, "r_all": 9}
        rs = self.offline.row_starts
        nnz_owned = int(rs[self.n_owned])
        nnz_all = int(rs[self.n_relevant])
        sizes = {"dij": nnz_owned, "lij": nnz_owned, "pij": nnz_owned * self.k,
                 "bounds": self.n_owned * self.n_bounds, "r": self.n_owned * self.k,
                 "lij_next": nnz_owned, "dij_all": nnz_all, "lij_all": nnz_all, "lij_next_all": nnz_all,
                 "r_all": self.n_relevant * self.k}
        out = np.empty(sizes[what], dtype=np.float64)
        self._check(self._f("debug_fetch")(self._ctx, codes[what], capi.as_ptr(out, capi.c_double_p),
                                           out.size))
        return out


class DeviceResidentTimeIntegrator:
    """TimeIntegrator::step(state_vector, t, t_final) executed inside the library (ryujin_hip_time_step_fn): the
    interface of TimeIntegrator below, one host synchronisation per Runge-Kutta step, Dirichlet data evaluated at
    the stage times by a callback (what contrib/hyperbolic_module_hip.h::time_step does on the ryujin side)."""

    def __init__(self, module: "HyperbolicModule", scheme: str = "erk 33", cfl_min=0.45, cfl_max=0.90,
                 cfl_recovery_strategy: str = "bang bang control", dirichlet_fn=None):
        self.m, self.scheme = module, scheme
        self.cfl_min, self.cfl_max, self.cfl_recovery_strategy = cfl_min, cfl_max, cfl_recovery_strategy
        self.dirichlet_fn = dirichlet_fn
        self.temp = [module.new_state_vector() for _ in range({"erk 43": 4, "erk 54": 5}.get(scheme, 3))]
        module.cfl = cfl_max

    def step(self, state: StateVector, t: float, t_final: float = np.finfo(np.float64).max):
        tau = self.m.time_step(self.scheme, state, self.temp, tau_max=t_final - t,
                               cfl_recovery=self.cfl_recovery_strategy, cfl_min=self.cfl_min, cfl_max=self.cfl_max,
                               t=t, dirichlet_fn=self.dirichlet_fn)
        return state, tau


class TimeIntegrator:
    """Explicit schemes of ryujin::TimeIntegrator that consist solely of
    prepare_state_vector + step<s> + sadd (source/time_integrator.template.h:279-470)."""

    def __init__(self, module: HyperbolicModule, scheme: str = "erk 33", cfl_min=0.45, cfl_max=0.90,
                 cfl_recovery_strategy: str = "bang bang control", dirichlet_fn=None):
        self.m = module
        self.scheme = scheme
        self.cfl_min, self.cfl_max = cfl_min, cfl_max
        self.cfl_recovery_strategy = cfl_recovery_strategy
        self.dirichlet_fn = dirichlet_fn  # t -> [n_bdry, k] array or None
        # temp_ vectors per scheme: TimeIntegrator::prepare() (:163-205)
        n_temp = {"erk 43": 4, "erk 54": 5}.get(scheme, 3)
        self.temp = [module.new_state_vector() for _ in range(n_temp)]
        # TimeIntegrator::prepare(): hyperbolic_module_->cfl(cfl_max_)  (:150)
        module.cfl = cfl_max

    def _prepare(self, sv, t):
        self.m.prepare_state_vector(sv, t, self.dirichlet_fn(t) if self.dirichlet_fn else None)

    def step(self, state: StateVector, t: float, t_final: float = np.finfo(np.float64).max):
        """Returns (new_state, tau). The handles are swapped like state_vector.swap(temp)."""
        tau_max = t_final - t
        single = {"ssprk 22": self._ssprk22, "ssprk 33": self._ssprk33, "erk 11": self._erk11,
                  "erk 22": self._erk22, "erk 33": self._erk33, "erk 43": self._erk43,
                  "erk 54": self._erk54}[self.scheme]
        if self.cfl_recovery_strategy == "bang bang control":
            self.m.id_violation_strategy = capi.IDV_RAISE_EXCEPTION
            self.m.cfl = self.cfl_max
        try:
            return single(state, t, tau_max)
        except Restart:
            if self.cfl_recovery_strategy == "none":
                raise
            self.m.id_violation_strategy = capi.IDV_WARN
            self.m.cfl = self.cfl_min
            return single(state, t, tau_max)

    def _swap(self, state, idx):
        new = self.temp[idx]
        self.temp[idx] = state
        return new

    def _erk11(self, sv, t, tau_max):
        self._prepare(sv, t)
        tau = self.m.step(sv, [], [], self.temp[0], 0.0, tau_max)
        return self._swap(sv, 0), tau

    def _ssprk22(self, sv, t, tau_max):
        T = self.temp
        self._prepare(sv, t)
        tau = self.m.step(sv, [], [], T[0], 0.0, tau_max)
        self._prepare(T[0], t + 1.0 * tau)
        self.m.step(T[0], [], [], T[1], tau)
        self.m.sadd(T[1], 1.0 / 2.0, 1.0 / 2.0, sv)
        return self._swap(sv, 1), tau

    def _ssprk33(self, sv, t, tau_max):
        T = self.temp
        self._prepare(sv, t)
        tau = self.m.step(sv, [], [], T[0], 0.0, tau_max)
        self._prepare(T[0], t + 1.0 * tau)
        self.m.step(T[0], [], [], T[1], tau)
        self.m.sadd(T[1], 1.0 / 4.0, 3.0 / 4.0, sv)
        self._prepare(T[1], t + 0.5 * tau)
        self.m.step(T[1], [], [], T[0], tau)
        self.m.sadd(T[0], 2.0 / 3.0, 1.0 / 3.0, sv)
        return self._swap(sv, 0), tau

    def _erk22(self, sv, t, tau_max):
        T = self.temp
        self._prepare(sv, t)
        tau = self.m.step(sv, [], [], T[0], 0.0, tau_max / 2.0)
        self._prepare(T[0], t + 1.0 * tau)
        self.m.step(T[0], [sv], [-1.0], T[1], tau)
        return self._swap(sv, 1), 2.0 * tau

    def _erk33(self, sv, t, tau_max):
        T = self.temp
        self._prepare(sv, t)
        tau = self.m.step(sv, [], [], T[0], 0.0, tau_max / 3.0)
        self._prepare(T[0], t + 1.0 * tau)
        self.m.step(T[0], [sv], [-1.0], T[1], tau)
        self._prepare(T[1], t + 2.0 * tau)
        self.m.step(T[1], [sv, T[0]], [0.75, -2.0], T[2], tau)
        return self._swap(sv, 2), 3.0 * tau

    def _erk43(self, sv, t, tau_max):
        """step_erk_43 (time_integrator.template.h:405-440)"""
        T = self.temp
        self._prepare(sv, t)
        tau = self.m.step(sv, [], [], T[0], 0.0, tau_max / 4.0)
        self._prepare(T[0], t + 1.0 * tau)
        self.m.step(T[0], [sv], [-1.0], T[1], tau)
        self._prepare(T[1], t + 2.0 * tau)
        self.m.step(T[1], [T[0]], [-1.0], T[2], tau)
        self._prepare(T[2], t + 3.0 * tau)
        self.m.step(T[2], [T[0], T[1]], [5.0 / 3.0, -10.0 / 3.0], T[3], tau)
        return self._swap(sv, 3), 4.0 * tau

    ERK54 = dict(c=0.2, a_21=+0.2, a_31=+0.26075582269554909, a_32=+0.13924417730445096,
                 a_41=-0.25856517872570289, a_42=+0.91136274166280729, a_43=-0.05279756293710430,
                 a_51=+0.21623276431503774, a_52=+0.51534223099602405, a_53=-0.81662794199265554,
                 a_54=+0.88505294668159373, a_61=-0.10511678454691901, a_62=+0.87880047152100838,
                 a_63=-0.58903404061484477, a_64=+0.46213380485434047)

    def _erk54(self, sv, t, tau_max):
        """step_erk_54 (time_integrator.template.h:443-510)"""
        T = self.temp
        a = self.ERK54
        c = a["c"]
        self._prepare(sv, t)
        tau = self.m.step(sv, [], [], T[0], 0.0, tau_max / 5.0)
        self._prepare(T[0], t + 1.0 * tau)
        self.m.step(T[0], [sv], [(a["a_31"] - a["a_21"]) / c], T[1], tau)
        self._prepare(T[1], t + 2.0 * tau)
        self.m.step(T[1], [sv, T[0]], [(a["a_41"] - a["a_31"]) / c, (a["a_42"] - a["a_32"]) / c], T[2], tau)
        self._prepare(T[2], t + 3.0 * tau)
        self.m.step(T[2], [sv, T[0], T[1]],
                    [(a["a_51"] - a["a_41"]) / c, (a["a_52"] - a["a_42"]) / c, (a["a_53"] - a["a_43"]) / c],
                    T[3], tau)
        self._prepare(T[3], t + 4.0 * tau)
        self.m.step(T[3], [sv, T[0], T[1], T[2]],
                    [(a["a_61"] - a["a_51"]) / c, (a["a_62"] - a["a_52"]) / c, (a["a_63"] - a["a_53"]) / c,
                     (a["a_64"] - a["a_54"]) / c], T[4], tau)
        return self._swap(sv, 4), 5.0 * tau


class HostStateVector:
    """A HOST state vector as the reference's caller owns it (source/state_vector.h:47-51): U [n_relevant, k] and
    the precomputed values [n_relevant, n_prec] in numpy arrays. swap() exchanges the storage of two of them, as
    std::tuple::swap / dealii Vector::swap do (time_integrator.template.h:296,325)."""

    def __init__(self, module: "HyperbolicModule", U: np.ndarray | None = None):
        self.U = np.zeros((module.n_relevant, module.k)) if U is None else np.array(U, dtype=np.float64, copy=True)
        self.precomputed = np.zeros((module.n_relevant, module.n_prec))

    def swap(self, other: "HostStateVector"):
        self.U, other.U = other.U, self.U
        self.precomputed, other.precomputed = other.precomputed, self.precomputed


class HostMirroredModule:
    """Python twin of the mirroring of contrib/hyperbolic_module_hip.h: what the adapter does when an UNMODIFIED
    TimeIntegrator (host sadd()/swap(), time_integrator.template.h:18-25,279-510) drives it. Device twins are keyed
    by the storage of U (the data pointer), the host array is the authority at every call, and only what a call can
    have changed is written back (ryujin_hip_state_download_prepared / _owned). Used by the parity tests of those
    entry points and by bench.py's host-mirrored line; device library only."""

    def __init__(self, module: HyperbolicModule, pin: bool = True, mirror_derived: bool = True):
        self.m = module
        self.pin, self.mirror_derived = pin, mirror_derived
        self.twins: dict[int, StateVector] = {}
        self._pinned: list[np.ndarray] = []  # kept alive while registered, unregistered by close()
        self.alpha = np.zeros(module.n_relevant)
        if pin:
            self._register(self.alpha)

    def _register(self, a: np.ndarray):
        self.m._check(self.m._lib.ryujin_hip_host_register(self.m._ctx, a.ctypes.data, a.nbytes))
        self._pinned.append(a)

    def twin_of(self, sv: HostStateVector) -> StateVector:
        key = sv.U.ctypes.data
        if key not in self.twins:
            self.twins[key] = StateVector(self.m)
            if self.pin:
                self._register(sv.U)
                self._register(sv.precomputed)
        return self.twins[key]

    def prepare_state_vector(self, sv: HostStateVector, t: float, dirichlet: np.ndarray | None = None):
        m, twin = self.m, self.twin_of(sv)
        twin.upload(sv.U)
        m.prepare_state_vector(twin, t, dirichlet)
        m._check(m._lib.ryujin_hip_state_download_prepared(m._ctx, twin.handle, capi.as_ptr(sv.U, capi.c_double_p)))
        if self.mirror_derived:
            m._check(m._lib.ryujin_hip_state_download_precomputed(m._ctx, twin.handle,
                                                                  capi.as_ptr(sv.precomputed, capi.c_double_p)))

    def step(self, old: HostStateVector, stage_state_vectors, stage_weights, new: HostStateVector,
             tau: float = 0.0, tau_max: float = np.finfo(np.float64).max) -> float:
        m = self.m
        for sv in [old, *stage_state_vectors]:
            assert sv.U.ctypes.data in self.twins, "old and stage state vectors have to be prepared"
        twin_new = self.twin_of(new)
        try:
            return m.step(self.twins[old.U.ctypes.data], [self.twins[s.U.ctypes.data] for s in stage_state_vectors],
                          stage_weights, twin_new, tau, tau_max)
        finally:
            m._check(m._lib.ryujin_hip_state_download_owned(m._ctx, twin_new.handle, capi.as_ptr(new.U, capi.c_double_p)))
            if self.mirror_derived:
                m._check(m._lib.ryujin_hip_get_alpha(m._ctx, capi.as_ptr(self.alpha, capi.c_double_p)))

    def close(self):
        for twin in self.twins.values():
            twin.free()
        self.twins.clear()
        for a in self._pinned:
            self.m._lib.ryujin_hip_host_unregister(self.m._ctx, a.ctypes.data)
        self._pinned.clear()

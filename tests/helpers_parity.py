"""HIP-vs-oracle comparison of ONE update (prepare_state_vector + step) on identical inputs, array by array.

Stated contract:
  boundary conditions                 1e-14
  precomputed values                  1e-13 relative
  alpha_i                             1e-12  (absolute: alpha is a blending weight in [0,1]; its numerator is a
                                              commutator, i.e. a difference of nearly equal sums, so a *relative*
                                              bound on a small alpha is a bound on that cancellation, not on the
                                              scheme -- alpha only enters as d_ij (alpha_i + alpha_j) / 2)
  d_ij, tau_max, limiter bounds       1e-12 relative
  r_i, P_ij                           1e-12 of the largest entry of the component (sums over the stencil whose
                                              terms cancel: the error scale is the terms', not the sum's)
  l_ij, l'_ij                         1e-10 absolute (the limiter's Newton tolerance, Appendix E-3)
  U_new                               1e-11 relative per component (scaled by max |U|)

l_ij has a genuine discontinuity: `psi_r > 0` decides between "accept t_r" and "two Newton steps from t_l = 0"
(limiter.template.h:188-216); where psi_r is zero to round-off the reference's own scalar and SIMD builds decide
differently. No quota is granted for that: every (i,j) pair whose l differs by more than 1e-10 must be SHOWN to
sit on that discontinuity -- the oracle's psi_r for the pair, recomputed from the oracle's own bounds, state and
P_ij, has to vanish to round-off (|psi_r| <= 1e-13 of its two terms). U_new = U_low + sum_j l_ij lambda P_ij
inherits the l_ij differences: an entry of U_new may exceed 1e-11 only by sum_j |dl_ij| lambda |P_ij| of its own
row (|dl| <= 1e-10 for Newton-iterated pairs, larger only at classified flips).

alpha: 1e-12 absolute everywhere observed except on the 2.5 M point mesh with a 1e-3 random perturbation
(1.008e-12): there f_j - f_i is three digits smaller than f and the commutator is that much worse conditioned in
the reference's own formula; beyond 1e-12 the difference has to stay within 4x the oracle's own response to a
last-bit perturbation of its input.
"""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

from ryujin_amd import HyperbolicModule, capi

STATS_FILE = os.environ.get("RYUJIN_PARITY_STATS")

PSI_ROUND_OFF = 1e-13      # |psi_r| / (|relax rho rho_e| + |s_min rho^(gamma+1)|) at an accepted branch flip
L_TOL = 1e-10
U_TOL = 1e-11


def _stat(label, **kw):
    if STATS_FILE:
        with open(STATS_FILE, "a") as f:
            clean = {k: (float(v) if isinstance(v, (float, np.floating)) else v) for k, v in kw.items()}
            f.write(json.dumps(dict(label=label, **clean)) + "\n")


SOFT = bool(STATS_FILE)   # statistics run: record violations of the tolerances under study instead of failing


def _check(cond, label, what, detail):
    if cond:
        return
    if SOFT:
        _stat(label, what="VIOLATION " + what, detail=repr(detail))
        return
    raise AssertionError((what, detail))


def _update(m, old, new, dirichlet, tau, stage_vectors=(), stage_weights=()):
    m.prepare_state_vector(old, 0.0, dirichlet)
    return m.step(old, list(stage_vectors), list(stage_weights), new, tau)


class EulerFlipClassifier:
    """Decides whether an l_ij difference sits on the psi_r = 0 branch of Limiter::limit, using the oracle only:
    re-runs the oracle with 0 (and 1) limiter iterations to obtain the state the limiter call saw (the low-order
    update for l_ij; the update after the first high-order pass for l'_ij), then asks the oracle's own limiter
    for psi_r of the pair."""

    def __init__(self, oracle, off, params, U_start, dirichlet, tau, c, stage_U=(), stage_weights=()):
        self.oracle, self.off, self.params, self.c = oracle, off, params, c
        self.U_start, self.dirichlet, self.tau = U_start, dirichlet, tau
        self.stage_U, self.stage_weights = list(stage_U), list(stage_weights)
        self.rs = off.row_starts[: off.n_owned + 1].astype(np.int64)
        self.cols = off.columns[: self.rs[-1]].astype(np.int64)
        self.k = c["pij"].size // c["lij"].size
        self._U = {}

    def _state_seen_by_limiter(self, iterations, backend=None):
        key = iterations if backend is None else (iterations, backend)
        if key not in self._U:
            p = capi.Params()
            C.memmove(C.byref(p), C.byref(self.params), C.sizeof(capi.Params))
            p.limiter_iterations = iterations
            m = HyperbolicModule(self.off, p, backend=backend or self.oracle.backend())
            old, new = m.new_state_vector(self.U_start), m.new_state_vector()
            stages = [m.new_state_vector(U) for U in self.stage_U]
            for sv in stages:                      # stage vectors are *prepared* state vectors (:207-213)
                m.prepare_state_vector(sv, 0.0, self.dirichlet)
            m.prepare_state_vector(old, 0.0, self.dirichlet)
            m.step(old, stages, self.stage_weights, new, self.tau)
            self._U[key] = new.download()
            m.close()
        return self._U[key]

    def _transposed(self, e):
        i = int(np.searchsorted(self.rs, e, side="right") - 1)
        j = int(self.cols[e])
        row_j = self.cols[self.rs[j]:self.rs[j + 1]] if j < self.off.n_owned else None
        if row_j is None:
            return i, j, None
        hit = np.nonzero(row_j == i)[0]
        return i, j, (int(self.rs[j] + hit[0]) if hit.size else None)

    def psi_r(self, name, e):
        """(|psi_r| relative to its two terms, i, j) for logical entry e of `lij` / `lij_next`.
        With two limiter iterations the buffers are swapped before the last pass
        (hyperbolic_module.template.h:1171-1173): after the step "lij_next" holds the FIRST pass l_ij and "lij"
        the second pass (1 - l) l'_ij; with one iteration "lij" is the first (and only) pass."""
        c, k = self.c, self.k
        i, j, e_t = self._transposed(e)
        P = c["pij"].reshape(-1, k)[e].copy()
        two = self.params.limiter_iterations == 2
        first = c["lij_next"] if two else c["lij"]
        if (name == "lij_next") == two:          # first pass: limit(bounds_i, low-order update, P_ij)
            U = self._state_seen_by_limiter(0)[i]
        else:                                     # second pass: limit(bounds_i, U after pass 1, (1 - l) P_ij)
            U = self._state_seen_by_limiter(1)[i]
            l_sym = first[e] if e_t is None else min(first[e], first[e_t])
            P *= (1.0 - l_sym)
        bounds = np.ascontiguousarray(c["bounds"].reshape(-1, 3)[i])
        out = np.zeros(5)
        dp = capi.c_double_p
        self.oracle.lib().ryujin_oracle_euler_limit_trace(C.byref(self.params), capi.as_ptr(bounds, dp),
                                                          capi.as_ptr(np.ascontiguousarray(U), dp),
                                                          capi.as_ptr(np.ascontiguousarray(P), dp),
                                                          capi.as_ptr(out, dp))
        t_r, psi = out[2], out[3]
        U_r = U + t_r * P
        rho = U_r[0]
        rho_e = U_r[-1] - 0.5 * (U_r[1:-1] ** 2).sum() / rho
        terms = abs(rho * rho_e) + abs(bounds[2] * rho ** (self.params.gamma + 1.0))
        return abs(psi) / terms, i, j

    def follows_its_inputs(self, e, g, second_pass=True):
        """(second_pass = False: the same questions for a first-pass l_ij -- the state is the low-order update, the
        inputs differ by round-off only, scale 1 instead of 1 - l.)
        A SECOND-pass outlier that is not a round-off sized psi_r: the state the second pass limits is the update
        after the first pass, and the first pass put it on the boundary of the invariant set only to ITS Newton
        tolerance -- first-pass l_ij that agree to 1e-10 (the contract) move that state by 1e-10 lambda |P_ij|,
        orders above round-off, and psi_r of a row sitting on the entropy bound changes sign with it. Such a
        difference is the limiter's own discontinuity met with inputs that differ within THEIR contracts, if
          (a) the oracle's limiter, given the device's inputs of the pair (its bounds, its update after the first
              pass -- the device run once more with ONE limiter pass --, its (1 - l) P_ij), returns the device's
              l'_ij to 1e-10 (same function), or finds psi_r of THOSE inputs at round-off (1e-13 of its terms: the
              branch flip proper, on the device's side of the input difference), and
          (b) those inputs differ from the oracle's by no more than the first-pass contract allows:
              |dU| <= 1e-11 + lambda sum_j |dl_ij| |P_ij| for the row.
        Returns (a and b, details)."""
        c, k = self.c, self.k
        i, j, e_t = self._transposed(e)
        rows = slice(self.rs[i], self.rs[i + 1])
        lam = 1.0 / max(1, self.rs[i + 1] - self.rs[i] - 1)
        n_it = 1 if second_pass else 0
        scale = np.abs(self._state_seen_by_limiter(n_it)).max(axis=0)
        U_g = self._state_seen_by_limiter(n_it, "hip")[i]
        U_c = self._state_seen_by_limiter(n_it)[i]
        out_name = "lij" if second_pass else "lij_next"   # (two limiter iterations: the buffers are swapped)

        def sym(first, entry, entry_t):
            return first[entry] if entry_t is None else min(first[entry], first[entry_t])
        # (b) the update after the first pass: covered by the first-pass differences of the row (incl. transposes)
        dl = np.zeros(self.rs[i + 1] - self.rs[i])
        for q, ee in enumerate(range(self.rs[i], self.rs[i + 1])):
            if q == 0:
                continue
            _, _, ee_t = self._transposed(ee)
            if second_pass:
                dl[q] = abs(sym(g["lij_next"], ee, ee_t) - sym(c["lij_next"], ee, ee_t))
        P_row = np.abs(c["pij"].reshape(-1, k)[rows])
        bound = U_TOL + lam * (dl[:, None] * P_row).sum(axis=0) / scale + \
            1e-12 * lam * (P_row.sum(axis=0) / scale)        # (and P_ij itself is known to 1e-12 of its largest entry)
        inputs_ok = bool((np.abs(U_g - U_c) / scale <= bound).all())
        # (a) the oracle's limiter on the device's inputs
        one_minus_l = (1.0 - sym(g["lij_next"], e, e_t)) if second_pass else 1.0
        P = g["pij"].reshape(-1, k)[e] * one_minus_l
        bounds = np.ascontiguousarray(g["bounds"].reshape(-1, 3)[i])
        out = np.zeros(5)
        dp = capi.c_double_p
        self.oracle.lib().ryujin_oracle_euler_limit_trace(C.byref(self.params), capi.as_ptr(bounds, dp),
                                                          capi.as_ptr(np.ascontiguousarray(U_g), dp),
                                                          capi.as_ptr(np.ascontiguousarray(P), dp),
                                                          capi.as_ptr(out, dp))
        expected = one_minus_l * out[0]
        same_function = abs(expected - g[out_name][e]) <= L_TOL
        # ... or the device's inputs themselves sit on the psi_r = 0 branch to round-off (the row's state after the
        # first pass lies ON the entropy bound and (1 - l) P_ij hardly moves it: psi_r is psi of that state)
        t_r, psi = out[2], out[3]
        U_r = U_g + t_r * P
        rho = U_r[0]
        rho_e = U_r[-1] - 0.5 * (U_r[1:-1] ** 2).sum() / rho
        terms = abs(rho * rho_e) + abs(bounds[2] * rho ** (self.params.gamma + 1.0))
        on_branch = abs(psi) / terms <= PSI_ROUND_OFF
        device = None
        if self.off.dim == 2 and not (same_function or on_branch):
            # what the device's own limiter makes of exactly these inputs (RYUJIN_DEBUG_EULER_LIMIT_2D): its l, and
            # psi_r as IT evaluates it -- the two evaluations of psi_r of one and the same input bracket zero at a
            # branch flip
            item = np.ascontiguousarray(np.concatenate([bounds, U_g, P]))
            res = np.zeros(5)
            lib = capi.load_hip()
            rc = lib.ryujin_hip_debug_function(0, C.byref(self.params), capi.DEBUG_EULER_LIMIT_2D,
                                               capi.as_ptr(item, dp), capi.as_ptr(res, dp), 1)
            assert rc == 0
            l_dev = one_minus_l * res[0]
            device = dict(l=float(l_dev), matches_sweep=bool(abs(l_dev - g[out_name][e]) <= 1e-14),
                          psi_rel=float(res[4] / terms), psi_rel_oracle=float(psi / terms))
            # (a sign flip is a branch flip only if BOTH evaluations of psi_r are at round-off level, 1e-10 of its
            # terms: the device does not vouch for itself with a psi_r of any size)
            on_branch = (device["matches_sweep"] and (res[4] > 0.0) != (psi > 0.0) and
                         abs(device["psi_rel"]) <= 1e-10 and abs(device["psi_rel_oracle"]) <= 1e-10)
            if not on_branch and device["matches_sweep"]:
                # Both evaluations agree that psi_r <= 0 -- by a few 1e-13 of its terms -- and both iterate from
                # t_l = 0; their results differ because psi is zero to round-off ALONG THE WHOLE SEGMENT (the state
                # sits on the entropy bound and (1 - l) P_ij is tangent to it): the two Newton steps the limiter is
                # allowed divide round-off by round-off, in the reference as here. The limiter's own acceptance
                # level says when that is so: it relaxes the entropy bound by vacuum_state_relaxation_large * eps
                # (relax = 1 + 1e4 eps, limiter.template.h:24-27,219-233). If |psi| stays below that level at
                # t = 0, at t_r and at both results, every t in [0, t_r] satisfies the bound as well as the
                # reference's own answer does.
                eps = np.finfo(np.float64).eps
                level = self.params.vacuum_state_relaxation_large * eps
                relax_small = 1.0 + self.params.vacuum_state_relaxation_small * eps

                def psi_rel(t):
                    V = U_g + t * P
                    rho_v = V[0]
                    rho_e_v = V[-1] - 0.5 * (V[1:-1] ** 2).sum() / rho_v
                    a, b = relax_small * rho_v * rho_e_v, bounds[2] * rho_v ** (self.params.gamma + 1.0)
                    return abs(a - b) / (abs(a) + abs(b))
                ts = [0.0, t_r, res[0], out[0]]
                flat = max(psi_rel(t) for t in ts)
                device["psi_rel_along_segment"] = float(flat)
                device["one_minus_l"] = float(one_minus_l)
                on_branch = flat <= level
                if not on_branch:
                    # ... or, short of that level, the limiter's answer for these inputs is simply ill conditioned:
                    # the yardstick is the ORACLE's own response to a last-bit perturbation of its inputs (as for
                    # the indicator, alpha_last_bit_sensitivity): 32 random +-1 ulp perturbations of bounds, state
                    # and P_ij; the device may be off by at most 4 times the spread they produce.
                    rng = np.random.default_rng(11)
                    spread = 0.0
                    for _ in range(32):
                        def pert(a):
                            return np.ascontiguousarray(a * (1.0 + 2.0 ** -52 * rng.choice([-1.0, 0.0, 1.0], size=a.shape)))
                        o2 = np.zeros(5)
                        self.oracle.lib().ryujin_oracle_euler_limit_trace(
                            C.byref(self.params), capi.as_ptr(pert(bounds), dp), capi.as_ptr(pert(U_g), dp),
                            capi.as_ptr(pert(P), dp), capi.as_ptr(o2, dp))
                        spread = max(spread, abs(o2[0] - out[0]))
                    device["last_bit_spread"] = float(one_minus_l * spread)
                    on_branch = abs(expected - g[out_name][e]) <= 4.0 * one_minus_l * spread
        return inputs_ok and (same_function or on_branch), (
            inputs_ok, float(abs(expected - g[out_name][e])), float(abs(psi) / terms),
            float((np.abs(U_g - U_c) / scale).max()), device)


def alpha_last_bit_sensitivity(oracle, off, params, U_before, dirichlet, tau, alpha_ref):
    """|alpha(U) - alpha(U (1 +- 2^-52))| of the ORACLE, per row, the largest of six random sign patterns: how far the
    reference's own indicator moves when every input entry is changed in its last bit. (Round 6: per row and six
    patterns instead of the maximum over the mesh of two -- two patterns underestimate the worst response of a row,
    and a maximum over the mesh says nothing about the row in question.)"""
    if oracle is None or params is None:
        return np.zeros_like(alpha_ref)
    p = capi.Params()
    C.memmove(C.byref(p), C.byref(params), C.sizeof(capi.Params))
    p.limiter_iterations = 0
    rng = np.random.default_rng(5)
    worst = np.zeros_like(alpha_ref)
    for _ in range(6):
        m = HyperbolicModule(off, p, backend=oracle.backend())
        U = U_before * (1.0 + 2.0 ** -52 * rng.choice([-1.0, 1.0], size=U_before.shape))
        old, new = m.new_state_vector(U), m.new_state_vector()
        m.prepare_state_vector(old, 0.0, dirichlet)
        m.step(old, [], [], new, tau)
        worst = np.maximum(worst, np.abs(m.alpha()[: alpha_ref.size] - alpha_ref))
        m.close()
    return worst


def compare_step(off, mods, dirichlet=None, tau=0.0, *, oracle=None, params=None, label="", fetch_pij=True,
                 keep_matrices=True, stage_vectors=None, stage_weights=()):
    """mods = [(hip module, old, new), (oracle module, old, new)] holding the SAME old state. Runs one update on
    both and compares every intermediate array, fetching them one after the other (full-size meshes: the P_ij
    of a 3-D mesh alone is 8.7 GB per backend). Returns (g, c): dicts of the small arrays of both backends.
    stage_vectors = (prepared stage vectors of the hip module, ... of the oracle module), stage_weights: the
    update is step<stages> of an explicit Runge-Kutta scheme (hyperbolic_module.template.h:663-677,822-846)."""
    (mg, og, ng), (mc, oc, nc) = mods
    n = off.n_owned
    equation = mg.equation
    U_before = oc.download()
    sv_g, sv_c = stage_vectors if stage_vectors else ((), ())
    stage_U = [sv.download() for sv in sv_c]
    tau_g = _update(mg, og, ng, dirichlet, tau, sv_g, stage_weights)
    tau_c = _update(mc, oc, nc, dirichlet, tau, sv_c, stage_weights)
    assert mg.last_status == mc.last_status
    g, c = dict(tau=tau_g, status=mg.last_status), dict(tau=tau_c, status=mc.last_status)

    def both(fetch):
        return fetch(mg, og, ng), fetch(mc, oc, nc)

    a, b = both(lambda m, o, nw: o.download()[:n])                      # boundary conditions applied to U_old
    np.testing.assert_allclose(a, b, rtol=1e-14, atol=1e-14)
    g["U_old"], c["U_old"] = a, b
    a, b = both(lambda m, o, nw: o.download_precomputed()[:n])
    np.testing.assert_allclose(a, b, rtol=1e-13)
    g["prec"], c["prec"] = a, b
    a, b = both(lambda m, o, nw: m.alpha()[:n])
    g["alpha"], c["alpha"] = a, b
    d_alpha = np.abs(a - b).max()
    _stat(label, what="alpha", abs=d_alpha, rel=(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)).max())
    if d_alpha > 1e-12:
        # alpha_i = |N| / D with N a commutator: in nearly uniform flow (f_j - f_i small against f) the
        # quotient is ill conditioned IN THE REFERENCE'S FORMULA (indicator.h:230-257). The yardstick is then
        # the oracle's own sensitivity to a last-bit perturbation of its input.
        sens = alpha_last_bit_sensitivity(oracle, off, params, U_before, dirichlet, tau, b)
        _stat(label, what="alpha_sensitivity", sens=float(sens.max()))
        beyond = np.abs(a - b) > 1e-12
        _check(params is not None and (np.abs(a - b)[beyond] <= 4.0 * sens[beyond]).all(), label, 'alpha',
               (d_alpha, float(sens.max()), int(beyond.sum())))
    a, b = both(lambda m, o, nw: m.debug_fetch("dij"))
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-300)
    if keep_matrices:
        g["dij"], c["dij"] = a, b
    assert abs(tau_g - tau_c) <= 1e-12 * tau_c
    a, b = both(lambda m, o, nw: m.debug_fetch("bounds"))
    # (relative 1e-12; the absolute floor, 1e-20 of the largest bound, only lets underflow-sized entries through:
    # kinetic-energy bounds of 1e-37 next to a dry shallow-water node differ in their last bits between operation orders)
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-20 * np.abs(b).max())
    g["bounds"], c["bounds"] = a, b
    a, b = both(lambda m, o, nw: m.debug_fetch("r"))
    k = mg.k
    r_scale = np.abs(b.reshape(-1, k)).max(axis=0)
    r_err = (np.abs(a - b).reshape(-1, k) / np.maximum(r_scale, 1e-300)).max()
    _stat(label, what="r", rel_to_max=r_err)
    _check(r_err <= 1e-12, label, 'r', r_err)
    g["r"], c["r"] = a, b

    U_g, U_c = both(lambda m, o, nw: nw.download()[:n])
    g["U"], c["U"] = U_g, U_c
    scale = np.maximum(np.abs(U_c).max(axis=0), 1e-3 * np.abs(U_c).max())

    lg, lc = both(lambda m, o, nw: m.debug_fetch("lij"))
    c["lij"] = lc
    dl_first = np.abs(lg - lc)
    if keep_matrices:
        g["lij"] = lg
    del lg
    ln_g, ln_c = both(lambda m, o, nw: m.debug_fetch("lij_next"))
    c["lij_next"] = ln_c
    dl_next = np.abs(ln_g - ln_c)
    if keep_matrices:
        g["lij_next"] = ln_g
    del ln_g

    if fetch_pij:
        pg, pc = both(lambda m, o, nw: m.debug_fetch("pij"))
        # The diagonal entry P_ii is written by step 4 and never read again -- steps 5, 6, 7 loop over
        # col_idx >= 1 (hyperbolic_module.template.h:941,964,1107,1137). With stages == 0 it is identically 0
        # (-flux_ii + 1 * flux_ii), and the HIP path, which recomputes P_ij in step 5 instead of storing it in
        # step 4, does not write it at all (it keeps whatever an earlier multi-stage step left there): excluded.
        diag = off.row_starts[:n].astype(np.int64)
        pg.reshape(-1, k)[diag] = pc.reshape(-1, k)[diag]
        p_scale = np.abs(pc.reshape(-1, k)).max(axis=0)
        p_err = (np.abs(pg - pc).reshape(-1, k) / np.maximum(p_scale, 1e-300)).max()
        _stat(label, what="pij", rel_to_max=p_err)
        if p_err > 1e-12:   # say where
            e_bad = int((np.abs(pg - pc).reshape(-1, k) / np.maximum(p_scale, 1e-300)).max(axis=1).argmax())
            rs_ = off.row_starts[: n + 1].astype(np.int64)
            i_bad = int(np.searchsorted(rs_, e_bad, side="right") - 1)
            _check(False, label, 'pij', (p_err, "entry", e_bad, "row", i_bad, "col_idx", e_bad - int(rs_[i_bad]),
                                         pg.reshape(-1, k)[e_bad].tolist(), pc.reshape(-1, k)[e_bad].tolist()))
        c["pij"] = pc
        if keep_matrices:
            g["pij"] = pg
        del pg

    # ---- l_ij: 1e-10 absolute; anything beyond must sit on the psi_r = 0 discontinuity
    flipped_rows = set()
    n_flips = {}
    for name, dl in (("lij", dl_first), ("lij_next", dl_next)):
        # no systematic difference: the typical |dl| is round-off. (Where P_ij is negligible -- quiescent water,
        # uniform flow -- l_ij is a quotient of two round-off sized numbers in the reference itself and says
        # nothing; the median is taken over the pairs with a non-negligible P_ij, as the 1e-10 bound below.)
        if not (np.median(dl) == 0.0 or np.median(dl) < 1e-14):
            if "pij" not in c:
                c["pij"] = mc.debug_fetch("pij")
            relevant = (np.abs(c["pij"].reshape(-1, k)) / scale).max(axis=1) > 1e-3
            med = np.median(dl[relevant]) if relevant.any() else 0.0
            _stat(label, what=name + "_median", all=float(np.median(dl)), relevant=float(med))
            assert med < 1e-14, (name, float(med))
        idx = np.nonzero(dl > L_TOL)[0]
        if idx.size and "pij" not in c:   # full-size run: P_ij of the oracle only, and only now
            c["pij"] = mc.debug_fetch("pij")
        if idx.size:
            # Where |P_ij| < 1e-3 max|U| the quotient the limiter forms, (rho_max - rho_U) / |rho_P| and its
            # relatives, is round-off dominated in the reference itself: l_ij may differ there, but then the pair
            # -- limit() places U + t P on the boundary of the invariant set to its Newton tolerance, which fixes
            # t |P_ij| (in units of the state), not t, once |P_ij| is small: the 1e-10 contract is applied to the
            # LIMITED UPDATE there, |dl| |P_ij| / |U| <= 1e-10 (what the pair can still move in U_new is bounded
            # row by row further down). Anything else goes through the branch-flip classification.
            rs_ = off.row_starts[: n + 1].astype(np.int64)
            rows_of = np.searchsorted(rs_, idx, side="right") - 1
            lam_of = 1.0 / np.maximum(rs_[rows_of + 1] - rs_[rows_of] - 1, 1)
            p_rel = (np.abs(c["pij"].reshape(-1, k)[idx]) / scale).max(axis=1)
            small = p_rel <= 1e-3
            effect = dl[idx] * lam_of * p_rel
            _stat(label, what=name + "_small_P", n=int(small.sum()),
                  max_dl_p=float((dl[idx] * p_rel)[small].max()) if small.any() else 0.0,
                  max_effect=float(effect[small].max()) if small.any() else 0.0)
            idx = idx[~(small & (dl[idx] * p_rel <= L_TOL))]   # everything else: a classified branch flip
        n_flips[name] = int(idx.size)
        _stat(label, what=name, n_outliers=int(idx.size), n=int(dl.size), max=float(dl.max()))
        if idx.size == 0:
            continue
        if not (equation == capi.EQ_EULER and oracle is not None and params is not None):
            _check(False, label, name + " differs and cannot be classified", (int(idx.size), float(dl[idx].max())))
            continue
        if "flip" not in g:
            g["flip"] = EulerFlipClassifier(oracle, off, params, U_before, dirichlet, tau_c, c, stage_U,
                                            stage_weights)
        # isolated pairs, not a systematic difference: at most 64 per sweep whatever the mesh size, and EVERY one of
        # them is classified
        _check(idx.size <= 64, label, name + " too many outliers", int(idx.size))
        for e in idx[:64]:
            rel, i, j = g["flip"].psi_r(name, int(e))
            _stat(label, what=name + "_flip", entry=int(e), dl=float(dl[e]), psi_rel=float(rel))
            if rel > PSI_ROUND_OFF and params.limiter_iterations == 2 and not stage_U:
                # not a round-off sized psi_r of the oracle's inputs. Second pass: psi_r is sized by the first pass's
                # Newton tolerance; either pass: psi may vanish to round-off along the whole segment (the limiter's
                # Newton steps then divide round-off by round-off, in the reference as here). Decided on the DEVICE's
                # inputs, with the oracle's and the device's own limiter (follows_its_inputs).
                for key in ("lij", "lij_next", "pij"):
                    if key not in g:
                        g[key] = mg.debug_fetch(key)
                ok, detail = g["flip"].follows_its_inputs(int(e), g, second_pass=(name == "lij"))
                _stat(label, what=name + "_follows_inputs", entry=int(e), ok=bool(ok), detail=repr(detail))
                _check(ok, label, name + " differs beyond what its inputs and the limiter's own relaxation explain",
                       (int(e), float(dl[e]), float(rel), detail))
                flipped_rows.update((i, j))
                continue
            _check(rel <= PSI_ROUND_OFF, label, name + " outlier off the psi_r = 0 branch",
                   (int(e), float(dl[e]), float(rel)))
            flipped_rows.update((i, j))

    err = np.abs(U_g - U_c) / scale
    _stat(label, what="U", max=float(err.max()), n_over=int((err > U_TOL).sum()))
    over = np.nonzero((err > U_TOL).any(axis=1))[0]
    if over.size:
        # U_new = U_low + sum_j min(l_ij, l_ji) lambda P_ij (two passes): an l_ij that is only defined up to the
        # limiter's Newton tolerance (1e-10), or sits on a classified branch flip, moves U_new by
        # |dl| lambda |P_ij|. Every entry beyond 1e-11 must be covered by exactly that propagated difference.
        if "pij" not in c:
            c["pij"] = mc.debug_fetch("pij")
        rs = off.row_starts[: n + 1].astype(np.int64)
        cols = off.columns[: rs[-1]].astype(np.int64)
        P = np.abs(c["pij"].reshape(-1, k))
        for i in over:
            sl = slice(rs[i], rs[i + 1])
            dl = dl_first[sl] + dl_next[sl]
            for e in range(rs[i] + 1, rs[i + 1]):          # the transposed entries l_ji enter through min()
                j = cols[e]
                if j < n:
                    hit = np.nonzero(cols[rs[j]:rs[j + 1]] == i)[0]
                    if hit.size:
                        dl[e - rs[i]] += dl_first[rs[j] + hit[0]] + dl_next[rs[j] + hit[0]]
            lam = 1.0 / max(1, rs[i + 1] - rs[i] - 1)
            bound = U_TOL + lam * (dl[:, None] * P[sl]).sum(axis=0) / scale
            _check((err[i] <= bound).all(), label, "U_new beyond 1e-11 + the propagated l_ij differences",
                   (int(i), err[i].tolist(), bound.tolist()))
    g["n_flips"] = n_flips
    return g, c

"""Partitioned runs, rank by rank, for BOTH backends: n contexts (one host thread each) that own the parts of a
mesh -- HIP contexts with the in-process transport (the event graph of the RCCL leg), oracle contexts with ghost
vectors / ghost rows / reductions exchanged through shared numpy buffers at the oracle's synchronisation points.
Every rank returns whatever `body(module, part, rank)` returns, so that the two partitioned runs can be compared
PER RANK, intermediates and ghost ranges included (the reference pins its exchange
with real 4-rank runs, tests/common/sparsity_pattern_simd_01.cc, tests/euler/check-mass-conservation_02)."""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np

from ryujin_amd import HyperbolicModule, capi


def exchange_lists(part):
    """The exchange pattern of one rank's offline data (ryujin_hip_offline, include/ryujin_hip.h) as numpy
    arrays; works for the generator's parts and for the views of helpers_unstructured.partition alike."""
    o = part.c.contents
    n = o.n_nbr
    ptr = np.ctypeslib.as_array(o.row_starts, shape=(o.n_relevant + 1,)).astype(np.int64)
    send_off = [o.send_off[q] for q in range(n + 1)] if n else [0]
    recv_off = [o.recv_off[q] for q in range(n + 1)] if n else [o.n_owned]
    row_send_off = [o.row_send_off[q] for q in range(n + 1)] if n else [0]
    send_idx = np.array([o.send_idx[q] for q in range(send_off[-1])], dtype=np.int64)
    row_pos = np.array([ptr[o.row_send_row[q]] + o.row_send_col[q] for q in range(row_send_off[-1])],
                       dtype=np.int64)
    return dict(nbr=[o.nbr_rank[q] for q in range(n)], send_off=send_off, recv_off=recv_off, send_idx=send_idx,
                row_send_off=row_send_off, row_pos=row_pos, ptr=ptr, n_relevant=o.n_relevant)


def _join(threads, out, n_ranks):
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive(), "rank thread hung"
    for r in range(n_ranks):
        if isinstance(out[r], BaseException):
            raise out[r]
    return [out[r] for r in range(n_ranks)]


def run_hip_ranks(parts, make_params, body, device=0):
    """body(module, part, rank) on n HIP contexts of one GPU, in-process transport."""
    lib = capi.load_hip()
    n_ranks = len(parts)
    comms = (C.c_void_p * n_ranks)()
    assert lib.ryujin_hip_comm_init_local(comms, n_ranks, device) == 0
    out = {}

    def worker(r):
        try:
            m = HyperbolicModule(parts[r], make_params(), backend="hip", comm=C.c_void_p(comms[r]), device=device)
            out[r] = body(m, parts[r], r)
            m.close()
        except BaseException as e:  # noqa: BLE001 -- surfaced in the main thread
            out[r] = e

    try:
        return _join([threading.Thread(target=worker, args=(r,)) for r in range(n_ranks)], out, n_ranks)
    finally:
        for r in range(n_ranks):
            lib.ryujin_hip_comm_destroy(C.c_void_p(comms[r]))


def run_oracle_ranks(oracle, parts, make_params, body):
    """body(module, part, rank) on n oracle contexts; the exchanges of the reference (SURVEY.md section 2.2:
    ghost vectors, matrix ghost rows, MPI::min / logical_or) go through shared buffers."""
    n_ranks = len(parts)
    barrier = threading.Barrier(n_ranks)
    mail, scratch, out = {}, [0.0] * n_ranks, {}
    lib = oracle.load()

    def worker(r):
        try:
            x = exchange_lists(parts[r])
            nbr, ptr = x["nbr"], x["ptr"]
            send_off, recv_off, send_idx = x["send_off"], x["recv_off"], x["send_idx"]
            row_send_off, row_pos = x["row_send_off"], x["row_pos"]

            def exchange(user, what, data, n_comp):
                if what in (10, 11):
                    scratch[r] = data[0]
                    barrier.wait()
                    val = min(scratch) if what == 10 else max(scratch)
                    barrier.wait()
                    data[0] = val
                    return
                if what < 4 or what == 6:
                    a = np.ctypeslib.as_array(data, shape=(x["n_relevant"] * n_comp,)).reshape(-1, n_comp)
                    for q, p in enumerate(nbr):
                        mail[(r, p)] = a[send_idx[send_off[q]:send_off[q + 1]]].copy()
                    barrier.wait()
                    for q, p in enumerate(nbr):
                        a[recv_off[q]:recv_off[q + 1]] = mail[(p, r)]
                else:
                    a = np.ctypeslib.as_array(data, shape=(int(ptr[-1]),))
                    for q, p in enumerate(nbr):
                        mail[(r, p)] = a[row_pos[row_send_off[q]:row_send_off[q + 1]]].copy()
                    barrier.wait()
                    for q, p in enumerate(nbr):
                        a[int(ptr[recv_off[q]]):int(ptr[recv_off[q + 1]])] = mail[(p, r)]
                barrier.wait()

            cb = oracle.EXCHANGE_FN(exchange)
            m = HyperbolicModule(parts[r], make_params(), backend=oracle.backend())
            lib.ryujin_oracle_set_exchange(m._ctx, cb, None)
            out[r] = body(m, parts[r], r)
            m.close()
        except BaseException as e:  # noqa: BLE001
            out[r] = e
            barrier.abort()

    return _join([threading.Thread(target=worker, args=(r,)) for r in range(n_ranks)], out, n_ranks)


def one_update_with_intermediates(U_local, dirichlet_of=None, tau=0.0):
    """body: upload the rank's local state, run ONE update, return every array the rank holds afterwards --
    over the whole locally relevant range wherever the module keeps a ghost range / ghost rows."""

    def body(m, part, r):
        old, new = m.new_state_vector(U_local[r]), m.new_state_vector()
        m.prepare_state_vector(old, 0.0, dirichlet_of(part) if dirichlet_of else None)
        used = m.step(old, [], [], new, tau)
        n = part.n_owned
        return dict(tau=used, status=m.last_status,
                    U_old=old.download(),                     # boundary conditions + ghost U
                    prec=old.download_precomputed(),          # ghost range: exchanged (oracle) / local (HIP)
                    alpha=m.alpha(),                          # ghost range: exchanged
                    dij=m.debug_fetch("dij"),                 # owned rows INCLUDING their ghost columns
                    bounds=m.debug_fetch("bounds"),
                    r=m.debug_fetch("r_all"),                 # ghost range: exchanged
                    pij=m.debug_fetch("pij"),
                    lij=m.debug_fetch("lij_all"),             # ghost ROWS: exchanged
                    lij_next=m.debug_fetch("lij_next_all"),
                    U=new.download()[:n])

    return body


def compare_rank(part, g, c, k, label="", scales=None):
    """One rank of the HIP run (g) against the same rank of the oracle run (c), the contract of
    helpers_parity.py applied to the rank's WHOLE locally relevant range. scales = (r_scale, p_scale): the
    largest entry per component of r_i / P_ij over the WHOLE mesh (all ranks), the yardstick of the single-rank
    contract -- a rank whose slab holds nearly uniform flow has no meaningful maximum of its own."""
    n, nr = part.n_owned, part.n_relevant
    assert g["status"] == c["status"], label
    assert abs(g["tau"] - c["tau"]) <= 1e-12 * c["tau"], (label, g["tau"], c["tau"])
    np.testing.assert_allclose(g["U_old"], c["U_old"], rtol=1e-14, atol=1e-14, err_msg=label + " U_old incl. ghosts")
    np.testing.assert_allclose(g["prec"], c["prec"], rtol=1e-13, err_msg=label + " precomputed incl. ghosts")
    assert np.abs(g["alpha"] - c["alpha"]).max() <= 1e-12, (label, "alpha incl. ghosts")
    if nr > n:  # the comparison above is not vacuous on the ghost range
        assert np.abs(c["alpha"][n:]).max() > 0.0 or np.abs(c["r"].reshape(nr, k)[n:]).max() > 0.0, label
    np.testing.assert_allclose(g["dij"], c["dij"], rtol=1e-12, atol=1e-300, err_msg=label + " d_ij incl. ghost columns")
    np.testing.assert_allclose(g["bounds"], c["bounds"], rtol=1e-12, err_msg=label + " bounds")
    r_scale = scales[0] if scales else np.maximum(np.abs(c["r"].reshape(nr, k)).max(axis=0), 1e-300)
    assert (np.abs(g["r"] - c["r"]).reshape(nr, k) / r_scale).max() <= 1e-12, (label, "r incl. ghosts")
    diag = np.ctypeslib.as_array(part.c.contents.row_starts, shape=(nr + 1,)).astype(np.int64)[:n]
    g["pij"].reshape(-1, k)[diag] = c["pij"].reshape(-1, k)[diag]   # P_ii is never read (helpers_parity.py)
    p_scale = scales[1] if scales else np.maximum(np.abs(c["pij"].reshape(-1, k)).max(axis=0), 1e-300)
    p_err = (np.abs(g["pij"] - c["pij"]).reshape(-1, k) / p_scale).max(axis=1)
    if p_err.max() > 1e-12:   # say where: row, column, owned / export / ghost
        ptr_ = np.ctypeslib.as_array(part.c.contents.row_starts, shape=(nr + 1,)).astype(np.int64)
        cols_ = np.ctypeslib.as_array(part.c.contents.columns, shape=(int(ptr_[nr]),))
        bad = np.nonzero(p_err > 1e-12)[0]
        rows_ = np.searchsorted(ptr_, bad, side="right") - 1
        detail = [(int(e), int(i), int(cols_[e]), float(p_err[e]), g["pij"].reshape(-1, k)[e].tolist(),
                   c["pij"].reshape(-1, k)[e].tolist()) for e, i in list(zip(bad, rows_))[:6]]
        raise AssertionError((label, "P_ij", int(bad.size), float(p_err.max()), "n_export", part.n_export, "n_owned", n,
                              "ghost columns among the bad entries", int((cols_[bad] >= n).sum()), detail))
    scale = np.maximum(np.abs(c["U"]).max(axis=0), 1e-3 * np.abs(c["U"]).max())
    # l_ij, l'_ij of the OWNED rows: 1e-10 absolute (the limiter's Newton tolerance). Where P_ij is negligible the
    # quotient the limiter forms is round-off dominated in the reference itself; such a pair may differ by more,
    # but then its EFFECT on the update, |dl| lambda |P_ij|, must stay below 1e-11 of the solution scale.
    ptr = np.ctypeslib.as_array(part.c.contents.row_starts, shape=(nr + 1,)).astype(np.int64)
    nnz_owned = int(ptr[n])
    lam = np.repeat(1.0 / np.maximum(np.diff(ptr[: n + 1]) - 1, 1), np.diff(ptr[: n + 1]))
    effect_scale = (np.abs(c["pij"].reshape(-1, k)) / scale).max(axis=1) * lam
    accepted = {}
    for name in ("lij", "lij_next"):
        d = np.abs(g[name][:nnz_owned] - c[name][:nnz_owned])
        out = np.nonzero(d > 1e-10)[0]
        assert (d[out] * effect_scale[out] <= 1e-11).all(), \
            (label, name, "differs beyond 1e-10 where P_ij matters", float((d[out] * effect_scale[out]).max()))
        accepted[name] = set(out.tolist())
    assert (np.abs(g["U"] - c["U"]) / scale).max() <= 1e-11, (label, "U_new")
    return accepted


def global_scales(parts, ref, k):
    """(max |r_i|, max |P_ij|) per component over the owned rows of all ranks"""
    r = np.max([np.abs(ref[q]["r"].reshape(-1, k)[: p.n_owned]).max(axis=0) for q, p in enumerate(parts)], axis=0)
    P = np.max([np.abs(ref[q]["pij"].reshape(-1, k)).max(axis=0) for q in range(len(parts))], axis=0)
    return np.maximum(r, 1e-300), np.maximum(P, 1e-300)


def compare_ghost_rows(parts, hip, ref, accepted):
    """The ghost rows of l_ij / l'_ij every rank RECEIVED: (i) in the HIP run they are bitwise the entries the
    neighbour holds at its send positions (row_send_row / row_send_col) -- the transport moved the right entries
    to the right places; (ii) against the oracle rank's ghost rows they agree to 1e-10, except where they are
    copies of an owned entry of the neighbour that compare_rank accepted by its negligible effect."""
    lists = [exchange_lists(p) for p in parts]
    n_checked = 0
    for r, x in enumerate(lists):
        for q, p in enumerate(x["nbr"]):
            g0, g1 = int(x["ptr"][x["recv_off"][q]]), int(x["ptr"][x["recv_off"][q + 1]])
            y = lists[p]
            qq = y["nbr"].index(r)
            src = y["row_pos"][y["row_send_off"][qq]:y["row_send_off"][qq + 1]]
            assert src.size == g1 - g0, (r, p)
            for name in ("lij", "lij_next"):
                assert np.array_equal(hip[r][name][g0:g1], hip[p][name][src]), (r, p, name, "ghost rows are copies")
                d = np.abs(hip[r][name][g0:g1] - ref[r][name][g0:g1])
                for e in np.nonzero(d > 1e-10)[0]:
                    assert int(src[e]) in accepted[p][name], (r, p, name, int(e), float(d[e]))
            n_checked += g1 - g0
    return n_checked


def compare_rank_files(oracle, parts, prefix, make_params, dirichlet_of, k):
    """Parent side of the multi-process runs (tests/rccl_worker.py `intermediates`): every rank stored
    <prefix>.rank<r>.npz -- its local state before and its arrays after one update; the partitioned ORACLE is run
    from the same states and compared rank by rank, ghost rows included."""
    hip = [dict(np.load(f"{prefix}.rank{r}.npz")) for r in range(len(parts))]
    for h in hip:
        h["tau"], h["status"] = float(h["tau"]), int(h["status"])
    ref = run_oracle_ranks(oracle, parts, make_params,
                           one_update_with_intermediates([h["U_local"] for h in hip], dirichlet_of))
    scales = global_scales(parts, ref, k)
    accepted = [compare_rank(part, hip[r], ref[r], k, label=f"rank {r}", scales=scales) for r, part in enumerate(parts)]
    n = compare_ghost_rows(parts, hip, ref, accepted)
    assert n > 0
    return hip, ref

"""Test helper: OfflineData of a genuinely unstructured mesh -- continuous P1 elements on a Delaunay
triangulation of a disk (or P1 tetrahedra in a ball) -- assembled with numpy exactly as the reference assembles its matrices
(source/offline_data.template.h:566-576 `c_ij = int phi_i grad phi_j`, `m_ij = int phi_i phi_j`;
:790-802 lumped mass; :1246-1361 boundary normals = normalised sum of the face integrals of phi_i n;
:1369-1463 coupling boundary pairs). Nothing on the hot path is specific to Q1 quadrilaterals: it sees
row lengths between 4 and ~10, c_ij that are antisymmetric only in the interior, boundary normals in every
direction and varying m_i."""
import numpy as np
from scipy.spatial import Delaunay

from helpers_layout import OfflineView
from ryujin_amd import capi


def disk_points(n_rings, seed=7, jitter=0.25):
    """Concentric rings of points (radius k / n_rings, 6 k points each), jittered inside, exact on the
    boundary ring, so that the hull of the triangulation is a regular polygon inscribed in the unit circle."""
    rng = np.random.default_rng(seed)
    pts = [np.zeros((1, 2))]
    for k in range(1, n_rings + 1):
        m = 6 * k
        phi = 2.0 * np.pi * (np.arange(m) + 0.5 * (k % 2)) / m
        r = np.full(m, k / n_rings)
        if k < n_rings:
            r = r + jitter / n_rings * rng.uniform(-1.0, 1.0, m)
            phi = phi + jitter * 2.0 * np.pi / m * rng.uniform(-1.0, 1.0, m)
        pts.append(np.column_stack([r * np.cos(phi), r * np.sin(phi)]))
    return np.concatenate(pts)


def ball_points(n_interior, n_surface, seed=11):
    """Points of a unit ball: a Fibonacci lattice on the sphere (the hull of the triangulation) and
    uniformly distributed random points inside radius 0.93."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_surface) + 0.5
    phi = np.arccos(1.0 - 2.0 * k / n_surface)
    theta = np.pi * (1.0 + 5.0 ** 0.5) * k
    surf = np.column_stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)])
    inner = []
    while len(inner) < n_interior:
        q = rng.uniform(-1.0, 1.0, 3)
        if np.linalg.norm(q) < 0.93:
            inner.append(q)
    return np.concatenate([np.zeros((1, 3)), np.array(inner), surf])


def p1_offline(points, boundary_id=capi.BC_SLIP):
    """Assemble the OfflineData arrays of continuous P1 elements on the Delaunay triangulation (2-D) or
    tetrahedralisation (3-D) of `points`. Returns (OfflineView, dict with simplices, measure, rows, is_bdry)."""
    tri = Delaunay(points)
    T = tri.simplices
    n, dim = points.shape
    x = points
    fact = 2.0 if dim == 2 else 6.0
    c_acc, m_acc = {}, {}
    measure = 0.0
    for t in T:
        p = x[t]
        M = (p[1:] - p[0]).T                      # columns: edge vectors from vertex 0
        V = abs(np.linalg.det(M)) / fact
        if V < 1e-14:                             # degenerate sliver of (nearly) cospherical hull points
            continue
        measure += V
        Minv = np.linalg.inv(M)                   # rows: grad phi_1 .. grad phi_dim
        grads = np.vstack([-Minv.sum(axis=0), Minv])
        for a in range(dim + 1):
            for b in range(dim + 1):
                key = (int(t[a]), int(t[b]))
                c_acc[key] = c_acc.get(key, 0.0) + V / (dim + 1.0) * grads[b]
                m_acc[key] = m_acc.get(key, 0.0) + V / ((dim + 1.0) * (dim + 2.0)) * (2.0 if a == b else 1.0)
    # boundary facets: those of the convex hull; int_F phi_i n dS = |F| n / dim for each of its vertices
    centre = x.mean(axis=0)
    nrm = np.zeros((n, dim))
    is_bdry = np.zeros(n, dtype=bool)
    for f in tri.convex_hull:
        q = x[f]
        if dim == 2:
            t = q[1] - q[0]
            nu = np.array([t[1], -t[0]])          # |e| n
        else:
            nu = 0.5 * np.cross(q[1] - q[0], q[2] - q[0])   # |F| n
        if np.dot(nu, q.mean(axis=0) - centre) < 0:
            nu = -nu
        for v in f:
            nrm[v] += nu / dim
            is_bdry[v] = True
    rows = [[i] for i in range(n)]
    for (i, j) in c_acc:
        if i != j:
            rows[i].append(j)
    rows = [[r[0]] + sorted(r[1:]) for r in rows]
    row_starts = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
    columns = np.concatenate([np.array(r, dtype=np.uint32) for r in rows])
    cij = np.array([c_acc[(i, j)] for i, r in enumerate(rows) for j in r])
    mij = np.array([m_acc[(i, j)] for i, r in enumerate(rows) for j in r])
    mi = np.add.reduceat(mij, row_starts[:-1].astype(np.int64))
    b_i = np.flatnonzero(is_bdry).astype(np.uint32)
    b_normal = nrm[b_i] / np.linalg.norm(nrm[b_i], axis=1)[:, None]
    p_i, p_col, p_j = [], [], []
    for i in b_i:
        for col_idx, j in enumerate(rows[i]):
            if col_idx > 0 and is_bdry[j]:
                p_i.append(i), p_col.append(col_idx), p_j.append(j)
    off = OfflineView(dim, 0, 0, n, n, 1, row_starts, columns, cij, mij, mi, 1.0 / mi, mi.sum(), b_i, b_normal,
                      np.full(len(b_i), boundary_id, dtype=np.uint8), p_i, p_col, p_j)
    off.positions = x
    off.row_starts, off.columns, off.cij_csr, off.mij_csr, off.mi = row_starts, columns, cij, mij, mi
    return off, dict(triangles=T, area=measure, rows=rows, is_bdry=is_bdry)


def partition(off, info, owner, bathymetry=None):
    """Split a single-rank OfflineView into per-rank OfflineViews for an ARBITRARY ownership map
    (owner[g] = rank of global node g), in the reference's local numbering and exchange semantics
    (include/ryujin_hip.h; source/offline_data.template.h:210-272, sparse_matrix_simd.template.h:61-74,
    :247-259): owned = [exported | rest], ghosts sorted by (owner, owner's local index); ghost rows keep the
    diagonal and the columns owned by this rank; a rank sends the owned entries of its exported rows whose
    column belongs to the receiver. Ranks may have any number of neighbours and a node may be exported
    to several of them -- topologies the slab partition of the mesh generator never produces."""
    rows = info["rows"]
    n = len(rows)
    n_ranks = int(owner.max()) + 1
    rs = off.row_starts.astype(np.int64)
    entry = {(i, int(off.columns[e])): e for i in range(n) for e in range(rs[i], rs[i + 1])}
    is_bdry = info["is_bdry"]
    g_normal = np.zeros((n, off.dim))
    g_normal[off._keep["b_i"]] = off._keep["b_normal"]
    g_id = np.zeros(n, dtype=np.uint8)
    g_id[off._keep["b_i"]] = off._keep["b_id"]

    owned, loc = [], []
    for r in range(n_ranks):
        mine = [g for g in range(n) if owner[g] == r]
        exported = [g for g in mine if any(owner[j] != r for j in rows[g][1:])]
        rest = [g for g in mine if g not in set(exported)]
        order = exported + rest
        owned.append((order, len(exported)))
        loc.append({g: k for k, g in enumerate(order)})

    views = []
    for r in range(n_ranks):
        order, n_export = owned[r]
        n_owned = len(order)
        ghosts = sorted({j for g in order for j in rows[g][1:] if owner[j] != r},
                        key=lambda j: (owner[j], loc[owner[j]][j]))
        l2g = order + ghosts
        lidx = dict(loc[r])
        lidx.update({g: n_owned + k for k, g in enumerate(ghosts)})
        nbrs = sorted({int(owner[j]) for j in ghosts})
        recv_off = [n_owned]
        for p in nbrs:
            recv_off.append(recv_off[-1] + sum(1 for j in ghosts if owner[j] == p))
        lrows, lc, lm, l_inc, l_minv = [], [], [], [], []
        dg = getattr(off, "_dg", None)       # discontinuous ansatz: (incidence, mass_matrix_inverse)
        for g in l2g:
            keep = rows[g][1:] if owner[g] == r else [j for j in rows[g][1:] if owner[j] == r]
            cols = sorted(keep, key=lambda j: lidx[j])
            lrows.append([lidx[g]] + [lidx[j] for j in cols])
            for j in [g] + cols:
                lc.append(off.cij_csr[entry[(g, j)]])
                lm.append(off.mij_csr[entry[(g, j)]])
                if dg is not None:
                    l_inc.append(dg[0][entry[(g, j)]])
                    l_minv.append(dg[1][entry[(g, j)]])
        row_starts = np.cumsum([0] + [len(x) for x in lrows]).astype(np.uint64)
        columns = np.concatenate([np.array(x, dtype=np.uint32) for x in lrows])
        send_off, send_idx, row_send_off, row_send_row, row_send_col = [0], [], [0], [], []
        for p in nbrs:
            exp_p = sorted(lidx[g] for g in order if any(owner[j] == p for j in rows[g][1:]))
            send_idx.extend(exp_p)
            send_off.append(len(send_idx))
            ghosts_of_p = [lidx[j] for j in ghosts if owner[j] == p]
            ghost_range = (min(ghosts_of_p), max(ghosts_of_p) + 1) if ghosts_of_p else None
            for i, c in ghost_row_send_entries(lrows, exp_p, ghost_range):
                row_send_row.append(i)
                row_send_col.append(c)
            row_send_off.append(len(row_send_row))
        b_g = [g for g in order if is_bdry[g]]
        b_i = np.array([lidx[g] for g in b_g], dtype=np.uint32)
        p_i, p_col, p_j = [], [], []
        for g in b_g:
            i = lidx[g]
            for c in range(1, len(lrows[i])):
                if is_bdry[l2g[lrows[i][c]]]:
                    p_i.append(i), p_col.append(c), p_j.append(lrows[i][c])
        mi = off.mi[l2g]
        v = OfflineView(off.dim, n_export, n_owned, n_owned, len(l2g), 1, row_starts, columns, np.array(lc), np.array(lm),
                        mi, 1.0 / mi, off.measure_of_omega, b_i, g_normal[b_g].reshape(-1, off.dim), g_id[b_g],
                        p_i, p_col, p_j)
        k = v._keep
        k["nbr_rank"] = np.array(nbrs, dtype=np.int32)
        for name, val in (("send_off", send_off), ("send_idx", send_idx), ("recv_off", recv_off),
                          ("row_send_off", row_send_off), ("row_send_row", row_send_row),
                          ("row_send_col", row_send_col)):
            k[name] = np.array(val, dtype=np.uint32)
        o = v._o
        o.n_nbr = len(nbrs)
        o.nbr_rank = capi.as_ptr(k["nbr_rank"], capi.c_int_p)
        for name in ("send_off", "send_idx", "recv_off", "row_send_off", "row_send_row", "row_send_col"):
            setattr(o, name, capi.as_ptr(k[name], capi.c_u32_p))
        if dg is not None:
            from helpers_dg import attach_dg
            attach_dg(v, np.array(l_inc), np.array(l_minv))
        v.positions = off.positions[l2g]
        v.global_ids = np.array(l2g, dtype=np.int64)
        v.b_positions = off.positions[b_g].reshape(-1, off.dim)
        if bathymetry is not None:
            v.set_initial_precomputed(bathymetry[l2g])
        views.append(v)
    return views


def ghost_row_send_entries(local_rows, exported_rows, ghost_range):
    """The (row, col_idx) entries of a matrix that ONE neighbour rank receives into its ghost rows
    (ryujin_hip_offline::row_send_row / row_send_col, include/ryujin_hip.h; the reference's
    SparsityPatternSIMD::entries_to_be_sent, source/sparse_matrix_simd.template.h:196-264): for every row
    exported to that rank, in export order, the diagonal and then the entries whose column lies in the ghost
    range RECEIVED from the same rank -- a ghost row only holds the transposes of owned entries (:61-74).
    ghost_range = (begin, end) in local indices, or None if that rank sends us nothing: then nothing is sent
    to it either (:229-247, the unmatched import target).
    This is a binding of THE C function that states the rule (include/ryujin_exchange_lists.h, exported by
    libryujin_synth.so): the same code builds the lists of the mesh generator (bench.py --gpus N, rccl_worker.py),
    of every partitioned test and of the deal.II-side adapter. Pinned against the reference's
    tests/common/sparsity_pattern_simd_01.mpirun=4.output in tests/test_send_lists_golden.py."""
    if ghost_range is None:
        return []
    lib = capi.load_synth()
    ptr = np.zeros(len(local_rows) + 1, dtype=np.uint64)
    ptr[1:] = np.cumsum([len(r) for r in local_rows])
    cols = np.array([c for r in local_rows for c in r], dtype=np.uint32)
    rows = np.ascontiguousarray(exported_rows, dtype=np.uint32)
    args = (capi.as_ptr(ptr, capi.c_u64_p), capi.as_ptr(cols, capi.c_u32_p), capi.as_ptr(rows, capi.c_u32_p),
            rows.size, int(ghost_range[0]), int(ghost_range[1]))
    n = lib.ryujin_synth_ghost_row_send_entries(*args, None, None)
    out_row, out_col = np.zeros(max(n, 1), dtype=np.uint32), np.zeros(max(n, 1), dtype=np.uint32)
    lib.ryujin_synth_ghost_row_send_entries(*args, capi.as_ptr(out_row, capi.c_u32_p),
                                            capi.as_ptr(out_col, capi.c_u32_p))
    return [(int(out_row[q]), int(out_col[q])) for q in range(n)]


def run_partitioned_oracle(oracle, views, params, U0_global, n_updates, dirichlet_fn=None):
    """One oracle context per rank, one host thread each; ghost vectors, matrix ghost rows and the min/or
    reductions are exchanged through shared numpy buffers at the oracle's synchronisation points. Returns
    (U in global numbering, list of tau per update)."""
    import threading

    from ryujin_amd import HyperbolicModule
    n_ranks = len(views)
    barrier = threading.Barrier(n_ranks)
    mail, scratch, out = {}, [0.0] * n_ranks, {}
    lib = oracle.load()

    def worker(r):
        try:
            v = views[r]
            k = v._keep
            nbr, ptr = k["nbr_rank"].tolist(), k["row_starts"].astype(np.int64)
            send_off, recv_off = k["send_off"].tolist(), k["recv_off"].tolist()
            send_idx = k["send_idx"].astype(np.int64)
            row_send_off = k["row_send_off"].tolist()
            row_pos = ptr[k["row_send_row"].astype(np.int64)] + k["row_send_col"].astype(np.int64)

            def exchange(user, what, data, n_comp):
                if what in (10, 11):
                    scratch[r] = data[0]
                    barrier.wait()
                    val = min(scratch) if what == 10 else max(scratch)
                    barrier.wait()
                    data[0] = val
                    return
                if what < 4 or what == 6:
                    a = np.ctypeslib.as_array(data, shape=(v.n_relevant * n_comp,)).reshape(-1, n_comp)
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
            m = HyperbolicModule(v, params, backend=oracle.backend())
            lib.ryujin_oracle_set_exchange(m._ctx, cb, None)
            a, b = m.new_state_vector(U0_global[v.global_ids]), m.new_state_vector()
            taus = []
            for _ in range(n_updates):
                m.prepare_state_vector(a, 0.0, dirichlet_fn(v) if dirichlet_fn else None)
                taus.append(m.step(a, [], [], b))
                a, b = b, a
            out[r] = (a.download()[: v.n_owned], taus)
        except Exception as e:  # noqa: BLE001 -- surfaced in the main thread
            out[r] = e
            barrier.abort()

    threads = [threading.Thread(target=worker, args=(r,)) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
        assert not t.is_alive(), "rank thread hung"
    for r in range(n_ranks):
        if isinstance(out[r], Exception):
            raise out[r]
    U = np.empty_like(U0_global)
    for r in range(n_ranks):
        U[views[r].global_ids[: views[r].n_owned]] = out[r][0]
    return U, [out[r][1] for r in range(n_ranks)]

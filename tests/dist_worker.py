"""Worker for the world_size>1 CPU tests (launched by torch.distributed.run, gloo backend):
every rank owns an x-slab of the mesh (ryujin_amd.offline partition), runs the CPU oracle with the
ghost exchange done over torch.distributed -- vector ghosts, matrix ghost rows, min/or reductions
at exactly the reference's synchronisation points -- and rank 0 stores the
gathered result. The parent test compares it with a single-rank run.

usage: dist_worker.py <out.npz> <cells_per_unit> <n_updates> [euler|aeos]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_py  # noqa: E402
from ryujin_amd import HyperbolicModule, capi, offline  # noqa: E402
from ryujin_amd.initial_states import euler_uniform  # noqa: E402


def main():
    out_path, cpu, n_updates = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    equation = capi.EQ_EULER_AEOS if (len(sys.argv) > 4 and sys.argv[4] == "aeos") else capi.EQ_EULER
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()

    spec = offline.mach3_step_2d(cpu, n_ranks=world, rank=rank)
    off = offline.SyntheticOffline(spec)
    o = off.c.contents
    n_nbr = o.n_nbr
    nbr = [o.nbr_rank[q] for q in range(n_nbr)]
    send_off = [o.send_off[q] for q in range(n_nbr + 1)]
    recv_off = [o.recv_off[q] for q in range(n_nbr + 1)]
    send_idx = np.array([o.send_idx[q] for q in range(send_off[-1])], dtype=np.int64)
    row_send_off = [o.row_send_off[q] for q in range(n_nbr + 1)]
    ptr = off.row_starts.astype(np.int64)
    row_send_pos = np.array([ptr[o.row_send_row[q]] + o.row_send_col[q] for q in range(row_send_off[-1])],
                            dtype=np.int64)

    def exchange(user, what, data, n_comp):
        if what in (10, 11):
            t = torch.tensor([data[0]], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN if what == 10 else dist.ReduceOp.MAX)
            data[0] = float(t[0])
            return
        reqs, recvs = [], []
        if what < 4 or what == 6:   # vector ghosts: dealii Partitioner::update_ghost_values
            v = np.ctypeslib.as_array(data, shape=(off.n_relevant * n_comp,)).reshape(-1, n_comp)
            for q in range(n_nbr):
                buf = torch.from_numpy(np.ascontiguousarray(v[send_idx[send_off[q]:send_off[q + 1]]]))
                reqs.append(dist.isend(buf, nbr[q], tag=what))
                r = torch.empty((recv_off[q + 1] - recv_off[q], n_comp), dtype=torch.float64)
                recvs.append((q, r, dist.irecv(r, nbr[q], tag=what)))
            for q, r, h in recvs:
                h.wait()
                v[recv_off[q]:recv_off[q + 1]] = r.numpy()
        else:          # matrix ghost rows: SparseMatrixSIMD::update_ghost_rows
            m = np.ctypeslib.as_array(data, shape=(int(ptr[-1]),))
            for q in range(n_nbr):
                buf = torch.from_numpy(np.ascontiguousarray(m[row_send_pos[row_send_off[q]:row_send_off[q + 1]]]))
                reqs.append(dist.isend(buf, nbr[q], tag=what))
                lo, hi = int(ptr[recv_off[q]]), int(ptr[recv_off[q + 1]])
                r = torch.empty(hi - lo, dtype=torch.float64)
                recvs.append((lo, hi, r, dist.irecv(r, nbr[q], tag=what)))
            for lo, hi, r, h in recvs:
                h.wait()
                m[lo:hi] = r.numpy()
        for h in reqs:
            h.wait()

    cb = oracle_py.EXCHANGE_FN(exchange)
    lib = oracle_py.load()
    m = HyperbolicModule(off, equation=equation, backend=oracle_py.backend())
    lib.ryujin_oracle_set_exchange(m._ctx, cb, None)
    m.cfl = 0.9
    U0 = euler_uniform(off.positions)
    U0 *= 1.0 + 1e-3 * np.sin(7.0 * off.positions[:, :1] + 3.0 * off.positions[:, 1:2])  # deterministic, rank independent
    dirichlet = euler_uniform(off.b_positions)
    a, b = m.new_state_vector(U0), m.new_state_vector()
    taus = []
    for _ in range(n_updates):
        m.prepare_state_vector(a, 0.0, dirichlet)
        taus.append(m.step(a, [], [], b))
        a, b = b, a
    U = a.download()[: off.n_owned]
    gid = off.global_ids[: off.n_owned].astype(np.int64)
    gathered = [None] * world
    dist.all_gather_object(gathered, (gid, U, taus, m.alpha()[: off.n_owned]))
    if rank == 0:
        np.savez(out_path, gid=np.concatenate([g[0] for g in gathered]),
                 U=np.concatenate([g[1] for g in gathered]),
                 alpha=np.concatenate([g[3] for g in gathered]),
                 taus=np.array([g[2] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Worker of the multi-GPU RCCL test (launched by torch.distributed.run, one rank per GPU): every rank owns an
x-slab of the mesh, creates its HIP context over a real RCCL communicator (ncclCommInitRank from the id that
rank 0 broadcasts over gloo) and runs (a) stage-wise forward-Euler updates, (b) device-resident SSPRK33 steps
with the deferred collectives, on the 2-D Mach-3 step and on the 3-D cylinder channel. Rank 0 stores the
gathered owned entries; the parent test compares them with a single-GPU run of the same problem.

usage: rccl_worker.py <out.npz> <case> <n_updates>     case: step2d:<cells per unit> | cylinder3d:<cells per unit>
       rccl_worker.py <prefix> <case> <n_warm> intermediates[:system]
           every rank stores <prefix>.rank<r>.npz: its arrays after one update incl. ghost range / ghost rows, for the
           rank-by-rank comparison with the partitioned oracle (":system": system-scope events)"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ryujin_amd import HyperbolicModule, capi, offline  # noqa: E402
from ryujin_amd.initial_states import euler_uniform  # noqa: E402


def make_spec(case, n_ranks=1, rank=0):
    kind, n = case.split(":")
    if kind == "step2d":
        return offline.mach3_step_2d(int(n), n_ranks=n_ranks, rank=rank)
    return offline.cylinder_channel_3d(int(n), length_units=1.5, n_ranks=n_ranks, rank=rank)


def initial(off):
    U0 = euler_uniform(off.positions)
    return U0 * (1.0 + 1e-3 * np.sin(7.0 * off.positions[:, :1] + 3.0 * off.positions[:, 1:2]))


def run(off, comm, device, n_updates):
    """Returns (global ids, U after n_updates stage-wise updates + 2 SSPRK33 steps, taus, alpha, integrals)."""
    m = HyperbolicModule(off, equation=capi.EQ_EULER, backend="hip", comm=comm, device=device)
    m.cfl = 0.9
    dirichlet = euler_uniform(off.b_positions) if off.n_bdry else None
    a, b = m.new_state_vector(initial(off)), m.new_state_vector()
    taus = []
    for _ in range(n_updates):
        m.prepare_state_vector(a, 0.0, dirichlet)
        taus.append(m.step(a, [], [], b))
        a, b = b, a
    alpha = m.alpha()[: off.n_owned].copy()
    temps = [b, m.new_state_vector(), m.new_state_vector()]
    for _ in range(2):
        taus.append(m.time_step("ssprk 33", a, temps, dirichlet))
    integrals = m.integrals(a)
    return (off.global_ids[: off.n_owned].astype(np.int64), a.download()[: off.n_owned], np.array(taus), alpha,
            integrals)


def intermediates(off, comm, device, n_warm, out_prefix, rank, switches=None):
    """Develop the flow for n_warm updates, then run ONE update and store every array the rank holds afterwards --
    ghost range and ghost rows included (tests/helpers_partitioned.py) -- together with the rank's local state
    before that update: the parent runs the partitioned ORACLE from those states and compares rank by rank."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers_partitioned import one_update_with_intermediates
    p = capi.Params()
    capi.load_hip().ryujin_hip_default_params(C.byref(p), capi.EQ_EULER, off.dim)
    p.cfl = 0.9
    for name, value in (switches or {}).items():
        setattr(p, name, value)
    m = HyperbolicModule(off, p, backend="hip", comm=comm, device=device)
    dirichlet = euler_uniform(off.b_positions) if off.n_bdry else None
    a, b = m.new_state_vector(initial(off)), m.new_state_vector()
    for _ in range(n_warm):
        m.prepare_state_vector(a, 0.0, dirichlet)
        m.step(a, [], [], b)
        a, b = b, a
    U_local = a.download()
    res = one_update_with_intermediates([U_local], lambda part: dirichlet)(m, off, 0)
    np.savez(f"{out_prefix}.rank{rank}.npz", U_local=U_local, **res)
    info = m.exchange_info()
    m.close()
    return info


def main_stub(rendezvous):
    """The same worker as a plain process on a ONE-GPU box: every rank on device 0, tests/cpp/librccl_stub.so
    LD_PRELOADed in front of RCCL (tests/test_rccl_stub.py). No torch: rank and world size from the environment, the
    unique id and the results through files in `rendezvous`."""
    import time
    out_path, case, n_updates = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    capi.load_synth()
    lib = capi.load_hip()
    # the double is what the library's RCCL calls bind to (LD_PRELOAD): RCCL proper would refuse ranks that share a device
    err = C.CDLL(None).ncclGetErrorString
    err.restype = C.c_char_p
    assert b"rccl stub" in err(4), "tests/cpp/librccl_stub.so is not in front of librccl.so"
    uid = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
    uid_file = os.path.join(rendezvous, "unique_id")
    if rank == 0:
        assert lib.ryujin_hip_comm_unique_id(uid) == 0, lib.ryujin_hip_last_error()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid.raw)
        os.rename(uid_file + ".tmp", uid_file)
    else:
        while not os.path.exists(uid_file):
            time.sleep(0.01)
        uid = C.create_string_buffer(open(uid_file, "rb").read(), capi.UNIQUE_ID_BYTES)
    comm = C.c_void_p()
    rc = lib.ryujin_hip_comm_init(C.byref(comm), uid, rank, world, 0)
    assert rc == 0, lib.ryujin_hip_last_error()
    off = offline.SyntheticOffline(make_spec(case, world, rank))
    v = [C.c_int(-1) for _ in range(5)]
    lib.ryujin_hip_comm_info(comm, *[C.byref(x) for x in v])
    assert v[3].value == world, ("ncclCommCount", v[3].value, world)
    if len(sys.argv) > 4 and sys.argv[4].startswith("intermediates"):
        switches = {"system_scope_events": 1} if sys.argv[4].endswith(":system") else None
        info = intermediates(off, comm, 0, n_updates, out_path, rank, switches)
        assert info["n_exchanges"] >= 5 * (n_updates + 1), info
    else:
        gid, U, taus, alpha, integrals = run(off, comm, 0, n_updates)
        np.savez(f"{out_path}.rank{rank}.npz", gid=gid, U=U, taus=taus, alpha=alpha, integrals=integrals)
    lib.ryujin_hip_comm_destroy(comm)


def main():
    if os.environ.get("RYUJIN_RCCL_STUB_DIR"):
        return main_stub(os.environ["RYUJIN_RCCL_STUB_DIR"])
    # torch only here: the parent test imports this module for run() / make_spec() into a process that has
    # libryujin_hip.so loaded already, and torch coming second would bring a second ROCm runtime along
    # (heap corruption at exit). In the worker torch comes FIRST, as in bench.py.
    import torch
    import torch.distributed as dist
    out_path, case, n_updates = sys.argv[1], sys.argv[2], int(sys.argv[3])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    local_rank = int(os.environ.get("LOCAL_RANK", rank))
    n_visible = torch.cuda.device_count()
    assert n_visible >= world, f"{world} ranks need {world} GPUs, {n_visible} visible"
    device = local_rank % n_visible

    if rank == 0:
        capi.load_synth()
        lib = capi.load_hip()
    dist.barrier()
    lib = capi.load_hip()
    uid = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
    if rank == 0:
        assert lib.ryujin_hip_comm_unique_id(uid) == 0, lib.ryujin_hip_last_error()
    t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
    dist.broadcast(t, src=0)
    uid = C.create_string_buffer(bytes(t.tolist()), capi.UNIQUE_ID_BYTES)
    comm = C.c_void_p()
    rc = lib.ryujin_hip_comm_init(C.byref(comm), uid, rank, world, device)
    assert rc == 0, lib.ryujin_hip_last_error()

    off = offline.SyntheticOffline(make_spec(case, world, rank))
    if len(sys.argv) > 4 and sys.argv[4].startswith("intermediates"):
        # rccl_worker.py <prefix> <case> <n_warm> intermediates[:system]  -- rank-by-rank comparison with the oracle
        switches = {"system_scope_events": 1} if sys.argv[4].endswith(":system") else None
        info = intermediates(off, comm, device, n_updates, out_path, rank, switches)
        v = [C.c_int(-1) for _ in range(5)]
        lib.ryujin_hip_comm_info(comm, *[C.byref(x) for x in v])
        assert v[3].value == world, ("ncclCommCount", v[3].value, world)
        assert info["n_exchanges"] >= 5 * (n_updates + 1), info
        dist.barrier()
        lib.ryujin_hip_comm_destroy(comm)
        return
    gid, U, taus, alpha, integrals = run(off, comm, device, n_updates)

    def gather(x):
        objs = [None] * world if rank == 0 else None
        dist.gather_object(x, objs, dst=0)
        return objs

    gids, Us, tauss, alphas, ints = gather(gid), gather(U), gather(taus), gather(alpha), gather(integrals)
    if rank == 0:
        np.savez(out_path, gid=np.concatenate(gids), U=np.concatenate(Us), taus=np.stack(tauss),
                 alpha=np.concatenate(alphas), integrals=np.stack(ints))
    dist.barrier()
    lib.ryujin_hip_comm_destroy(comm)


if __name__ == "__main__":
    main()

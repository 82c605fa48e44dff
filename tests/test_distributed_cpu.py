"""N>1 path on CPU: world_size 2 and 3, gloo, real processes (torch.distributed.run), as the reference's
own distributed tests use real `mpirun -np 4`. The partitioned run must reproduce the
single-rank run: same tau on every rank, U to round-off (local numbering -- hence summation order --
differs between partitions), cf. tests/euler/check-mass-conservation_02.mpirun={1,4}.output."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from ryujin_amd import HyperbolicModule, capi, offline
from ryujin_amd.initial_states import euler_uniform

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_rank(oracle, cpu, n_updates, equation=capi.EQ_EULER):
    off = offline.SyntheticOffline(offline.mach3_step_2d(cpu))
    m = HyperbolicModule(off, equation=equation, backend=oracle.backend())
    m.cfl = 0.9
    U0 = euler_uniform(off.positions)
    U0 *= 1.0 + 1e-3 * np.sin(7.0 * off.positions[:, :1] + 3.0 * off.positions[:, 1:2])
    dirichlet = euler_uniform(off.b_positions)
    a, b = m.new_state_vector(U0), m.new_state_vector()
    taus = []
    for _ in range(n_updates):
        m.prepare_state_vector(a, 0.0, dirichlet)
        taus.append(m.step(a, [], [], b))
        a, b = b, a
    return off.global_ids.astype(np.int64), a.download(), np.array(taus), m.alpha()


@pytest.mark.parametrize("world,equation", [(2, "euler"), (3, "euler"), (2, "aeos")])
def test_partitioned_oracle_matches_single_rank(oracle, tmp_path, world, equation):
    """world_size 2 and 3 over gloo; the EulerAEOS variant exchanges four precomputed values per DoF after
    each of its two precomputation cycles."""
    cpu, n_updates = 20, 4
    out = str(tmp_path / "dist.npz")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), out, str(cpu), str(n_updates), equation]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = np.load(out)
    gid, U, taus, alpha = _single_rank(oracle, cpu, n_updates,
                                       capi.EQ_EULER_AEOS if equation == "aeos" else capi.EQ_EULER)
    # every rank used the same tau, and it is the single-rank tau
    assert np.all(np.abs(d["taus"] - taus[None, :]) <= 1e-13 * taus[None, :])
    order_ref = np.argsort(gid)
    order = np.argsort(d["gid"])
    assert np.array_equal(gid[order_ref], d["gid"][order])   # ownership is a partition
    scale = np.abs(U).max(axis=0)
    err = np.abs(d["U"][order] - U[order_ref]) / scale
    assert err.max() < 1e-12, err.max()
    assert np.abs(d["alpha"][order] - alpha[order_ref]).max() < 1e-10

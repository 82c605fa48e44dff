"""The RCCL transport with MORE THAN ONE RANK: real processes, one per GPU, ncclSend/ncclRecv ghost exchange
over xGMI, ncclAllReduce of tau_max and of the restart flags (SURVEY.md section 8 row a-13 / e). RCCL refuses
two ranks on one device, so these tests need >= 2 GPUs and skip on the single-GPU box; there the same stream /
event choreography runs through the in-process transport (test_gpu_parity.py::test_partitioned_*).
The reference pins its exchange the same way, with real `mpirun -np 4` runs
(tests/euler/check-mass-conservation_02.mpirun=4.output)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.multigpu]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    """hipGetDeviceCount through the runtime libryujin_hip.so is linked against. (NOT torch.cuda: importing
    torch into a process that already loaded libryujin_hip.so brings a second copy of the ROCm runtime with it,
    which corrupts the heap at exit; bench.py imports torch first for the same reason.)"""
    import ctypes as C
    try:
        from ryujin_amd import capi
        n = C.c_int(0)
        # dlsym on the library's handle also searches its dependencies: the very runtime it is linked against
        if capi.load_hip().hipGetDeviceCount(C.byref(n)) != 0:
            return 0
        return n.value
    except Exception:
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_hung = []  # a launch that had to be killed: the remaining cases would hang the same way, skip them


def _run_group(cmd, env=None, timeout=420):
    """Run `cmd` in its own process group; on a timeout kill the whole group (launcher AND its rank processes,
    which would otherwise keep their GPUs) and remember that the RCCL path hung."""
    import signal
    if _hung:
        pytest.skip(f"an earlier multi-GPU launch hung ({_hung[0]})")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)  # the group this call created, nothing else
        out, err = proc.communicate()
        _hung.append(" ".join(cmd[-4:]))
        pytest.fail(f"multi-GPU launch timed out after {timeout} s: {' '.join(cmd)}\n{err[-4000:]}")
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def _launch(world, script_args, timeout=420):
    env = dict(os.environ, OMP_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *script_args]
    return _run_group(cmd, env, timeout)


@pytest.mark.parametrize("world,case", [(2, "step2d:40"), (4, "step2d:40"), (8, "step2d:60"), (2, "cylinder3d:16"),
                                        (8, "cylinder3d:24")])
def test_rccl_partitioned_run_matches_single_gpu(tmp_path, world, case):
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs (RCCL refuses several ranks on one device)")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rccl_worker
    from ryujin_amd import offline

    n_updates = 5
    out = str(tmp_path / "rccl.npz")
    res = _launch(world, [os.path.join(ROOT, "tests", "rccl_worker.py"), out, case, str(n_updates)])
    assert res.returncode == 0, res.stderr[-4000:]
    d = np.load(out)
    gid, U, taus, alpha, integrals = rccl_worker.run(offline.SyntheticOffline(rccl_worker.make_spec(case)), None, 0,
                                                     n_updates)
    # every rank used the same tau (the all-reduced minimum), and it is the single-GPU tau
    assert np.all(np.abs(d["taus"] - taus[None, :]) <= 1e-12 * taus[None, :])
    o1, o2 = np.argsort(gid), np.argsort(d["gid"])
    assert np.array_equal(gid[o1], d["gid"][o2])          # ownership is a partition of the mesh
    scale = np.abs(U).max(axis=0)
    # different local numbering = different summation order: round-off, amplified over 5 + 6 updates
    assert (np.abs(d["U"][o2] - U[o1]) / scale).max() < 1e-11
    assert np.abs(d["alpha"][o2] - alpha[o1]).max() < 1e-10
    # ncclAllReduce(sum) of the conservation monitor: every rank holds the global integrals
    assert np.allclose(d["integrals"], integrals[None, :], rtol=1e-12)


@pytest.mark.parametrize("world,case,events", [(2, "step2d:30", "device"), (3, "step2d:30", "device"),
                                               (3, "step2d:30", "system"), (4, "cylinder3d:8", "device"),
                                               (8, "cylinder3d:16", "device")])
def test_rccl_ranks_against_the_partitioned_oracle(oracle, tmp_path, world, case, events):
    """The decisive check for the RCCL leg: every rank of a real multi-process run against THE SAME rank of the
    partitioned oracle after one update on identical inputs -- d_ij on ghost columns, alpha and r on the ghost
    range, l_ij / l'_ij including the ghost rows received over xGMI (which must be bitwise the entries the sender
    holds at its send positions) -- with device-scope and with system-scope events between the two streams. The same
    comparison runs on one GPU with the in-process transport (tests/test_partitioned_vs_oracle.py); the worker also
    asserts ncclCommCount == world and at least 5 exchanges per update."""
    if _n_gpus() < world:
        pytest.skip(f"needs {world} GPUs (RCCL refuses several ranks on one device)")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rccl_worker
    from helpers_partitioned import compare_rank_files
    from ryujin_amd import capi, offline
    from ryujin_amd.initial_states import euler_uniform
    prefix = str(tmp_path / "ranks")
    mode = "intermediates" + (":system" if events == "system" else "")
    res = _launch(world, [os.path.join(ROOT, "tests", "rccl_worker.py"), prefix, case, "12", mode])
    assert res.returncode == 0, res.stderr[-4000:]
    parts = [offline.SyntheticOffline(rccl_worker.make_spec(case, world, r)) for r in range(world)]

    def make_params():
        p = oracle.default_params(capi.EQ_EULER, parts[0].dim)
        p.cfl = 0.9
        return p
    compare_rank_files(oracle, parts, prefix, make_params,
                       lambda part: euler_uniform(part.b_positions) if part.n_bdry else None, parts[0].dim + 2)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_self_launches_its_ranks(n):
    """`python bench.py --gpus N` invoked like the single-GPU line: launches N ranks itself and prints ONE
    JSON line with n_gpus == N."""
    if _n_gpus() < n:
        pytest.skip(f"needs {n} GPUs")
    res = _run_group([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "6",
                      "--warmup", "3", "--develop-time", "0.3", "--develop", "30", "--cells-per-unit", "200", "--watchdog", "360"])
    assert res.returncode == 0, res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["scaling"] == "weak" and d["value"] > 0 and d["n_warnings"] == 0
    # what RCCL itself saw, and the verdict of the events self-check
    assert d["rccl"]["ranks"] == n and len(d["rccl"]["n_neighbours_per_rank"]) == n
    assert d["rccl"]["exchanges_per_update"] == 5.0 and d["events"]["check"] is not None

"""The multi-rank branch of the library executed by REAL PROCESSES on a one-GPU box (SURVEY.md section 8 rows a-13 / e).

RCCL refuses several ranks on one device, so tests/test_multigpu_rccl.py skips on the single-GPU boxes every round of
this build has had. Here the same worker (tests/rccl_worker.py), the same cases and the same comparisons run with
tests/cpp/librccl_stub.so LD_PRELOADed in front of librccl.so: a test double that serves the twelve RCCL entry points the
library imports, stream ordered like RCCL's own, for processes that share device 0 (messages staged through POSIX
shared memory; tests/cpp/rccl_stub.cc). What this executes that the in-process transport of
tests/test_partitioned_vs_oracle.py does not: ryujin_hip_comm_init over a broadcast unique id, the `!comm->local`
branches of exchange_vector / exchange_matrix_rows (ncclGroupStart, ncclSend / ncclRecv counts, offsets and peers per
neighbour, ncclGroupEnd on the exchange stream), ncclAllReduce(min) of tau_max, (max) of the restart flags and (sum) of
the conservation monitor on the compute stream, and the device-resident SSPRK33 driver with its deferred collectives
-- in separate address spaces, every rank with its own context. What it does not: RCCL itself, xGMI, several devices."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "cpp", "librccl_stub.so")


def build_stub():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "tests", "cpp", "rccl_stub.cc")
    if os.path.exists(STUB) and os.path.getmtime(STUB) >= os.path.getmtime(src):
        return STUB
    subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", src,
                    "-lrt", "-o", STUB], check=True, capture_output=True)
    return STUB


def launch(world, args, tmp_path, timeout=600):
    """`world` processes of rccl_worker.py on device 0, the stub in front of RCCL; kills the whole group on a hang"""
    import signal
    build_stub()
    rendezvous = str(tmp_path / "rendezvous")
    os.makedirs(rendezvous, exist_ok=True)
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LD_PRELOAD=STUB, RYUJIN_RCCL_STUB_DIR=rendezvous,
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), *args], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      start_new_session=True))
    errs = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=timeout)
            errs.append((p.returncode, err))
    except subprocess.TimeoutExpired:
        for p in procs:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        pytest.fail(f"{world} ranks over the RCCL stub timed out after {timeout} s")
    for rc, err in errs:
        assert rc == 0, err[-4000:]


@pytest.mark.parametrize("world,case", [(2, "step2d:40"), (4, "step2d:40"), (8, "step2d:60"), (3, "cylinder3d:12"),
                                        (8, "cylinder3d:24")])
def test_stub_partitioned_run_matches_single_gpu(tmp_path, world, case):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rccl_worker
    from ryujin_amd import offline
    n_updates = 5
    out = str(tmp_path / "rccl")
    launch(world, [out, case, str(n_updates)], tmp_path)
    ranks = [np.load(f"{out}.rank{r}.npz") for r in range(world)]
    d = {k: (np.stack([x[k] for x in ranks]) if k in ("taus", "integrals") else np.concatenate([x[k] for x in ranks]))
         for k in ("gid", "U", "taus", "alpha", "integrals")}
    gid, U, taus, alpha, integrals = rccl_worker.run(offline.SyntheticOffline(rccl_worker.make_spec(case)), None, 0,
                                                     n_updates)
    # every rank used the same tau (the all-reduced minimum), and it is the single-GPU tau
    assert np.all(np.abs(d["taus"] - taus[None, :]) <= 1e-12 * taus[None, :])
    o1, o2 = np.argsort(gid), np.argsort(d["gid"])
    assert np.array_equal(gid[o1], d["gid"][o2])          # ownership is a partition of the mesh
    scale = np.abs(U).max(axis=0)
    assert (np.abs(d["U"][o2] - U[o1]) / scale).max() < 1e-11
    assert np.abs(d["alpha"][o2] - alpha[o1]).max() < 1e-10
    # all-reduce(sum) of the conservation monitor: every rank holds the global integrals
    assert np.allclose(d["integrals"], integrals[None, :], rtol=1e-12)


@pytest.mark.parametrize("world,case,events", [(2, "step2d:30", "device"), (3, "step2d:30", "system"),
                                               (4, "cylinder3d:8", "device"), (8, "cylinder3d:16", "device")])
def test_stub_ranks_against_the_partitioned_oracle(oracle, tmp_path, world, case, events):
    """every rank of the multi-process run against THE SAME rank of the partitioned oracle after one update on identical
    inputs, ghost range and received ghost rows included (as test_multigpu_rccl.py does over real RCCL)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rccl_worker
    from helpers_partitioned import compare_rank_files
    from ryujin_amd import capi, offline
    from ryujin_amd.initial_states import euler_uniform
    prefix = str(tmp_path / "ranks")
    mode = "intermediates" + (":system" if events == "system" else "")
    launch(world, [prefix, case, "12", mode], tmp_path)
    parts = [offline.SyntheticOffline(rccl_worker.make_spec(case, world, r)) for r in range(world)]

    def make_params():
        p = oracle.default_params(capi.EQ_EULER, parts[0].dim)
        p.cfl = 0.9
        return p
    compare_rank_files(oracle, parts, prefix, make_params,
                       lambda part: euler_uniform(part.b_positions) if part.n_bdry else None, parts[0].dim + 2)


def _bench(args, preload=True, timeout=900):
    """`python bench.py <args>` in its own process group (killed as a group on a hang), the stub in front of RCCL"""
    import json
    import signal
    build_stub()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if preload:
        env["LD_PRELOAD"] = STUB
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *args]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)
        out, err = proc.communicate()
        pytest.fail(f"bench.py {' '.join(args)} timed out after {timeout} s\n{err[-4000:]}")
    assert proc.returncode == 0, err[-6000:]
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.parametrize("n,extra", [(2, ["--cells-per-unit", "200"]),
                                     (8, ["--workload", "cylinder3d", "--size", "24"])])
def test_bench_scaling_run_dress_rehearsal_on_one_gpu(n, extra):
    """`bench.py --gpus N` END TO END before the driver's one-shot 8-GPU run -- the
    self-launch through torch.distributed.run, per-rank slab generation, every rank's own coarse run, the RCCL
    communicator from the id broadcast over gloo, the device- vs system-scope events check, the `rccl` object counted
    by the library, the watchdog, ONE JSON line from rank 0 -- with all ranks on device 0 over the RCCL test double.
    Asserts what tests/test_multigpu_rccl.py::test_bench_self_launches_its_ranks asserts on a multi-GPU box."""
    d = _bench(["--gpus", str(n), "--steps", "6", "--warmup", "3", "--develop-time", "0.3", "--develop", "30",
                "--reps", "2", "--watchdog", "800", *extra])
    assert d["n_gpus"] == n and d["scaling"] == "weak" and d["value"] > 0 and d["n_warnings"] == 0
    assert d["rccl"]["ranks"] == n and len(d["rccl"]["n_neighbours_per_rank"]) == n
    # x-slabs: the two end ranks have one neighbour, the others two
    assert sorted(d["rccl"]["n_neighbours_per_rank"]) == sorted([1, 1] + [2] * (n - 2))
    assert d["rccl"]["exchanges_per_update"] == 5.0 and abs(d["rccl"]["allreduces_per_update"] - 2.0 / 3.0) < 1e-12
    assert d["events"]["check"] is not None and d["events"]["kind"] == "device", d["events"]
    # weak scaling: every rank holds the same share, value is the whole job's
    cfg = d["config"]
    assert abs(cfg["gridpoints_total"] - n * cfg["gridpoints_per_gpu"]) <= 0.1 * cfg["gridpoints_total"]  # whole node planes per rank: coarse on a small mesh
    assert abs(d["value"] - cfg["dofs_total"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6) <= 1e-6 * d["value"]
    assert "cpu_baseline" not in d  # rank 0 at N = 1 only


def test_bench_line_under_the_launcher_agrees_with_the_plain_line():
    """the N = 1 point of the scaling curve (torch.distributed + an RCCL communicator of one rank, --force-dist) must
    be the bench line: same mesh, same state, same kernels; ms_per_step within the spread of two runs on one box"""
    common = ["--steps", "30", "--warmup", "6", "--develop-time", "0.3", "--develop", "60", "--cells-per-unit", "400",
              "--no-cpu-baseline", "--binding", "device"]
    plain = _bench(common, preload=False)
    dist = _bench(["--force-dist", *common], preload=False)  # real RCCL: one rank on one device is allowed
    assert dist["n_gpus"] == 1 and dist["rccl"]["ranks"] == 1 and "rccl" not in plain
    assert dist["config"]["gridpoints_total"] == plain["config"]["gridpoints_total"]
    assert dist["config"]["simulated_time_at_end"] == plain["config"]["simulated_time_at_end"]  # same taus, bit for bit
    assert abs(dist["ms_per_step"] / plain["ms_per_step"] - 1.0) < 0.08, (dist["ms_per_step"], plain["ms_per_step"])

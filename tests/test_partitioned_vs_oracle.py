"""Partitioned HIP against the partitioned ORACLE, rank by rank, intermediates and ghost ranges included
(SURVEY.md section 8 rows a-13 / e; BASELINE configs[3] is the 8-rank cylinder).

The other partitioned tests compare a partitioned HIP run with a single-rank HIP run on U after several updates:
a ghost value that is wrong but cancels in U (a symmetric error in l_ij / l_ji, a ghost alpha that only enters a
rim pair) would pass there. Here every rank of a 3-rank HIP run (in-process transport = the event graph of the
RCCL leg) is compared with THE SAME rank of a 3-rank oracle run after one update on identical inputs:
  U after the boundary conditions and the precomputed values over the whole locally relevant range,
  alpha and r on the ghost range (received), d_ij of the owned rows including their ghost columns,
  bounds, P_ij, and l_ij / l'_ij including the ghost ROWS received from the neighbours,
with the tolerances of helpers_parity.py. Meshes: the 2-D Mach-3 step and the 3-D cylinder channel in miniature
(Dirichlet + do-nothing + slip + staircase cylinder with coupling boundary pairs), x-slabs of the generator; and
the unstructured P1 disk cut into 4 sectors (up to 3 neighbours per rank, nodes exported to several ranks)."""
import numpy as np
import pytest

from helpers_partitioned import (compare_ghost_rows, compare_rank, global_scales, one_update_with_intermediates, run_hip_ranks,
                                 run_oracle_ranks)
from ryujin_amd import capi, offline
from ryujin_amd.initial_states import euler_radial_contrast, euler_uniform


def _params(oracle, equation, dim, cfl=0.9):
    def make():
        p = oracle.default_params(equation, dim)
        p.cfl = cfl
        return p
    return make


def _develop_on(run, parts, make_params, U_init, n_warm, dirichlet_of):
    """n_warm updates of the partitioned run `run`; returns every rank's local state (owned + ghost range)."""
    def body(m, part, r):
        a, b = m.new_state_vector(U_init[r]), m.new_state_vector()
        for _ in range(n_warm):
            m.prepare_state_vector(a, 0.0, dirichlet_of(part) if dirichlet_of else None)
            m.step(a, [], [], b)
            a, b = b, a
        return a.download()
    return run(parts, make_params, body)


def _step_parts(n_ranks, cpu=30):
    return [offline.SyntheticOffline(offline.mach3_step_2d(cpu, n_ranks=n_ranks, rank=r)) for r in range(n_ranks)]


def _mach3_initial(parts):
    out = []
    for p in parts:
        U0 = euler_uniform(p.positions)
        out.append(U0 * (1.0 + 1e-3 * np.sin(7.0 * p.positions[:, :1] + 3.0 * p.positions[:, 1:2])))
    return out


def _mach3_dirichlet(part):
    return euler_uniform(part.b_positions) if part.n_bdry else None


def test_partitioned_oracle_helper_reproduces_the_single_rank_oracle(oracle):
    """CPU: the rank-by-rank oracle runner of this file against a single-rank oracle run (the partitioned oracle
    is the yardstick of the GPU tests below; tests/test_distributed_cpu.py checks the same over gloo processes)."""
    from ryujin_amd import HyperbolicModule
    n_ranks, n_updates = 3, 3
    parts = _step_parts(n_ranks, cpu=20)
    make = _params(oracle, capi.EQ_EULER, 2)
    states = _develop_on(lambda *a: run_oracle_ranks(oracle, *a), parts, make, _mach3_initial(parts), n_updates,
                         _mach3_dirichlet)
    single = offline.SyntheticOffline(offline.mach3_step_2d(20))
    m = HyperbolicModule(single, make(), backend=oracle.backend())
    a, b = m.new_state_vector(_mach3_initial([single])[0]), m.new_state_vector()
    for _ in range(n_updates):
        m.prepare_state_vector(a, 0.0, _mach3_dirichlet(single))
        m.step(a, [], [], b)
        a, b = b, a
    U_ref = a.download()
    lookup = {int(g): i for i, g in enumerate(single.global_ids)}
    scale = np.abs(U_ref).max(axis=0)
    for r, part in enumerate(parts):
        rows = [lookup[int(g)] for g in part.global_ids[: part.n_owned]]
        assert (np.abs(states[r][: part.n_owned] - U_ref[rows]) / scale).max() < 1e-12


def _compare_partitioned(oracle, parts, equation, dim, U_init, dirichlet_of, n_warm, cfl=0.9):
    make = _params(oracle, equation, dim, cfl)
    k = {capi.EQ_EULER: dim + 2, capi.EQ_SHALLOW_WATER: dim + 1}[equation]
    # develop the flow on the GPU ranks, hand every rank's state to both backends, compare one update
    states = _develop_on(run_hip_ranks, parts, make, U_init, n_warm, dirichlet_of)
    body = one_update_with_intermediates(states, dirichlet_of)
    hip = run_hip_ranks(parts, make, body)
    ref = run_oracle_ranks(oracle, parts, make, body)
    scales = global_scales(parts, ref, k)
    accepted = [compare_rank(part, hip[r], ref[r], k, label=f"rank {r}", scales=scales)
                for r, part in enumerate(parts)]
    assert compare_ghost_rows(parts, hip, ref, accepted) > 0
    # the comparison of the ghost rows of l_ij is not vacuous: some ghost-row entry was actually limited
    ghost_l = np.concatenate([ref[r]["lij_next"][int(parts[r].row_starts[parts[r].n_owned]):] for r in range(len(parts))])
    assert ghost_l.size and ghost_l.min() < 1.0
    return hip, ref


@pytest.mark.gpu
def test_partitioned_step_mesh_rank_by_rank_against_the_oracle(oracle):
    parts = _step_parts(3, cpu=30)
    _compare_partitioned(oracle, parts, capi.EQ_EULER, 2, _mach3_initial(parts), _mach3_dirichlet, n_warm=40)


@pytest.mark.gpu
def test_partitioned_miniature_cylinder_rank_by_rank_against_the_oracle(oracle):
    """BASELINE configs[3] in miniature: the 3-D cylinder channel (h = 1/6), 3 x-slabs."""
    parts = [offline.SyntheticOffline(offline.cylinder_channel_3d(6, n_ranks=3, rank=r)) for r in range(3)]
    U_init = []
    for p in parts:
        U0 = euler_uniform(p.positions)
        U_init.append(U0 * (1.0 + 1e-3 * np.sin(5.0 * p.positions[:, :1] + 3.0 * p.positions[:, 1:2] +
                                                2.0 * p.positions[:, 2:3])))
    _compare_partitioned(oracle, parts, capi.EQ_EULER, 3, U_init, _mach3_dirichlet, n_warm=25)


@pytest.mark.gpu
@pytest.mark.parametrize("equation", ["euler", "shallow_water"])
def test_partitioned_unstructured_sectors_rank_by_rank_against_the_oracle(oracle, equation):
    """An arbitrary partition (up to 3 neighbours per rank, nodes exported to several ranks), Euler and shallow
    water with bathymetry."""
    from helpers_unstructured import disk_points, p1_offline, partition
    from test_oracle_unstructured import sector_owner
    off, info = p1_offline(disk_points(18))
    x = off.positions
    if equation == "euler":
        eq, Z = capi.EQ_EULER, None
        U0 = euler_radial_contrast(x, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.35,
                                   center=(0.1, -0.05))
    else:
        eq = capi.EQ_SHALLOW_WATER
        r = np.linalg.norm(x - np.array([0.1, -0.05]), axis=1)
        Z = 0.5 * np.linalg.norm(x, axis=1) ** 2 + 0.03 * np.cos(6.0 * x[:, 0])
        off.set_initial_precomputed(Z)
        U0 = np.zeros((off.n_owned, 3))
        U0[:, 0] = np.maximum(np.where(r < 0.3, 0.8, 0.4) - Z, 0.0)
    views = partition(off, info, sector_owner(x, 4), bathymetry=Z)
    assert max(v.c.contents.n_nbr for v in views) >= 2
    U_init = [U0[v.global_ids] for v in views]
    _compare_partitioned(oracle, views, eq, 2, U_init, None, n_warm=10, cfl=0.5)


@pytest.mark.gpu
def test_partitioned_q1_annulus_rank_by_rank_against_the_oracle(oracle):
    """The skewed-quadrilateral annulus (tests/helpers_q1_quads.py) cut into four sectors around a point off the
    centre: every rank holds a piece of both curved walls, boundary nodes are exported to neighbours, the sector cuts
    cross the twisted rings obliquely. Every rank against the same rank of the partitioned oracle, ghost rows
    included."""
    from helpers_q1_quads import annulus_mesh, q1_quads_offline
    from helpers_unstructured import partition
    from test_oracle_unstructured import sector_owner
    pts, quads, edges = annulus_mesh(16, 80)
    off, info = q1_quads_offline(pts, quads, edges)
    x = off.positions
    U0 = euler_radial_contrast(x, inner=(1.0, 0.0, 10.0), outer=(0.125, 0.0, 0.1), radius=0.2, center=(0.7, 0.0))
    views = partition(off, info, sector_owner(x, 4))
    assert max(v.c.contents.n_nbr for v in views) >= 2 and all(v.n_bdry > 0 for v in views)
    U_init = [U0[v.global_ids] for v in views]
    _compare_partitioned(oracle, views, capi.EQ_EULER, 2, U_init, None, n_warm=40, cfl=0.5)


@pytest.mark.gpu
def test_the_rank_file_plumbing_of_the_rccl_test_on_one_gpu(oracle, tmp_path):
    """tests/test_multigpu_rccl.py::test_rccl_ranks_against_the_partitioned_oracle needs several GPUs. Its worker
    function (rccl_worker.intermediates: develop, one update, store the rank's arrays) and its parent side
    (helpers_partitioned.compare_rank_files) run here on ONE GPU with the in-process transport, so that the first
    multi-GPU box exercises nothing but RCCL itself for the first time."""
    import ctypes as C
    import threading

    import rccl_worker
    from helpers_partitioned import compare_rank_files
    lib = capi.load_hip()
    world, case = 3, "step2d:30"
    parts = [offline.SyntheticOffline(rccl_worker.make_spec(case, world, r)) for r in range(world)]
    comms = (C.c_void_p * world)()
    assert lib.ryujin_hip_comm_init_local(comms, world, 0) == 0
    prefix, out = str(tmp_path / "ranks"), {}

    def worker(r):
        try:
            out[r] = rccl_worker.intermediates(parts[r], C.c_void_p(comms[r]), 0, 12, prefix, r)
        except BaseException as e:  # noqa: BLE001
            out[r] = e
    threads = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    for r in range(world):
        assert not isinstance(out[r], BaseException), out[r]
        assert out[r]["n_exchanges"] >= 5 * 13 and len(out[r]["neighbours"]) == (1 if r in (0, world - 1) else 2)
        lib.ryujin_hip_comm_destroy(C.c_void_p(comms[r]))

    def make_params():
        p = oracle.default_params(capi.EQ_EULER, 2)
        p.cfl = 0.9
        return p
    compare_rank_files(oracle, parts, prefix, make_params, _mach3_dirichlet, 4)

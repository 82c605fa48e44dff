#!/usr/bin/env python3
"""bench.py -- the hot path (prepare_state_vector + step<0>, one forward-Euler update per "step")
on the BASELINE.json workload: 2-D Euler Mach-3 forward-facing step, ~10M DoFs (k=4, ~2.5M Q1
gridpoints) per GPU, SSPRK33 stage sequence, synthetic structured mesh, data resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU; the mesh is lengthened N-fold and
   slab-partitioned by DoF ownership: weak scaling; ghost exchange over RCCL inside the library,
   torch.distributed(gloo) only carries the 128-byte RCCL id, the barriers and the max-reduction)

Prints ONE JSON line on rank 0. `value` = k * N_q(all ranks) * K / t_max-over-ranks / 1e6.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(dim: int, k: int, S: float, n_prec: int = 2, n_bounds: int = 3) -> dict:
    """ALGORITHMIC bytes per gridpoint-update and sweep:
    every array of the reference's 7-sweep structure touched once per sweep, neighbour gathers free."""
    d, p, b = dim, n_prec, n_bounds
    return {
        "1 prepare_state_vector": 8 * k + 8 * p,
        "2 dij_alpha": (8 * k + 8 * p + 4 * S + 8 * d * S + 8) + (8 * S + 8),
        "3 dij_diag_tau": (4 * S + 4 * S + 8 * S + 8) + 8 * S,
        "4 low_order": (8 * k + 8 * p + 8 + 16 + 4 * S + 8 * S + 8 * d * S) + (8 * k + 8 * k + 8 * b + 8 * k * S),
        "5 pij_lij": (8 * b + 16 + 8 * k + 8 * k + 4 * S + 8 * S + 8 * k * S) + (8 * k * S + 8 * S),
        "6 high_order_next_lij": (8 * k + 8 * S + 4 * S + 8 * k * S + 8 * b) + (8 * k + 8 * S),
        "7 high_order": (8 * k + 8 * S + 4 * S + 8 * k * S) + 8 * k,
    }


KERNEL_OF_SWEEP = {  # sweep name -> kernel-name prefixes in the rocprof summaries
    "2 dij_alpha": ("k_dij_alpha",), "2a alpha (k_alpha)": ("k_alpha",), "2b dij (k_dij)": ("k_dij<",),
    "3 dij_diag_tau": ("k_dij_diag",), "4 low_order": ("k_low_order",),
    "5 pij_lij": ("k_lij_stage0", "k_pij_lij"), "6 high_order_next_lij": ("k_high_order_next_cached", "k_high_order<"),
    "7 high_order": ("k_high_order_last_cached", "k_high_order<"),
}


def source_fingerprint() -> str:
    """Hash of the kernel sources (ryujin_amd/csrc/*.{hip,hpp,h}, comments and whitespace stripped): what a committed
    profile has to have been taken with for its counters to be attached to a bench line of this tree."""
    import glob
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "ryujin_amd", "csrc", "*"))):
        if f.endswith((".hip", ".hpp", ".h")):
            text = open(f, encoding="utf-8", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            h.update(os.path.basename(f).encode())
            h.update(" ".join(text.split()).encode())
    return h.hexdigest()[:16]


def _pmc_kernel_row(sweep: str, workload: str):
    """(counters of the sweep's kernel as {column: mean per dispatch}, profile file, note) from the newest committed
    rocprofv3 --pmc passes of this same command (profiles/r*_pmc.md for the bench line, profiles/r*_pmc_<workload>.md
    for the other workloads). PMC cannot be collected inside this process. A profile taken with OTHER kernel sources
    than this tree's (its `kernel sources:` line against source_fingerprint()) is refused: row None, the note says
    why."""
    import glob
    suffix = "" if workload == "step2d" else "_" + workload
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc{suffix}.md")))
    if not files:
        return None, None, "no committed PMC pass of this workload"
    rel = os.path.relpath(files[-1], ROOT)
    rows, header = {}, None
    taken_with = None
    for line in open(files[-1]):
        if line.startswith("kernel sources:"):
            taken_with = line.split(":", 1)[1].strip()
        if line.startswith("| kernel"):
            header = [c.strip() for c in line.strip().strip("|").split("|")]
        if line.startswith("| k_"):
            cells = [c.strip() for c in line.strip().strip("|").split("|")]
            try:
                cols = header if header is not None and len(header) == len(cells) else (
                    ["kernel", "n"] + [f"c{q}" for q in range(len(cells) - 3)] + ["hbm_MB"])
                rows[cells[0]] = {c: float(v) for c, v in zip(cols[1:], cells[1:])}
            except ValueError:
                pass
    now = source_fingerprint()
    if taken_with != now:
        return None, rel, (f"REFUSED as stale: {rel} was taken with kernel sources {taken_with}, this tree is {now}")
    last = sweep.startswith("7")
    best = None  # several variants of a sweep's kernel may have run: the one with the most bytes in total
    for name, row in rows.items():
        for prefix in KERNEL_OF_SWEEP.get(sweep, ()):
            if name.startswith(prefix):
                if prefix == "k_high_order<" and (("true" in name) != last):
                    continue
                if best is None or row["hbm_MB"] * row["n"] > best["hbm_MB"] * best["n"]:
                    best = row
    if best is not None:
        return best, rel, None
    return None, rel, "the profile holds no kernel of this sweep"


def pmc_traffic_bytes(sweep: str, workload: str):
    """(HBM bytes per launch of the sweep's kernel, profile file, note): (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950
    correction of MI355X_MICROARCH.md, from _pmc_kernel_row()."""
    row, rel, note = _pmc_kernel_row(sweep, workload)
    return (row["hbm_MB"] * 1e6 if row is not None else None), rel, note


def pmc_valu_issue(sweep: str, workload: str, launch_ms: float):
    """The arithmetic side of the sweep's kernel from the same counter pass: VALU instructions per wave and the
    share of the launch the FP64 pipes need just to issue them (a wave64 VALU instruction occupies its 16-lane SIMD
    for 4 cycles; 1024 SIMDs at 2.4 GHz, MI355X_MICROARCH.md -- a lower bound: divisions, square roots and 64-bit
    integer multiplies are counted at their instruction count). None without a valid profile."""
    row, _, _ = _pmc_kernel_row(sweep, workload)
    if row is None or "SQ_INSTS_VALU" not in row or "SQ_WAVES" not in row or not row["SQ_WAVES"]:
        return None
    issue_ms = row["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9) * 1e3
    return {"valu_instructions_per_wave": round(row["SQ_INSTS_VALU"] / row["SQ_WAVES"]),
            "issue_ms": issue_ms, "issue_frac": issue_ms / launch_ms,
            "note": "VALU instructions x 4 cycles / (1024 SIMDs x 2.4 GHz) over the launch duration: the share of the "
                    "launch the FP64 pipes need to issue the kernel's arithmetic (with traffic_frac: how far the two add up)"}


def host_mirrored_ssprk33(m, U0, dirichlet, t0: float, n_rk: int, pin: bool, mirror_derived: bool) -> dict:
    """The drop-in path with an UNMODIFIED ryujin caller, timed: the reference's step_ssprk_33
    (time_integrator.template.h:302-328) -- prepare_state_vector / step<0> on HOST state vectors, sadd() and swap() on
    the host -- served by the mirroring of contrib/hyperbolic_module_hip.h (here its Python twin,
    ryujin_amd.module.HostMirroredModule: the same C-ABI calls in the same order). Per update: upload U, write back
    the boundary rows, download the precomputed values, download the new U (owned rows) and alpha. PCIe-inclusive;
    never the headline value."""
    from ryujin_amd.module import HostMirroredModule, HostStateVector
    hm = HostMirroredModule(m, pin=pin, mirror_derived=mirror_derived)
    state, temp = HostStateVector(m, U0), [HostStateVector(m), HostStateVector(m)]
    t, t_sadd = t0, 0.0

    def rk_step():
        nonlocal t, t_sadd
        hm.prepare_state_vector(state, t, dirichlet)
        tau = hm.step(state, [], [], temp[0], 0.0)
        hm.prepare_state_vector(temp[0], t + tau, dirichlet)
        hm.step(temp[0], [], [], temp[1], tau)
        s0 = time.perf_counter()
        temp[1].U *= 0.25
        temp[1].U += 0.75 * state.U
        t_sadd += time.perf_counter() - s0
        hm.prepare_state_vector(temp[1], t + 0.5 * tau, dirichlet)
        hm.step(temp[1], [], [], temp[0], tau)
        s0 = time.perf_counter()
        temp[0].U *= 2.0 / 3.0
        temp[0].U += (1.0 / 3.0) * state.U
        t_sadd += time.perf_counter() - s0
        state.swap(temp[0])
        t += tau

    rk_step()  # warm-up: twins, pinning
    t_sadd = 0.0
    w0 = time.perf_counter()
    for _ in range(n_rk):
        rk_step()
    wall = time.perf_counter() - w0
    hm.close()
    n_upd = 3 * n_rk
    bytes_per_update = 8 * (m.n_relevant * m.k + m.n_owned * m.k
                            + (m.n_relevant * (m.n_prec + 1) if mirror_derived else 0))
    return {"ms_per_update": wall / n_upd * 1e3,
            "ms_per_update_without_host_sadd": (wall - t_sadd) / n_upd * 1e3,
            "pcie_bytes_per_update": bytes_per_update,
            "pcie_gb_per_s": bytes_per_update / ((wall - t_sadd) / n_upd) / 1e9,
            "pinned_in_place": pin, "mirror_precomputed_and_alpha": mirror_derived, "ssprk33_steps": n_rk}


def usable_cores() -> int:
    """Physical cores this process may use: affinity mask, cgroup CPU quota, SMT siblings folded."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:  # count one thread per physical core
        sib = set()
        for c in os.sched_getaffinity(0):
            p = f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list"
            sib.add(open(p).read().strip())
        n = min(n, max(1, len(sib)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(spec, U0, dirichlet, budget_s: float = 15.0, equation: int = 0) -> dict:
    """The CPU restatement of the reference path (oracle/, OpenMP over all host cores) timed on a
    bounded sample of the SAME workload: n forward-Euler updates of the same mesh."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    from ryujin_amd import HyperbolicModule, capi, offline
    from ryujin_amd.workloads import Ssprk33Stages

    import build_oracle
    native = os.path.join(ROOT, "oracle", "build", "libryujin_oracle_native.so")
    path = None
    try:
        path = build_oracle.build_oracle(march_native=True, out=native)
    except Exception:
        path = None  # fall back to the portable build shipped with the snapshot
    # OMP_NUM_THREADS / OMP_PROC_BIND / OMP_PLACES were exported at the top of main(), before libgomp was loaded
    cores = int(os.environ.get("OMP_NUM_THREADS", usable_cores()))
    lib = oracle_py.load(path)
    lib.ryujin_oracle_set_flush_denormals(1)  # source/main.cc:26-36
    off = offline.SyntheticOffline(spec)
    m = HyperbolicModule(off, equation=equation, backend=(lib, "ryujin_oracle_"))
    m.cfl = 0.9
    drv = Ssprk33Stages(m, U0, dirichlet)
    t0 = time.perf_counter()
    drv.update()
    one = time.perf_counter() - t0
    n = int(max(3, min(60, budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        drv.update()
    dt = time.perf_counter() - t0
    k = m.k
    return {"value": k * off.n_owned * n / dt / 1e6, "unit": "MDoF-updates/s", "cores": cores,
            "kind": "port",
            "sample": f"{n} forward-Euler updates (SSPRK33 stages) of the same mesh "
                      f"({off.n_owned} gridpoints), OpenMP on {cores} threads, "
                      f"{'-march=native' if path else 'portable'} build, {dt:.1f} s",
            "mq_per_s": off.n_owned * n / dt / 1e6}


def self_launch_command(n_gpus: int, argv: list) -> list:
    """The command line that runs this script on n_gpus ranks of one node (one process per GPU, RCCL over
    xGMI inside the library; rendezvous on 127.0.0.1 -- the container hostname may not resolve)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--cells-per-unit", type=int, default=995,
                    help="mesh resolution h=1/N of the step geometry (995 -> ~2.5M gridpoints, ~10M DoFs per GPU)")
    ap.add_argument("--workload", default="step2d", choices=["step2d", "sedov3d", "sw2d", "step2d_aeos", "cylinder3d"],
                    help="step2d = BASELINE configs[1] (the bench line); sedov3d = configs[2] (3-D radial "
                         "contrast box, --size cells per direction, default 200 -> 8.1M gridpoints); "
                         "sw2d = configs[4] (shallow-water circular dam break, default 1825^2 gridpoints)")
    ap.add_argument("--size", type=int, default=0)
    ap.add_argument("--develop-time", type=float, default=None,
                    help="simulated time the flow is developed to before anything is timed (default per workload: "
                         "2.0 for the Mach-3 step, mid-way through the reference's run to t = 4.0, "
                         "prm/benchmarks/euler-mach3-forward-facing-step.prm:30). The flow runs to that time on a "
                         "mesh --coarse-factor times coarser, is interpolated to the benchmark mesh and re-sharpened "
                         "by --develop updates there. 0: no coarse run, --develop updates from the initial state")
    ap.add_argument("--coarse-factor", type=int, default=4)
    ap.add_argument("--save-state", default=None,
                    help="N = 1: write the developed state (the one the warm-up starts from) to this .npz file")
    ap.add_argument("--load-state", default=None,
                    help="N = 1: start from a state written by --save-state of the same command instead of developing "
                         "it (profiling passes: the coarse run would dispatch the same kernels on another mesh)")
    ap.add_argument("--develop", type=int, default=None,
                    help="untimed forward-Euler updates on the benchmark mesh before the warm-up (default 600 behind "
                         "a coarse run: shocks interpolated from the coarse mesh steepen to the fine mesh's width "
                         "within ~100 updates; 900 with --develop-time 0, the start-up phase of the flow)")
    ap.add_argument("--perturbation", type=float, default=0.0,
                    help="multiplicative random perturbation of the state the timed updates start from "
                         "(initial_values.template.h:198-218): the limiter becomes active wherever it is applied")
    ap.add_argument("--perturbed-fraction", type=float, default=1.0,
                    help="apply --perturbation to this fraction of the rows only (those with the lowest indices: a "
                         "contiguous part of the mesh), to sweep the fraction of limited slices")
    ap.add_argument("--stagewise", action="store_true",
                    help="drive every forward-Euler update through prepare_state_vector/step/sadd calls "
                         "(one host synchronisation per update) instead of ryujin_hip_time_step")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-process code path (torch.distributed + RCCL communicator) even for one rank")
    ap.add_argument("--probe-launch", action="store_true",
                    help="(test hook) initialise torch.distributed, print the rank layout as the JSON line and exit "
                         "before anything touches a GPU")
    ap.add_argument("--reps", type=int, default=5,
                    help="timed passes of exactly --steps steps each (barrier + synchronise on both sides of "
                         "every pass); the JSON line reports the median pass and the min/max spread")
    ap.add_argument("--system-events", action="store_true",
                    help="create the events between the compute and the exchange stream with the system-scope "
                         "fence (ryujin_hip_params::system_scope_events)")
    ap.add_argument("--no-events-check", action="store_true",
                    help="N > 1: skip the bitwise comparison of a few updates run with device-scope and with "
                         "system-scope events before the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--binding", default="both", choices=["both", "device", "host-mirrored"],
                    help="`value` is always the device-resident figure (state vectors in HBM); `host-mirrored`/`both` "
                         "add the PCIe-inclusive ms/update of the adapter serving an UNMODIFIED ryujin TimeIntegrator "
                         "(host state vectors mirrored at every call) as the object `binding` (N = 1)")
    ap.add_argument("--watchdog", type=int, default=1500,
                    help="abort the process after this many seconds (a hung collective must not hang the box)")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    args = ap.parse_args()

    # Library banners (gloo, RCCL) go to fd 1; keep stdout clean for the ONE JSON line.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    if args.watchdog > 0:
        # a timer thread, not SIGALRM: a Python signal handler does not run while the main thread sits in a
        # blocking C call (hipStreamSynchronize behind a hung collective); ctypes calls release the GIL
        import threading

        def _abort():
            sys.stderr.write(f"bench.py: watchdog expired after {args.watchdog} s, aborting\n")
            sys.stderr.flush()
            os._exit(3)
        _wd = threading.Timer(args.watchdog, _abort)
        _wd.daemon = True
        _wd.start()

    # the host driver only supports dmabuf IPC: RCCL's cross-process buffers need this
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # libgomp reads these when it is first loaded (the mesh generator and the CPU baseline use OpenMP): one
    # thread per usable core (cgroup quota / affinity, not the host's CPU count), pinned. torch.distributed.run
    # already exports OMP_NUM_THREADS=1 for multi-rank launches.
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
    os.environ.setdefault("OMP_PROC_BIND", "close")
    os.environ.setdefault("OMP_PLACES", "cores")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called like the single-GPU line (`python bench.py --gpus N ...`): launch the N ranks ourselves, one
        # per GPU, exactly as the driver's torch.distributed.run command would; rank 0 of the child job prints
        # the ONE JSON line on the stdout we inherited
        os.dup2(saved_stdout, 1)
        cmd = self_launch_command(args.gpus, sys.argv[1:])
        sys.stderr.write("bench.py: launching " + " ".join(cmd) + "\n")
        sys.stderr.flush()
        os.execv(cmd[0], cmd)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch {args.gpus} ranks "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}) or none at all")
    n_gpus = max(1, world)

    dist = None
    use_dist = n_gpus > 1 or args.force_dist
    if use_dist:
        # torch must be imported BEFORE libryujin_hip.so so that one copy of the ROCm runtime is used
        import torch  # noqa: F401
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    if args.probe_launch:
        import torch
        tt = torch.tensor([rank], dtype=torch.int64)
        if dist is not None:
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        if rank == 0:
            os.dup2(saved_stdout, 1)
            print(json.dumps({"probe": True, "world": world, "n_gpus": n_gpus, "rank_sum": int(tt[0])}), flush=True)
        return

    import ctypes as C

    import numpy as np

    from ryujin_amd import HyperbolicModule, capi, offline
    from ryujin_amd.workloads import Ssprk33Stages, benchmark_workload, developed_state

    if dist is not None:
        # the in-tree libraries are (re)built on demand: let rank 0 do that alone, the others load after it
        if rank == 0:
            capi.load_synth()
            capi.load_hip()
        dist.barrier()

    # ---- workload: BASELINE.json configs[1] per GPU, lengthened channel for N GPUs (weak scaling); the recipes live
    # in ryujin_amd/workloads.py so that the parity tests build the state the timed updates start from the same way
    wl = benchmark_workload(args.workload, n_gpus, cells_per_unit=args.cells_per_unit, size=args.size,
                            coarse_factor=args.coarse_factor)
    equation, make_spec, U0_fn, dirichlet_fn = wl.equation, wl.make_spec, wl.U0_fn, wl.dirichlet_fn
    resolution, coarse_resolution = wl.resolution, wl.coarse_resolution
    default_develop_time, workload_name = wl.default_develop_time, wl.name
    rng = np.random.default_rng(42 + rank)
    n = resolution
    spec = make_spec(resolution, n_gpus, rank)
    off = offline.SyntheticOffline(spec)
    dirichlet = dirichlet_fn(off.b_positions) if (dirichlet_fn is not None and off.n_bdry) else None
    develop_time = default_develop_time if args.develop_time is None else args.develop_time
    n_develop = args.develop if args.develop is not None else (600 if develop_time > 0.0 else 900)

    lib = capi.load_hip()
    comm = None
    device = local_rank
    if use_dist:
        import torch
        n_visible = torch.cuda.device_count()  # a launcher may expose one GPU per rank
        if n_visible > 0:
            device = local_rank % n_visible
        if n_gpus > 1 and 1 < n_visible < n_gpus:
            raise SystemExit(f"--gpus {n_gpus} but only {n_visible} devices are visible: RCCL needs one GPU per rank")
        uid = C.create_string_buffer(capi.UNIQUE_ID_BYTES)
        if rank == 0:
            assert lib.ryujin_hip_comm_unique_id(uid) == 0, lib.ryujin_hip_last_error()
        t = torch.frombuffer(bytearray(uid.raw), dtype=torch.uint8).clone()
        dist.broadcast(t, src=0)
        uid = C.create_string_buffer(bytes(t.tolist()), capi.UNIQUE_ID_BYTES)
        comm = C.c_void_p()
        rc = lib.ryujin_hip_comm_init(C.byref(comm), uid, rank, world, device)
        assert rc == 0, lib.ryujin_hip_last_error()

    def make_module(system_events: bool):
        p = capi.Params()
        lib.ryujin_hip_default_params(C.byref(p), equation, off.dim)
        p.system_scope_events = 1 if system_events else 0
        mod = HyperbolicModule(off, p, backend="hip", comm=comm, device=device)
        mod.cfl = 0.9
        return mod

    # ---- the state the timed updates start from. The reference's benchmark runs to t = 4.0; its cost does not
    # depend on the state, ours does (the limiter sweeps skip what an unlimited slice does not need), so the state
    # has to look like the benchmark: the flow is run to develop_time on a coarse mesh of the same domain (every
    # rank on its own, the whole domain, no communicator), interpolated to the benchmark mesh, and re-sharpened
    # by n_develop updates there.
    t_start = 0.0
    coarse = None
    if args.load_state:
        assert n_gpus == 1, "--load-state: one GPU"
        z = np.load(args.load_state)
        U0, t_start = z["U"], float(z["t"])
        assert U0.shape[0] == off.n_relevant, "the state file belongs to another mesh"
        coarse = {"loaded_from": os.path.basename(args.load_state)}
        n_develop = 0
    elif develop_time > 0.0:
        t0 = time.perf_counter()
        U0, t_start, coarse = developed_state(wl, off, develop_time, device)
        coarse["seconds"] = round(time.perf_counter() - t0, 2)
    else:
        U0 = U0_fn(off.positions)

    m = make_module(args.system_events)
    drv = Ssprk33Stages(m, U0, dirichlet)
    drv.t = t_start
    ctx = m._ctx

    def barrier():
        lib.ryujin_hip_synchronize(ctx)
        if dist is not None:
            dist.barrier()

    for _ in range(n_develop):  # untimed: let the flow develop / re-sharpen
        drv.update()
    if args.perturbation != 0.0:
        # the pessimistic variant: a random perturbation of the developed state makes the limiter work in every
        # perturbed row (--perturbed-fraction: a contiguous part of the mesh, to sweep the limited fraction)
        while drv.stage != 0:
            drv.update()
        U_p = drv.U.download()
        n_p = int(round(args.perturbed_fraction * U_p.shape[0]))
        U_p[:n_p] *= 1.0 + args.perturbation * rng.uniform(-1.0, 1.0, size=U_p[:n_p].shape)
        t_p = drv.t
        drv = Ssprk33Stages(m, U_p, dirichlet)
        drv.t = t_p
        for _ in range(6):
            drv.update()
    U_developed = drv.U.download() if (n_gpus == 1 and (not args.no_cpu_baseline or args.save_state)) else None
    if args.save_state and n_gpus == 1:
        while drv.stage != 0:
            drv.update()
        U_developed = drv.U.download()
        np.savez(args.save_state, U=U_developed, t=np.float64(drv.t))

    # ---- N > 1: device-scope against system-scope events, before anything is timed. The events that tie the
    # compute and the exchange stream carry no system-scope fence by default; whether that
    # is enough when a REMOTE GPU wrote the ghost data is decided here by the hardware: the same six updates from
    # the same developed state with either kind of event must agree bit for bit on every rank. If they do not,
    # the run continues on system-scope events and says so in the JSON line.
    events = {"kind": "system" if args.system_events else "device", "check": None}
    if n_gpus > 1 and not args.no_events_check and not args.system_events:
        import torch
        while drv.stage != 0:
            drv.update()
        U_start = drv.U.download()
        m_sys = make_module(True)
        drv_sys = Ssprk33Stages(m_sys, U_start, dirichlet)
        drv_dev = Ssprk33Stages(m, U_start, dirichlet)
        for _ in range(2):
            drv_dev.rk_step()
            drv_sys.rk_step()
        same = bool(np.array_equal(drv_dev.U.download(), drv_sys.U.download()))
        flag = torch.tensor([0 if same else 1], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag[0]) == 0:
            events["check"] = "2 SSPRK33 steps with device-scope and with system-scope events: bitwise equal on every rank"
            m_sys.close()
        else:
            events = {"kind": "system", "check": "device-scope events gave DIFFERENT results on some rank "
                                                 "(stale ghost data): timed on system-scope events"}
            m, ctx = m_sys, m_sys._ctx
            t_keep = drv.t
            drv = Ssprk33Stages(m, U_start, dirichlet)
            drv.t = t_keep
    for _ in range(args.warmup):
        drv.update()

    tmp = (C.c_double * 8)()
    n_upd = C.c_uint(0)

    def run_steps(k):
        # whole SSPRK33 steps go through the device-resident RK driver; a remainder (k % 3) stage-wise
        n_done = 0
        while n_done < k:
            if drv.stage == 0 and k - n_done >= 3 and not args.stagewise:
                drv.rk_step()
                n_done += 3
            else:
                drv.update()
                n_done += 1

    # ---- pass 1, the reported value: --reps passes of exactly K steps each, no per-sweep instrumentation (the
    # hipEvent pairs around every sweep cost a few per cent: each record is a barrier packet between two kernels).
    # Every pass is bracketed by barrier + synchronise on both sides; the MEDIAN pass is the reported one
    # (SURVEY.md section 8d), min and max travel with it.
    lib.ryujin_hip_set_timers(ctx, 0)
    t_at_start = drv.t
    walls, evs = [], []
    for _ in range(max(1, args.reps)):
        while drv.stage != 0:  # every pass starts at an SSPRK33 step boundary
            drv.update()
        barrier()
        t0 = time.perf_counter()
        lib.ryujin_hip_event_record(ctx, 0)
        run_steps(args.steps)
        lib.ryujin_hip_event_record(ctx, 1)
        barrier()
        w = time.perf_counter() - t0
        e = C.c_double()
        lib.ryujin_hip_event_elapsed_ms(ctx, C.byref(e))
        if dist is not None:
            import torch
            tt = torch.tensor([w], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)  # MAX over ranks, per pass
            w = float(tt[0])
        walls.append(w)
        evs.append(e.value)
    order = sorted(range(len(walls)), key=lambda q: walls[q])
    median_pass = order[(len(order) - 1) // 2]
    wall = walls[median_pass]
    ev_ms = C.c_double(evs[median_pass])

    # ---- pass 2, the roofline breakdown: the same K steps again with hipEvent pairs around every sweep
    while drv.stage != 0:
        drv.update()
    # (three times, per sweep the median of the three means: a single stalled launch -- 0.5 ms instead of 0.24 ms
    # for step 2 in one of the round-4 lines -- would otherwise name the wrong dominant sweep)
    lib.ryujin_hip_set_timers(ctx, 1)
    sweep_passes, ev_passes = [], []
    xinfo0 = m.exchange_info()
    for _ in range(3):
        lib.ryujin_hip_get_timers_accum(ctx, tmp, C.byref(n_upd), 1)  # reset the accumulators
        barrier()
        lib.ryujin_hip_event_record(ctx, 0)
        run_steps(args.steps)
        lib.ryujin_hip_event_record(ctx, 1)
        barrier()
        e = C.c_double()
        lib.ryujin_hip_event_elapsed_ms(ctx, C.byref(e))
        lib.ryujin_hip_get_timers_accum(ctx, tmp, C.byref(n_upd), 0)
        assert n_upd.value == args.steps, (n_upd.value, args.steps)
        sweep_passes.append(np.array(tmp[:]))
        ev_passes.append(e.value)
    xinfo1 = m.exchange_info()
    sweep_ms = np.median(np.array(sweep_passes), axis=0)
    ev_ms_instrumented = C.c_double(float(np.median(ev_passes)))

    binding = None
    if n_gpus == 1 and args.binding != "device":
        while drv.stage != 0:
            drv.update()
        U_now = drv.U.download()
        n_rk = max(2, min(10, args.steps // 3))
        binding = {"what": ("ms per update as a ryujin TimeLoop sees it, by how the caller is bound "
                            "(contrib/hyperbolic_module_hip.h, INTEGRATION.md section 2a)"),
                   "c_abi_device_resident": wall / args.steps * 1e3,
                   "patched_caller_time_step": None,  # filled below: time_step() + one download of U per RK step
                   "unmodified_caller_host_mirrored": host_mirrored_ssprk33(m, U_now, dirichlet, drv.t, n_rk, True, True),
                   "unmodified_caller_host_mirrored_U_only": host_mirrored_ssprk33(m, U_now, dirichlet, drv.t, n_rk, True, False),
                   "unmodified_caller_pageable": host_mirrored_ssprk33(m, U_now, dirichlet, drv.t, max(2, n_rk // 2), False, True)}
        # the patched caller in full mirroring: one upload + time_step + one download of the owned rows per RK step
        host_U = np.array(U_now, copy=True)
        lib.ryujin_hip_host_register(ctx, host_U.ctypes.data, host_U.nbytes)
        T3 = [m.new_state_vector() for _ in range(3)]
        sv = m.new_state_vector()
        w0 = time.perf_counter()
        for _ in range(n_rk):
            sv.upload(host_U)
            m.time_step("ssprk 33", sv, T3, dirichlet)
            lib.ryujin_hip_state_download_owned(ctx, sv.handle, capi.as_ptr(host_U, capi.c_double_p))
        binding["patched_caller_time_step"] = {"ms_per_update": (time.perf_counter() - w0) / (3 * n_rk) * 1e3,
                                               "mirroring": "full: U up and down once per SSPRK33 step",
                                               "device_resident": "= c_abi_device_resident (no transfers)"}
        lib.ryujin_hip_host_unregister(ctx, host_U.ctypes.data)
        for x in T3 + [sv]:
            x.free()

    n_q_local = off.n_owned
    if dist is not None:
        import torch
        nn = torch.tensor([n_q_local], dtype=torch.int64)
        dist.all_reduce(nn, op=dist.ReduceOp.SUM)
        n_q_total = int(nn[0])
    else:
        n_q_total = n_q_local

    # what RCCL and the library themselves report about the multi-rank run (rank 0's view + neighbour counts
    # of all ranks): "RCCL saw N ranks" can be read off the JSON line
    rccl = None
    if comm is not None:
        v = [C.c_int(-1) for _ in range(5)]
        lib.ryujin_hip_comm_info(comm, *[C.byref(x) for x in v])
        info = xinfo1
        nb = [len(info["neighbours"])]
        if dist is not None:
            gathered = [None] * world
            dist.all_gather_object(gathered, nb[0])
            nb = gathered
        rccl = {"ranks": v[3].value, "rank": v[2].value, "device": v[4].value,
                "neighbours_of_rank0": info["neighbours"], "n_neighbours_per_rank": nb,
                # counted by the library over the K steps of the instrumented pass (5 ghost exchanges per update:
                # U, alpha, r, l_ij, l'_ij; 2 all-reduces per SSPRK33 step: tau_max of the first stage, the flags)
                "exchanges_per_update": (xinfo1["n_exchanges"] - xinfo0["n_exchanges"]) / (3 * args.steps),
                "allreduces_per_update": (xinfo1["n_allreduces"] - xinfo0["n_allreduces"]) / (3 * args.steps)}

    if rank != 0:
        return

    k = m.k
    rs = off.row_starts
    S = float(rs[off.n_owned]) / off.n_owned
    alg = algorithmic_bytes(off.dim, k, S, n_prec=m.n_prec, n_bounds=m.n_bounds)
    if equation == capi.EQ_SHALLOW_WATER:  # + bathymetry (8 B) and m_ij (8S) in step 4, SURVEY 8d
        alg["4 low_order"] += 8 + 8 * S
    b_alg = sum(alg.values())
    # per-sweep mean kernel durations of rank 0 (hipEvent pairs on the library's stream); sweep 1
    # (prepare_state_vector) is not bracketed separately: it is the remainder of the event time
    per_sweep = {name: sweep_ms[i + 1] / args.steps for i, name in enumerate(list(alg)[1:])}
    per_sweep["1 prepare_state_vector"] = max(0.0, ev_ms_instrumented.value / args.steps - sum(per_sweep.values()))
    if sweep_ms[0] > 0.0:
        # step 2 runs as two kernels: the streaming indicator sweep (reads the stencil once: the sweep's
        # algorithmic reads) and the compute-bound Riemann sweep (its algorithmic share: the d_ij stores)
        t_alpha = sweep_ms[0] / args.steps
        t_both = per_sweep.pop("2 dij_alpha")
        b_both = alg.pop("2 dij_alpha")
        per_sweep["2a alpha (k_alpha)"] = t_alpha
        per_sweep["2b dij (k_dij)"] = t_both - t_alpha
        alg["2a alpha (k_alpha)"] = b_both - 8 * S
        alg["2b dij (k_dij)"] = 8 * S
    dom = max((n for n in alg if n != "1 prepare_state_vector"), key=lambda n: per_sweep[n])
    upd_gbs = b_alg * n_q_local / (ev_ms.value / args.steps * 1e-3) / 1e9
    # `alg` is the REFERENCE's sweep structure. The kernels of an update without stage vectors no longer
    # touch all of it: step 4 does not write P_ij, step 5 forms it from d_ij, m_ij and per-node vectors instead of
    # reading a first part, and -- while few slices are limited -- does not store it either.
    # `own` counts what THIS implementation's kernel touches once (same convention, gathers free): the numerator of
    # `roofline`, a true lower bound on its traffic (frac <= 1). The reference-structure figure travels along.
    limiter = m.limiter_statistics()
    own = dict(alg)
    if equation in (capi.EQ_EULER, capi.EQ_EULER_AEOS):
        own["4 low_order"] -= 8 * k * S
        own["5 pij_lij"] += -8 * k * S + 8 * S + 8 * k + 8 * k  # - first part; + m_ij, F_i read, V_i written
        # P_ij is written in the slices steps 6/7 read it in (counted by the library over the instrumented pass)
        own["5 pij_lij"] -= (1.0 - limiter["pij_stored_slice_fraction"]) * 8 * k * S
    # the tile map (1-D / 2-D): steps 3, 5, 6, 7 read a 16-byte descriptor per regular 64-entry tile instead of its
    # column indices / transposed positions (4 S bytes per row): not compulsory bytes of these kernels any more
    layout = m.layout_info()
    if layout["n_regular_tiles"]:
        saved = 4 * S * layout["regular_tile_fraction"] - 16.0 * layout["n_regular_tiles"] / off.n_owned
        for name in ("3 dij_diag_tau", "5 pij_lij", "6 high_order_next_lij", "7 high_order"):
            if name in own:
                own[name] -= saved
    dom_gbs = own[dom] * n_q_local / (per_sweep[dom] * 1e-3) / 1e9
    ref_gbs = alg[dom] * n_q_local / (per_sweep[dom] * 1e-3) / 1e9

    # the committed PMC passes were taken on the default problem of each workload: do not attach them to a run
    # of another size or with a perturbed state
    default_problem = (args.cells_per_unit == 995 and args.size == 0 and args.perturbation == 0.0 and n_gpus == 1 and
                       args.develop_time is None and args.develop is None and args.coarse_factor == 4)
    traffic, traffic_file, traffic_note = (pmc_traffic_bytes(dom, args.workload) if default_problem
                                           else (None, None, "not the default problem of the workload"))
    out = {
        "metric": "MDoF-updates/s per Euler forward step; achieved HBM GB/s vs roofline",
        "value": k * n_q_total * args.steps / wall / 1e6,
        "unit": "MDoF-updates/s",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3,
        "reps": len(walls), "ms_per_step_min": min(walls) / args.steps * 1e3,
        "ms_per_step_max": max(walls) / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload_name,
                   "gridpoints_per_gpu": n_q_local, "gridpoints_total": n_q_total,
                   "dofs_total": k * n_q_total, "nnz_per_row": round(S, 3),
                   "cells_per_unit": (args.cells_per_unit if args.workload.startswith("step2d") else n), "partition": f"x-slabs x{n_gpus}, equal gridpoint counts",
                   "cfl": 0.9, "limiter_iterations": 2,
                   # how the state the timed updates start from was made: run to develop_time on the coarse mesh,
                   # interpolated, develop_updates updates on this mesh (develop_time 0: from the initial state)
                   "develop_time": develop_time, "coarse_run": coarse, "develop_updates": n_develop,
                   "simulated_time_at_start": t_at_start, "simulated_time_at_end": drv.t, "perturbation": args.perturbation,
                   "perturbed_fraction": args.perturbed_fraction if args.perturbation != 0.0 else None},
        "mq_per_s": n_q_total * args.steps / wall / 1e6,
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": dom_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": dom_gbs / HBM_PEAK_GBS,
                     "traffic": traffic,
                     "traffic_source": (f"{traffic_file} (separate rocprofv3 --pmc passes of this command, bytes per "
                                        f"launch; kernel sources {source_fingerprint()})") if traffic else traffic_note,
                     "traffic_calibration": ("(2 FETCH_SIZE + WRITE_SIZE) x 1024; factor 2.000 measured for streaming "
                                             "reads of 4/8/16 B per lane, 1.98-1.99 for stencil-order gathers of "
                                             "32/64-byte records, WRITE_SIZE 1.000 "
                                             "(profiles/r03h_counter_calibration.md)") if traffic else None,
                     "traffic_frac": (traffic / (per_sweep[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "valu": pmc_valu_issue(dom, args.workload, per_sweep[dom]) if traffic else None,
                     "algorithmic_bytes_per_gridpoint": own[dom],
                     "mean_launch_ms": per_sweep[dom],
                     "reference_structure": {
                         "algorithmic_bytes_per_gridpoint": alg[dom], "achieved": ref_gbs,
                         "frac": ref_gbs / HBM_PEAK_GBS,
                         "note": ("the reference's bytes for this sweep (SURVEY 8d) over this kernel's time, as in "
                                  "rounds 1-2; above 1 when the kernel no longer moves them (P_ij neither read nor "
                                  "stored): not a bandwidth, `traffic` is")}},
        "limiter": limiter,  # fraction of limited slices and whether P_ij was stored
        "layout": layout,    # the tile map: tiles served by a descriptor instead of the index arrays
        "roofline_update": {"bound": "hbm", "achieved": upd_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": upd_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_gridpoint": b_alg,
                            "kind": ("effective: the reference's algorithmic bytes (SURVEY 8d) over this "
                                     "implementation's time -- bytes the kernels no longer move; the measured "
                                     "traffic of an update is in DESIGN.md section 0"),
                            "device_ms_per_update": ev_ms.value / args.steps,
                            "device_ms_per_update_instrumented": ev_ms_instrumented.value / args.steps},
        "sweep_ms": {n: round(v, 4) for n, v in sorted(per_sweep.items())},
        "n_warnings": m.n_warnings(),
        "events": events,
    }
    if rccl is not None:
        out["rccl"] = rccl
    if binding is not None:
        out["binding"] = binding
    if not args.no_cpu_baseline and n_gpus == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(spec, U_developed, dirichlet, args.cpu_budget, equation)
        except Exception as e:  # the baseline must never take the GPU number down with it
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    sys.stdout.flush()
    C.CDLL(None).fflush(None)  # C stdio of librccl (version banner) is block-buffered on a pipe: drain it to fd 2
    os.dup2(saved_stdout, 1)
    print(json.dumps(out), flush=True)
    # anything a library prints at exit must not follow the JSON line on stdout
    sys.stdout.flush()
    os.dup2(2, 1)


if __name__ == "__main__":
    main()

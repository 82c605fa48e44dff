"""bench.py's accounting, checked without a GPU: the algorithmic bytes per gridpoint-update are the ones
SURVEY.md section 8d derives (3164 / 10020 / 2844 B), and the committed bench lines in profiles/ carry every
field of the driver's contract."""
import glob
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_match_the_survey():
    import bench
    assert sum(bench.algorithmic_bytes(2, 4, 9).values()) == 48 + 316 + 224 + 700 + 860 + 556 + 460 == 3164
    assert sum(bench.algorithmic_bytes(3, 5, 27).values()) == 56 + 1044 + 656 + 2236 + 2820 + 1724 + 1484 == 10020
    sw = bench.algorithmic_bytes(2, 3, 9, n_prec=2, n_bounds=5)
    sw["4 low_order"] += 8 + 8 * 9        # bathymetry and m_ij in step 4
    assert sum(sw.values()) == 40 + 308 + 224 + 700 + 716 + 484 + 372 == 2844
    assert list(bench.algorithmic_bytes(2, 4, 9).values()) == [48, 316, 224, 700, 860, 556, 460]


def _committed_bench_lines():
    """every bench line committed from round 2 (second half) on; earlier ones predate the contract fields"""
    import re
    paths = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench*.json"))):
        tag = re.match(r"r(\d+)([a-z]?)_", os.path.basename(path))
        if tag and (int(tag.group(1)), tag.group(2)) >= (2, "e"):
            paths.append(path)
    return paths


def test_there_are_committed_bench_lines_to_check():
    assert len(_committed_bench_lines()) >= 1


@pytest.mark.parametrize("path", _committed_bench_lines())
def test_committed_bench_lines_follow_the_contract(path):
    lines = [ln for ln in open(path).read().splitlines() if ln.strip()]
    assert len(lines) == 1                                     # ONE JSON line
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d, key
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    if "reference_structure" in r:   # from r03x on: the numerator is what THIS implementation's kernel touches once
        assert 0.0 < r["frac"] <= 1.0 and r["reference_structure"]["frac"] >= r["frac"]
        assert r["traffic_frac"] is None or r["frac"] <= r["traffic_frac"] * 1.02 <= 1.02
        assert d["limiter"]["pij_stored"] in (True, False, "everywhere", "per slice", "per tile", "all", "tiles")
        # (bool: round 3; "all" / "tiles": the tile-storage experiment of round 4, profiles/r04c_*; "per tile": the
        # per-tile storage of round 5, with the stored / read / formed-by-step-6 fractions of the tiles next to it)
        if d["limiter"]["pij_stored"] == "per tile":
            lm = d["limiter"]
            assert 0.0 <= lm["tiles_formed_by_step6_fraction"] <= lm["tiles_read_fraction"] <= 1.0
            assert lm["tiles_read_fraction"] <= lm["tiles_stored_fraction"] + lm["tiles_formed_by_step6_fraction"] + 1e-12
        assert 0.0 <= d["limiter"]["limited_slice_fraction"] <= 1.0
    # value is consistent with the time per step and the size of the job
    dofs = d["config"]["dofs_total"]
    assert abs(d["value"] - dofs / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * d["value"]
    if "reps" in d:   # round 3 on: median of `reps` passes of exactly `steps` steps, spread reported
        assert d["reps"] >= 1 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    if "cpu_baseline" in d:
        c = d["cpu_baseline"]
        assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == d["unit"] and c["sample"]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` invoked exactly like the single-GPU line (no launcher around it) starts N ranks
    itself -- torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 -- and rank 0 prints the ONE
    JSON line. Checked on CPU with the launch probe (gloo, 2 ranks; nothing touches a GPU)."""
    import subprocess
    import sys

    import bench
    cmd = bench.self_launch_command(4, ["--gpus", "4", "--steps", "3"])
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--probe-launch"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    assert json.loads(lines[0]) == {"probe": True, "world": 2, "n_gpus": 2, "rank_sum": 1}


@pytest.mark.parametrize("mesh", ["step2d", "cylinder3d", "box3d"])
def test_bench_interpolates_the_coarse_state_to_the_benchmark_mesh(mesh):
    """bench.py develops the flow on a coarse mesh of the same domain and interpolates it to the benchmark mesh
    (interpolate_from_lattice): a function that is linear in the coordinates must arrive exactly, also next to the
    cut-outs (the step; the staircase cylinder, whose coarse staircase differs from the fine one), and no fine
    point may pick up a lattice node the coarse mesh does not have."""
    import numpy as np

    import bench
    from ryujin_amd import offline
    if mesh == "step2d":
        spec_c, spec_f = offline.mach3_step_2d(10), offline.mach3_step_2d(35)
    elif mesh == "cylinder3d":
        spec_c, spec_f = offline.cylinder_channel_3d(6, length_units=1.25), offline.cylinder_channel_3d(20, length_units=1.25)
    else:
        spec_c, spec_f = offline.box_3d(4), offline.box_3d(10)
    off_c, off_f = offline.SyntheticOffline(spec_c), offline.SyntheticOffline(spec_f)
    coeff = np.array([[1.0, -2.0, 0.5], [0.25, 3.0, -1.0]])[:, : off_c.dim]

    def f(x):
        return 2.0 + x @ coeff.T
    from ryujin_amd.workloads import interpolate_from_lattice
    U_f = interpolate_from_lattice(spec_c, off_c.positions[: off_c.n_owned], f(off_c.positions[: off_c.n_owned]),
                                         off_f.positions)
    assert U_f.shape == (off_f.n_relevant, 2) and np.isfinite(U_f).all()
    if mesh == "cylinder3d":
        # filled lattice nodes (inside the coarse staircase) carry averages of their neighbours: exact only away
        # from the cylinder; next to it the interpolant stays within the range of the surrounding values
        d = np.hypot(off_f.positions[:, 0] - 0.6, off_f.positions[:, 1])
        far = d > 0.25 + 2.0 / 6.0
        assert np.abs(U_f[far] - f(off_f.positions[far])).max() < 1e-12
        assert np.abs(U_f - f(off_f.positions)).max() < 1.5  # |grad f| h_c-sized
    else:
        assert np.abs(U_f - f(off_f.positions)).max() < 1e-12


def test_counter_profiles_of_other_kernel_sources_are_refused(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 --pmc pass; bench.py attaches it only if the pass was taken
    with the kernel sources of this tree (scripts/profile_round.sh stamps the summary with
    bench.source_fingerprint()): a stale or unstamped profile gives traffic = None and says why."""
    import bench
    fp = bench.source_fingerprint()
    assert len(fp) == 16 and fp == bench.source_fingerprint()
    (tmp_path / "profiles").mkdir()
    row = "| k_lij_stage0<Euler<2>, 1, false> | 39 | 1 | 2 | 3 | 2256.6 |\n"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "source_fingerprint", lambda: fp)
    prof = tmp_path / "profiles" / "r99_pmc.md"
    prof.write_text("# pmc\n" + row)                                    # no stamp at all (rounds 1-3)
    val, src, note = bench.pmc_traffic_bytes("5 pij_lij", "step2d")
    assert val is None and "REFUSED as stale" in note and src.endswith("r99_pmc.md")
    prof.write_text("# pmc\n" + row + "\nkernel sources: 0123456789abcdef\n")
    val, _, note = bench.pmc_traffic_bytes("5 pij_lij", "step2d")
    assert val is None and "0123456789abcdef" in note and fp in note
    prof.write_text("# pmc\n" + row + "\nkernel sources: %s\n" % fp)
    val, _, note = bench.pmc_traffic_bytes("5 pij_lij", "step2d")
    assert note is None and abs(val - 2256.6e6) < 1.0
    assert bench.pmc_traffic_bytes("5 pij_lij", "sedov3d")[2] == "no committed PMC pass of this workload"


def test_comments_do_not_change_the_fingerprint_of_the_kernel_sources(tmp_path, monkeypatch):
    import bench
    csrc = tmp_path / "ryujin_amd" / "csrc"
    csrc.mkdir(parents=True)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (csrc / "a.hpp").write_text("int f(int x) { return x + 1; } // plus one\n")
    one = bench.source_fingerprint()
    (csrc / "a.hpp").write_text("/* the successor */\nint f(int x)\n{\n  return x + 1;\n}\n")
    assert bench.source_fingerprint() == one
    (csrc / "a.hpp").write_text("int f(int x) { return x + 2; }\n")
    assert bench.source_fingerprint() != one

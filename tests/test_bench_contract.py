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
        assert isinstance(d["limiter"]["pij_stored"], bool) and 0.0 <= d["limiter"]["limited_slice_fraction"] <= 1.0
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

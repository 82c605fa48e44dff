"""The chain codes of the tile map on the CPU (ryujin_amd/csrc/host_layout.hpp: TileDesc::chain, chain_loads): the
relation the chained gathers of step 5 rely on, checked lane by lane on synthetic meshes, and how many tiles of a
lattice-numbered mesh they reach. Host logic only: no GPU, no HIP runtime."""
import json
import os
import subprocess

import pytest

from ryujin_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_layout_chains")


@pytest.fixture(scope="module")
def checker():
    synth = _build.build_synth()
    src = os.path.join(ROOT, "tests", "cpp", "host_layout_chains.cc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "ryujin_amd", "csrc"), src, synth,
                    "-Wl,-rpath," + os.path.dirname(synth), "-o", BIN], check=True)
    return BIN


def _run(checker, *args):
    out = subprocess.run([checker, *map(str, args)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    return json.loads(out.stdout)


@pytest.mark.parametrize("mesh, least", [((2, 300, 40, 1), 0.70), ((2, 70, 70, 1), 0.55), ((3, 120, 12, 12), 0.60),
                                         ((3, 20, 20, 20), 0.35), ((1, 500, 1, 1), 0.40)])
def test_chain_codes_hold_lane_by_lane(checker, mesh, least):
    """No lane outside a tile's mask violates the tile's relation; the end-lane-only flag says what the mask says; and on
    a lattice most off-diagonal tiles are chained: 6 of 8 columns in 2-D, 18 of 26 in 3-D where the lattice rows are
    long against a slice, fewer where every slice holds a row end or two (20^3: 21 nodes per row)."""
    r = _run(checker, *mesh)
    assert r["violations"] == 0, r
    assert r["n_chained_tiles"] >= least * r["off_diagonal_tiles"], r
    assert r["n_end_lane_tiles"] <= r["n_chained_tiles"]
    assert r["n_chained_entries"] <= 63 * r["n_chained_tiles"]

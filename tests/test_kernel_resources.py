"""Occupancy contract of the hot kernels (no GPU: hipcc's -Rpass-analysis=kernel-resource-usage on the gfx950 code
object, scripts/kernel_resources.py). The sweeps run at the waves per SIMD their registers allow, and several of this
round's experiments were decided by a cliff -- a spill reload inside a column loop drains the prefetch (scratch loads
count in vmcnt), eight more registers cost the 2-D step 6 two of its four waves: this pins what the final tree has, so
that a change which falls off one of the cliffs fails here and not on the next bench line."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel (as kernel_resources.py prints it) -> (max VGPRs, max scratch bytes per lane, min waves per SIMD)
CONTRACT = {
    "k_lij_stage0<Euler<2>, 1, false, true>": (168, 0, 3),    # step 5, C2 (P_ij per tile, chained gathers)
    "k_lij_stage0<Euler<3>, 1, false, false>": (168, 0, 3),   # step 5, C4 (chain masks; 16 B of scratch before round 6)
    "k_lij_stage0<Euler<3>, 1, true, false>": (168, 0, 3),    # step 5, C3 (P_ij per slice)
    "k_dij_alpha_records<Euler<2>, false>": (128, 0, 4),      # step 2, C2
    "k_dij_alpha_records<Euler<3>, false>": (168, 0, 3),      # step 2, 3-D (node records)
    "k_low_order<2, false, false, false>": (168, 28, 3),      # step 4, C2 (its 20 B of scratch sit outside the loop)
    "k_low_order<3, false, false, false>": (256, 0, 2),       # step 4, 3-D
    "k_high_order_next_cached<Euler<2>, 9, 9, false, 0>": (128, 0, 4),  # step 6, C2
    "k_high_order_last_cached<Euler<2>, 9, 3>": (84, 0, 6),   # step 7, C2
    "k_dij_diag_unrolled<9>": (32, 0, 8),                     # step 3, C2
    "k_pij_lij<ShallowWater<2>, false, false>": (128, 0, 4),  # step 5, C5
}


@pytest.fixture(scope="module")
def table():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "kernel_resources.py"), "k_lij_stage0<Euler",
                          "k_dij_alpha_records<Euler", "k_low_order<", "k_high_order_next_cached<Euler<2>",
                          "k_high_order_last_cached<Euler<2>", "k_dij_diag_unrolled", "k_pij_lij<ShallowWater<2>"],
                         capture_output=True, text=True, timeout=1200)
    rows = {}
    for line in out.stdout.splitlines():
        m = re.match(r"(.+?)\s+vgpr\s+(\d+)\s+agpr\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)", line)
        if m:
            rows[m.group(1).strip()] = tuple(int(m.group(k)) for k in (2, 4, 5))
    assert rows, out.stdout[-2000:] + out.stderr[-2000:]
    return rows


@pytest.mark.parametrize("kernel", sorted(CONTRACT))
def test_hot_kernel_keeps_its_registers_scratch_and_occupancy(table, kernel):
    assert kernel in table, sorted(table)
    vgpr, scratch, occupancy = table[kernel]
    max_vgpr, max_scratch, min_occupancy = CONTRACT[kernel]
    assert vgpr <= max_vgpr and scratch <= max_scratch and occupancy >= min_occupancy, (kernel, table[kernel])

"""The C++ host shim (ryujin_amd/csrc/hyperbolic_module_shim.hpp) that a ryujin maintainer binds:
it must compile against include/ryujin_hip.h alone and link with the shared libraries (CPU check),
and -- on the GPU -- reproduce the time axis and mean density of the reference's
check-mass-conservation_01 baseline when driven exactly like TimeIntegrator::step_ssprk_33."""
import os
import subprocess

import numpy as np
import pytest

from ryujin_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "shim_ssprk33")


def _build_exe():
    _build.build_synth()
    cmd = ["g++", "-O1", "-std=c++17", "-I" + _build.INCLUDE, "-I" + _build.CSRC,
           os.path.join(ROOT, "tests", "cpp", "shim_ssprk33.cc"), "-L" + _build.LIBDIR,
           "-lryujin_hip", "-lryujin_synth", "-Wl,-rpath," + _build.LIBDIR, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link,/opt/rocm/lib", "-o", EXE]
    subprocess.run(cmd, check=True, capture_output=True)


def test_shim_compiles_and_links():
    _build_exe()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_shim_reproduces_mass_conservation_golden(golden_dir):
    from test_oracle_golden_integration import golden_mass_conservation
    # (re)build unless the binary is newer than every header it was compiled against: a stale one would hand the
    # library a ryujin_hip_params of an older layout
    headers = [os.path.join(_build.INCLUDE, f) for f in os.listdir(_build.INCLUDE)] + \
        [os.path.join(_build.CSRC, "hyperbolic_module_shim.hpp"), os.path.join(ROOT, "tests", "cpp", "shim_ssprk33.cc")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(h) for h in headers):
        _build_exe()
    gold = golden_mass_conservation(golden_dir)
    outs = []
    for device_resident in ("0", "1"):  # stage-wise like step_ssprk_33 / the library's time_step
        out = subprocess.run([EXE, "6", device_resident], check=True, capture_output=True, text=True,
                             timeout=300).stdout
        got = np.array([[float(x) for x in line.split()] for line in out.strip().splitlines()])
        np.testing.assert_allclose(got[:, 0], gold[1:7, 0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(got[:, 1], gold[1:7, 1], rtol=0, atol=1e-12)
        outs.append(out)
    assert outs[0] == outs[1]

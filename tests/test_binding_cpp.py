"""The ryujin-side binding (contrib/): its deal.II-free half -- OfflineData -> ryujin_hip_offline in memory,
ParameterAcceptor values -> ryujin_hip_params, the host-StateVector -> device-handle cache -- is compiled with g++
and run against a mock that serves the generator's arrays through the accessor names of the reference
(tests/cpp/binding_fill.cc): the struct the adapter hands to ryujin_hip_create() must be the generator's own, bit
for bit, on every rank of a slab partition (2-D step with coupling boundary pairs, 3-D). The deal.II half
(contrib/hyperbolic_module_hip.h) cannot be compiled here; its patch is checked for applicability against the
two files of the reference it touches when /root/reference is present (this container only)."""
import os
import shutil
import subprocess

import pytest

from ryujin_amd import _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "binding_fill")


def _build_exe():
    _build.build_synth()
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + _build.INCLUDE, "-I" + os.path.join(ROOT, "contrib"),
           os.path.join(ROOT, "tests", "cpp", "binding_fill.cc"), "-L" + _build.LIBDIR, "-lryujin_hip",
           "-lryujin_synth", "-Wl,-rpath," + _build.LIBDIR, "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,-rpath-link,/opt/rocm/lib", "-o", EXE]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]


@pytest.mark.parametrize("dim,n_ranks", [(2, 1), (2, 3), (3, 4)])
def test_fill_from_accessors_reproduces_the_generator(dim, n_ranks):
    _build_exe()
    res = subprocess.run([EXE, str(dim), str(n_ranks)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = res.stdout.strip().splitlines()
    assert len(lines) == n_ranks + 1 and all(ln.endswith("identical") for ln in lines), res.stdout


def test_exporter_and_adapter_share_the_fill_loops():
    """one statement of the OfflineData walk and of the send-list rule: neither contrib header restates them"""
    for name in ("ryujin_export_offline.h", "hyperbolic_module_hip.h"):
        text = open(os.path.join(ROOT, "contrib", name)).read()
        assert "ryujin_hip_binding::fill_from_accessors<dim>" in text, name
        for walk in ("cij_matrix()", ".import_targets()", ".ghost_targets()", "coupling_boundary_pairs()"):
            assert walk not in text, (name, walk)
    binding = open(os.path.join(ROOT, "contrib", "ryujin_hip_binding.h")).read()
    assert binding.count("ryujin_ghost_row_send_entries(") == 2      # count + fill, nothing hand-written


@pytest.mark.skipif(not os.path.isdir("/root/reference/source") or shutil.which("patch") is None,
                    reason="needs the reference tree (build container only)")
@pytest.mark.parametrize("patch_file,files", [
    ("hyperbolic_module_hip.patch", ["hyperbolic_module.h", "hyperbolic_module.cc"]),
    ("ryujin_export_offline.patch", ["time_loop.h", "time_loop.template.h"])])
def test_patches_apply_to_the_reference(tmp_path, patch_file, files):
    os.makedirs(tmp_path / "source")
    for f in files:
        shutil.copy(os.path.join("/root/reference/source", f), tmp_path / "source" / f)
    res = subprocess.run(["patch", "-p1", "--dry-run", "-i", os.path.join(ROOT, "contrib", patch_file)],
                         cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr

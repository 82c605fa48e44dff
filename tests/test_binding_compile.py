"""The class a ryujin maintainer includes, in front of a compiler (the north_star's "keeping the
HyperbolicModule/TimeIntegrator call surface ... so it drops into time_loop unchanged").

contrib/hyperbolic_module_hip.h replaces the class of source/hyperbolic_module.h under RYUJIN_WITH_HIP
(contrib/hyperbolic_module_hip.patch). Here the recipe of contrib/README.md is carried out on a COPY of the reference
tree in a temporary directory -- copy the two headers, `patch -p1` both patches, write compile_time_options.h the way
cmake would -- and the reference's own, unmodified time_integrator.template.h (every scheme: step_ssprk_33,
step_erk_33, ... :207-560) and vtu_output.template.h are explicitly instantiated on top of the adapter for all four
Descriptions, together with the calls TimeLoop makes and contrib/ryujin_export_offline.h (tests/cpp/binding_compile.cc).

deal.II is not installed in this image: the compile runs against tests/cpp/dealii_mock/, stand-in headers with the
names and signatures ryujin's headers use and no behaviour. That is scaffolding for a TYPE check of the boundary --
every call between the reference's code and the adapter is resolved by the compiler against real code on both sides --
not an oracle and not a build of the reference: nothing compiled here is executed. What it cannot show (that the
stand-in signatures are deal.II's) is limited to the adapter's own deal.II calls: Partitioner::{ghost_targets,
import_targets, import_indices, local_to_global}, Vector::local_element, DoFTools::map_dofs_to_support_points,
ParameterAcceptor::add_parameter, Utilities::MPI::{this_mpi_process, n_mpi_processes}, dealii::Timer.

Runs where the reference tree is present (this container); skipped elsewhere."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers_reference_tree as reftree  # noqa: E402

MOCK = reftree.MOCK

pytestmark = pytest.mark.skipif(not reftree.available(),
                                reason="needs the reference tree, g++ and patch (the build container)")


@pytest.fixture(scope="module")
def patched_tree(tmp_path_factory):
    """all three patches: hyperbolic_module_hip.patch, ryujin_export_offline.patch, time_integrator_hip.patch"""
    return reftree.make_patched_tree(str(tmp_path_factory.mktemp("ryujin")))


@pytest.mark.parametrize("directory,description", [("euler", "Euler"), ("shallow_water", "ShallowWater"),
                                                   ("euler_aeos", "EulerAEOS"),
                                                   ("scalar_conservation", "ScalarConservation")])
def test_reference_time_integrator_compiles_on_the_adapter(patched_tree, tmp_path, directory, description):
    obj = os.path.join(tmp_path, "binding_compile.o")
    cmd = ["g++", "-std=c++17", "-c", "-fopenmp", "-Wall", "-DRYUJIN_WITH_HIP",
           f'-DRYUJIN_DESCRIPTION_HEADER="{directory}/description.h"', f"-DRYUJIN_DESCRIPTION={description}::Description",
           "-I" + MOCK, "-I" + patched_tree, "-I" + os.path.join(patched_tree, directory),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "binding_compile.cc"), "-o", obj]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-6000:]
    # no warning may point into the files of this repository
    ours = [ln for ln in res.stderr.splitlines() if "warning:" in ln and
            any(name in ln for name in ("hyperbolic_module_hip.h", "ryujin_hip_binding.h", "ryujin_export_offline.h"))]
    assert not ours, "\n".join(ours)
    # what was instantiated: the reference's schemes on the adapter's step<stages>, and nothing but the C ABI left open
    syms = subprocess.run(["nm", "-C", obj], capture_output=True, text=True).stdout
    for member in ("step_ssprk_33", "step_erk_33", "step_erk_54", "step_ssprk_22"):
        assert re.search(rf"TimeIntegrator<ryujin::{description}::Description, 2, double>::{member}\(", syms), member
    for stages in range(5):
        assert f"HyperbolicModule<ryujin::{description}::Description, 2, double>::step<{stages}>(" in syms, stages
    undefined_abi = sorted({ln.split()[-1] for ln in syms.splitlines() if " U ryujin_" in ln})
    header = open(os.path.join(ROOT, "include", "ryujin_hip.h")).read() + \
        open(os.path.join(ROOT, "include", "ryujin_offline_io.h")).read()
    assert undefined_abi and all(re.search(rf"\b{name}\s*\(", header) for name in undefined_abi), undefined_abi
    for name in ("ryujin_hip_create", "ryujin_hip_prepare_state_vector", "ryujin_hip_step", "ryujin_hip_time_step_fn",
                 "ryujin_hip_host_register", "ryujin_hip_state_download_owned", "ryujin_hip_state_download_prepared",
                 "ryujin_offline_write"):
        assert name in undefined_abi, (name, undefined_abi)

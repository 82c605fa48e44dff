"""A patched COPY of the reference tree in a temporary directory, the way contrib/README.md tells a ryujin maintainer
to make it: copy contrib/*.h into source/, `patch -p1` the contrib patches, write compile_time_options.h the way cmake
would. Used by the type check (tests/test_binding_compile.py) and by the executed binding test
(tests/test_binding_run.py). Nothing of the reference is committed: the copy lives in pytest's tmp_path.

CLI (for iterating by hand):  python tests/helpers_reference_tree.py /tmp/ryujin_patched
"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/source"
MOCK = os.path.join(ROOT, "tests", "cpp", "dealii_mock")

HEADERS = ("hyperbolic_module_hip.h", "ryujin_hip_binding.h", "ryujin_export_offline.h")
PATCHES = ("hyperbolic_module_hip.patch", "ryujin_export_offline.patch", "time_integrator_hip.patch")


def available() -> bool:
    return bool(os.path.isdir(REFERENCE) and shutil.which("g++") and shutil.which("patch"))


def make_patched_tree(top: str, patches=PATCHES) -> str:
    """Returns <top>/source. `patches`: which of contrib/*.patch to apply (hyperbolic_module_hip.patch alone = the
    unmodified caller)."""
    src = os.path.join(top, "source")
    if os.path.isdir(src):
        shutil.rmtree(src)
    shutil.copytree(REFERENCE, src)
    for name in HEADERS:
        shutil.copy(os.path.join(ROOT, "contrib", name), src)
    for patch in patches:
        res = subprocess.run(["patch", "-p1", "-i", os.path.join(ROOT, "contrib", patch)], cwd=top,
                             capture_output=True, text=True)
        assert res.returncode == 0, patch + "\n" + res.stdout + res.stderr
        assert "fuzz" not in res.stdout, patch + " does not apply exactly:\n" + res.stdout
    # compile_time_options.h as cmake writes it (CMakeLists.txt:69-80: NUMBER double, OpenMP on, checks off)
    text = open(os.path.join(src, "compile_time_options.h.in")).read().replace("@NUMBER@", "double")
    text = re.sub(r"#cmakedefine (\w+)",
                  lambda m: "#define " + m.group(1) if m.group(1) == "WITH_OPENMP" else "/* #undef %s */" % m.group(1),
                  text)
    open(os.path.join(src, "compile_time_options.h"), "w").write(text)
    return src


if __name__ == "__main__":
    print(make_patched_tree(sys.argv[1], PATCHES if len(sys.argv) < 3 else tuple(sys.argv[2:])))

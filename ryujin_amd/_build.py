"""Build recipes for the native libraries (explicit compiler invocations, in-tree outputs).

  ryujin_amd/lib/libryujin_synth.so   g++    synthetic OfflineData generator (host)
  ryujin_amd/lib/libryujin_hip.so     hipcc  HIP kernels + C ABI (gfx950)

The .so files are git-ignored but travel to the GPU box with the snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ryujin_amd", "csrc")
LIBDIR = os.path.join(ROOT, "ryujin_amd", "lib")
INCLUDE = os.path.join(ROOT, "include")

SYNTH_SO = os.path.join(LIBDIR, "libryujin_synth.so")
HIP_SO = os.path.join(LIBDIR, "libryujin_hip.so")


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def _run(cmd: list[str]) -> None:
    print("+", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.run(cmd, check=True)


def _sources(directory: str, exts: tuple[str, ...]) -> list[str]:
    out = []
    for base, _, files in os.walk(directory):
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def _headers() -> list[str]:
    return _sources(INCLUDE, (".h",))


def build_synth(force: bool = False) -> str:
    src = [os.path.join(CSRC, "offline_synthetic.cc"), os.path.join(CSRC, "offline_io.cc")]
    if force or not _newer(SYNTH_SO, src + _headers()):
        os.makedirs(LIBDIR, exist_ok=True)
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-Wall", "-I" + INCLUDE, "-I" + CSRC,
              *src, "-o", SYNTH_SO])
    return SYNTH_SO


def hipcc_path() -> str | None:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_hip(force: bool = False, defines: tuple = (), out: str | None = None) -> str:
    """defines/out: build an A/B variant (e.g. defines=("RYUJIN_OCC_DIJ=3",)) into another file;
    select it at run time with RYUJIN_HIP_LIB=<path>. The production library is always built with
    -ffp-contract=off (the parity contract); RYUJIN_FP_CONTRACT is honoured for variant builds only."""
    src = [os.path.join(CSRC, "ryujin_hip.hip")]
    deps = src + _sources(CSRC, (".hpp", ".h", ".hip")) + _headers()
    if out is not None:
        hipcc = hipcc_path()
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-ffp-contract=" + os.environ.get("RYUJIN_FP_CONTRACT", "off"), *["-D" + d for d in defines], "-I" + INCLUDE, "-I" + CSRC, *src,
              "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", out])
        return out
    if force or not _newer(HIP_SO, deps):
        hipcc = hipcc_path()
        if hipcc is None:
            if os.path.exists(HIP_SO):
                return HIP_SO  # GPU box without a usable hipcc: use the shipped build
            raise RuntimeError("hipcc not found and no prebuilt libryujin_hip.so")
        os.makedirs(LIBDIR, exist_ok=True)
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
              "-ffp-contract=off", "-I" + INCLUDE, "-I" + CSRC, *src,
              "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-o", HIP_SO])
    return HIP_SO


def build_all(force: bool = False) -> None:
    build_synth(force)
    build_hip(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)

"""Build recipe of the CPU oracle (TEST INFRASTRUCTURE): g++ on oracle/oracle_capi.cc ->
oracle/build/libryujin_oracle.so. Lives next to the oracle, not in the product package: only tests/,
__graft_entry__ (build()/smoke()) and bench.py's cpu_baseline leg may use it."""
from __future__ import annotations

import os
import subprocess
import sys

ORACLE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(ORACLE)
INCLUDE = os.path.join(ROOT, "include")
ORACLE_SO = os.path.join(ORACLE, "build", "libryujin_oracle.so")


def _sources(directory: str, exts: tuple) -> list:
    out = []
    for base, _, files in os.walk(directory):
        if os.path.basename(base) == "build":
            continue
        for f in files:
            if f.endswith(exts):
                out.append(os.path.join(base, f))
    return sorted(out)


def _newer(target: str, sources: list) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def build_oracle(force: bool = False, march_native: bool = False, out: str | None = None) -> str:
    """march_native / out: the -march=native build bench.py times as the CPU baseline on the GPU box's host."""
    target = out or ORACLE_SO
    src = [os.path.join(ORACLE, "oracle_capi.cc")]
    deps = src + _sources(ORACLE, (".hpp", ".h", ".cc")) + _sources(INCLUDE, (".h",))
    if force or not _newer(target, deps):
        os.makedirs(os.path.dirname(target), exist_ok=True)
        cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wall",
               *(["-march=native"] if march_native else []), "-I" + INCLUDE, "-I" + ORACLE, *src, "-o", target]
        print("+", " ".join(cmd), file=sys.stderr, flush=True)
        subprocess.run(cmd, check=True)
    return target


if __name__ == "__main__":
    build_oracle(force="--force" in sys.argv)

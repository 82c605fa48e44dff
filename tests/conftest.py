import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the oracle is an OpenMP library: under pytest-xdist every worker would bring its own team of spinning threads
if "PYTEST_XDIST_WORKER" in os.environ:
    os.environ.setdefault("OMP_NUM_THREADS", "2")
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "multigpu: needs >= 2 GPUs (RCCL with more than one rank); also carries "
                                       "`gpu`, skips itself on a single-GPU box")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

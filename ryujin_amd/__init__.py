"""ryujin_amd -- MI355X-native HyperbolicModule::step (hand-written HIP behind a C ABI).

Only the hot path lives here: csrc/ holds the HIP kernels and the
C ABI; module.py mirrors the reference's HyperbolicModule/TimeIntegrator call surface;
offline.py wraps the synthetic OfflineData generator.
"""
from . import capi  # noqa: F401
from .module import HyperbolicModule, Restart, StateVector, TauError, TimeIntegrator  # noqa: F401
from .offline import MeshSpec, SyntheticOffline  # noqa: F401

__all__ = ["capi", "HyperbolicModule", "Restart", "StateVector", "TauError", "TimeIntegrator",
           "MeshSpec", "SyntheticOffline"]

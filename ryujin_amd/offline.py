"""Synthetic OfflineData (structured Q1 meshes) -- Python handle over libryujin_synth.so.

Mesh recipes for the BASELINE.json configurations live here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class MeshSpec:
    dim: int
    n_cells: tuple
    lower: tuple
    upper: tuple
    bc: tuple  # ids on -x,+x,-y,+y,-z,+z
    cut_kind: int = capi.CUT_NONE
    cut_lo: tuple = (0.0, 0.0, 0.0)
    cut_hi: tuple = (0.0, 0.0, 0.0)
    cyl_center: tuple = (0.0, 0.0)
    cyl_radius: float = 0.0
    cut_bc: int = capi.BC_SLIP
    n_ranks: int = 1
    rank: int = 0
    name: str = field(default="mesh")

    def to_c(self) -> capi.SynthSpec:
        s = capi.SynthSpec()
        s.dim = self.dim
        pad3 = lambda t, fill: tuple(t) + (fill,) * (3 - len(t))  # noqa: E731
        s.n_cells = (C.c_uint32 * 3)(*pad3(self.n_cells, 1))
        s.lower = (C.c_double * 3)(*pad3(self.lower, 0.0))
        s.upper = (C.c_double * 3)(*pad3(self.upper, 1.0))
        bc = tuple(self.bc) + (capi.BC_DO_NOTHING,) * (6 - len(self.bc))
        s.bc = (C.c_int * 6)(*bc)
        s.cut_kind = self.cut_kind
        s.cut_lo = (C.c_double * 3)(*pad3(self.cut_lo, 0.0))
        s.cut_hi = (C.c_double * 3)(*pad3(self.cut_hi, 0.0))
        s.cyl_center = (C.c_double * 2)(*self.cyl_center)
        s.cyl_radius = self.cyl_radius
        s.cut_bc = self.cut_bc
        s.n_ranks = self.n_ranks
        s.rank = self.rank
        return s


class _OfflineArrays:
    """numpy accessors over a ryujin_hip_offline view (`self.c`); shared by the generator and the importer."""

    def _init_counts(self):
        o = self.c.contents
        self.n_export, self.n_internal = o.n_export, o.n_internal
        self.n_owned, self.n_relevant = o.n_owned, o.n_relevant
        self.measure_of_omega = o.measure_of_omega
        self.n_bdry, self.n_pairs = o.n_bdry, o.n_pairs

    def set_initial_precomputed(self, values: np.ndarray) -> None:
        """Attach initial_precomputed (shallow water: bathymetry Z_i per local index, ghosts included),
        hyperbolic_module.template.h:84-85."""
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        assert v.size == self.n_relevant
        self._initial_precomputed = v  # keep alive
        self.c.contents.initial_precomputed = capi.as_ptr(v, capi.c_double_p)

    def save(self, path: str, n_initial_precomputed: int | None = None) -> None:
        """Write the arrays as an OfflineData dump (include/ryujin_offline_io.h)."""
        nip = n_initial_precomputed
        if nip is None:
            nip = 1 if self.c.contents.initial_precomputed else 0
        pos = np.ascontiguousarray(self.positions)
        bpos = np.ascontiguousarray(self.b_positions)
        rc = self._lib.ryujin_offline_write(path.encode(), self.c, self.dim, nip,
                                            capi.as_ptr(pos, capi.c_double_p),
                                            capi.as_ptr(bpos, capi.c_double_p))
        if rc != 0:
            raise RuntimeError(self._lib.ryujin_offline_io_last_error().decode())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # numpy copies of the arrays (tests / initial data)
    def _arr(self, ptr, n, dtype):
        return capi.np_from_ptr(ptr, n, dtype)

    @property
    def row_starts(self):
        return self._arr(self.c.contents.row_starts, self.n_relevant + 1, np.uint64)

    @property
    def columns(self):
        return self._arr(self.c.contents.columns, self.nnz, np.uint32)

    @property
    def cij(self):
        return self._arr(self.c.contents.cij, self.nnz * self.dim, np.float64).reshape(-1, self.dim)

    @property
    def mij(self):
        return self._arr(self.c.contents.mij, self.nnz, np.float64)

    @property
    def mi(self):
        return self._arr(self.c.contents.mi, self.n_relevant, np.float64)

    @property
    def b_i(self):
        return self._arr(self.c.contents.b_i, self.n_bdry, np.uint32)

    @property
    def b_id(self):
        return self._arr(self.c.contents.b_id, self.n_bdry, np.uint8)

    @property
    def b_normal(self):
        return self._arr(self.c.contents.b_normal, self.n_bdry * self.dim, np.float64).reshape(-1, self.dim)

    @property
    def pairs(self):
        o = self.c.contents
        return (self._arr(o.p_i, self.n_pairs, np.uint32), self._arr(o.p_col, self.n_pairs, np.uint32),
                self._arr(o.p_j, self.n_pairs, np.uint32))


class SyntheticOffline(_OfflineArrays):
    """Owns a ryujin_synth handle; `.c` is the ryujin_hip_offline view passed to create()."""

    def __init__(self, spec: MeshSpec):
        self.spec = spec
        self._lib = capi.load_synth()
        cs = spec.to_c()
        self._h = self._lib.ryujin_synth_build(C.byref(cs))
        if not self._h:
            raise RuntimeError("ryujin_synth_build: " + self._lib.ryujin_synth_last_error().decode())
        self.c = self._lib.ryujin_synth_offline(self._h)  # POINTER(Offline)
        self.dim = spec.dim
        self._init_counts()
        self.nnz = int(self._lib.ryujin_synth_nnz(self._h))
        self.n_global = int(self._lib.ryujin_synth_n_global(self._h))

    def close(self):
        if self._h:
            self._lib.ryujin_synth_free(self._h)
            self._h = None

    @property
    def positions(self):
        return self._arr(self._lib.ryujin_synth_positions(self._h), self.n_relevant * self.dim,
                         np.float64).reshape(-1, self.dim)

    @property
    def global_ids(self):
        return self._arr(self._lib.ryujin_synth_global_ids(self._h), self.n_relevant, np.uint64)

    @property
    def b_positions(self):
        return self._arr(self._lib.ryujin_synth_bdry_positions(self._h), self.n_bdry * self.dim,
                         np.float64).reshape(-1, self.dim)


class ImportedOffline(_OfflineArrays):
    """An OfflineData dump read from disk: the same object as SyntheticOffline as far
    as HyperbolicModule is concerned. `spec` is None (there is no recipe to rebuild the mesh from)."""

    def __init__(self, path: str):
        self.spec = None
        self._lib = capi.load_synth()
        self._h = self._lib.ryujin_offline_read(path.encode())
        if not self._h:
            raise RuntimeError(self._lib.ryujin_offline_io_last_error().decode())
        self.c = self._lib.ryujin_offline_file_view(self._h)
        self.dim = int(self._lib.ryujin_offline_file_dim(self._h))
        self.n_initial_precomputed = int(self._lib.ryujin_offline_file_n_initial_precomputed(self._h))
        self._init_counts()
        self.nnz = int(self._lib.ryujin_offline_file_nnz(self._h))
        self.n_global = None

    def close(self):
        if self._h:
            self._lib.ryujin_offline_file_free(self._h)
            self._h = None

    @property
    def positions(self):
        p = self._lib.ryujin_offline_file_positions(self._h)
        if not p:
            raise ValueError("the dump holds no support point positions")
        return self._arr(p, self.n_relevant * self.dim, np.float64).reshape(-1, self.dim)

    @property
    def b_positions(self):
        p = self._lib.ryujin_offline_file_b_positions(self._h)
        if not p:
            raise ValueError("the dump holds no boundary positions")
        return self._arr(p, self.n_bdry * self.dim, np.float64).reshape(-1, self.dim)


# ------------------------------------------------------------------ mesh recipes

def rectangle_2d(n: int, lower=(0.0, 0.0), upper=(1.0, 1.0), bc=capi.BC_SLIP, n_ranks=1, rank=0,
                 ny: int | None = None) -> MeshSpec:
    bcs = (bc,) * 4 if isinstance(bc, int) else tuple(bc)
    return MeshSpec(2, (n, ny or n), lower, upper, bcs, n_ranks=n_ranks, rank=rank, name="rectangle")


def mach3_step_2d(cells_per_unit: int, length_units: int = 3, n_ranks=1, rank=0) -> MeshSpec:
    """[0,L]x[0,1] minus [0.6,L]x[0,0.2]: forward-facing step
    (geometry: source/geometry_step.h:41-93 without the rounded corner; data:
    prm/benchmarks/euler-mach3-forward-facing-step.prm). Left dirichlet, right do-nothing,
    everything else slip. cells_per_unit must be a multiple of 5 so that the step is grid aligned.
    For weak scaling the channel is lengthened (length_units, not necessarily an integer) and
    slab-partitioned along x with equal gridpoint counts per rank."""
    assert cells_per_unit % 5 == 0
    n_x = int(round(cells_per_unit * length_units))  # a fractional length is rounded to whole cells
    L = n_x / float(cells_per_unit)
    return MeshSpec(2, (n_x, cells_per_unit), (0.0, 0.0), (L, 1.0),
                    (capi.BC_DIRICHLET, capi.BC_DO_NOTHING, capi.BC_SLIP, capi.BC_SLIP),
                    cut_kind=capi.CUT_BOX, cut_lo=(0.6, -1.0, 0.0), cut_hi=(L + 1.0, 0.2, 0.0),
                    cut_bc=capi.BC_SLIP, n_ranks=n_ranks, rank=rank, name="mach3-step-2d")


def box_3d(n: int, lower=(-1.0, -1.0, -1.0), upper=(1.0, 1.0, 1.0), bc=capi.BC_SLIP, n_ranks=1, rank=0,
           nx: int | None = None) -> MeshSpec:
    bcs = (bc,) * 6 if isinstance(bc, int) else tuple(bc)
    return MeshSpec(3, (nx or n, n, n), lower, upper, bcs, n_ranks=n_ranks, rank=rank, name="box-3d")


def cylinder_channel_3d(cells_per_unit: int, length_units: int = 4, n_ranks=1, rank=0) -> MeshSpec:
    """[0,L]x[-1,1]x[-1,1] minus a staircase cylinder r=0.25 at x=0.6 (axis along z);
    prm/benchmarks/euler-mach3-cylinder-3d.prm."""
    n_x = int(round(cells_per_unit * length_units))
    L = n_x / float(cells_per_unit)
    return MeshSpec(3, (n_x, 2 * cells_per_unit, 2 * cells_per_unit),
                    (0.0, -1.0, -1.0), (L, 1.0, 1.0),
                    (capi.BC_DIRICHLET, capi.BC_DO_NOTHING, capi.BC_SLIP, capi.BC_SLIP, capi.BC_SLIP,
                     capi.BC_SLIP),
                    cut_kind=capi.CUT_CYLINDER, cyl_center=(0.6, 0.0), cyl_radius=0.25,
                    cut_bc=capi.BC_SLIP, n_ranks=n_ranks, rank=rank, name="mach3-cylinder-3d")

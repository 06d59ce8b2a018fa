"""Host-side mirror of the reference's operator interface for the grid-update path:
CartesianGrid / MeshField / boundary conditions / terms / integrators / LevelSetEquation /
integrate! (Python spelling: ``integrate_``), with the same names, argument meaning and error
behaviour, driving the HIP kernels through the C ABI (include/lsm.h).

The step loop, hooks, Δt arithmetic and error raising stay on the host exactly as in
src/timestepping.jl:101-122; only `_advance!` and `compute_cfl` cross into the library.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib as L

# ----------------------------------------------------------------------------- meshes.jl


class CartesianGrid:
    """CartesianGrid(lc, hc, n) — src/meshes.jl:1-5,34-42 — or CartesianGrid(lc, hc, meshsize=h) — src/meshes.jl:69-83:
    the cell count of each dimension is rounded UP, so the realised spacing is never coarser than `meshsize`."""

    def __init__(self, lc, hc, n=None, *, meshsize=None):
        lc, hc = tuple(lc), tuple(hc)
        if (n is None) == (meshsize is None):
            raise ValueError("pass either the node counts n or meshsize")
        if n is None:
            if len(lc) != len(hc):
                raise ValueError("lc and hc must have the same length")
            h = (meshsize,) * len(lc) if isinstance(meshsize, (int, float)) else tuple(meshsize)
            if len(h) != len(lc):
                raise ValueError("meshsize must be a scalar or have one entry per dimension")
            if not all(x > 0 for x in h):
                raise ValueError("meshsize must be positive in every dimension")
            if not all(b > a for a, b in zip(lc, hc)):
                raise ValueError("hc must be strictly greater than lc in every dimension")
            n = tuple(int(math.ceil((b - a) / x)) + 1 for a, b, x in zip(lc, hc, h))
        n = tuple(n)
        if not (len(lc) == len(hc) == len(n)):
            raise ValueError("all arguments must have the same length")
        self.lc = tuple(float(x) for x in lc)
        self.hc = tuple(float(x) for x in hc)
        self.n = tuple(int(x) for x in n)

    @property
    def ndim(self):
        return len(self.n)

    def size(self):
        return self.n

    def meshsize(self, dim=None):
        """(hc - lc) / (n - 1) — src/meshes.jl:109-110 (dim is 0-based)."""
        h = tuple((self.hc[d] - self.lc[d]) / (self.n[d] - 1) for d in range(self.ndim))
        return h if dim is None else h[dim]

    def getnode(self, I):
        """lc + (I-1)*h for a 0-based index tuple I — src/meshes.jl:114-130."""
        I = tuple(I)
        if not all(0 <= I[d] < self.n[d] for d in range(self.ndim)):
            raise ValueError(f"{I} is not a valid node index for this grid")
        h = self.meshsize()
        return tuple(self.lc[d] + float(I[d]) * h[d] for d in range(self.ndim))

    def coords(self):
        h = self.meshsize()
        return [self.lc[d] + np.arange(self.n[d], dtype=np.float64) * h[d] for d in range(self.ndim)]

    def __len__(self):
        return int(np.prod(self.n))

    def nodeindices(self):
        """All node indices (0-based tuples), first index fastest like CartesianIndices — src/meshes.jl:138."""
        return _cartesian_indices(self.n)

    def cellindices(self):
        """All cell indices: cell I is bounded by nodes I and I+1, n[d]-1 cells per dimension — src/meshes.jl:147."""
        return _cartesian_indices(tuple(k - 1 for k in self.n))

    def compute_index(self, x):
        """Index of the cell containing x, clamped to the grid — src/meshes.jl:155-169."""
        h = self.meshsize()
        return tuple(min(max(int(math.floor((x[d] - self.lc[d]) / h[d])), 0), self.n[d] - 2) for d in range(self.ndim))

    def getcell(self, I):
        """CartesianCell of cell I: lc = node I, hc = lc + h — src/meshes.jl:183-197 (unpacks as `lc, hc = cell`)."""
        I = tuple(I)
        if not all(0 <= I[d] < self.n[d] - 1 for d in range(self.ndim)):
            raise ValueError(f"{I} is not a valid cell index for this grid")
        h = self.meshsize()
        lc = tuple(self.lc[d] + float(I[d]) * h[d] for d in range(self.ndim))
        return CartesianCell(lc, tuple(lc[d] + h[d] for d in range(self.ndim)))

    def grid1d(self, dim=None):
        """Node coordinates along `dim` as LinRange(lc, hc, n) does — src/meshes.jl:90-91."""
        ax = [np.linspace(self.lc[d], self.hc[d], self.n[d]) for d in range(self.ndim)]
        return tuple(ax) if dim is None else ax[dim]

    def _c(self):
        g = L.LsmGrid()
        g.ndim = self.ndim
        for d in range(3):
            g.n[d] = self.n[d] if d < self.ndim else 1
            g.lc[d] = self.lc[d] if d < self.ndim else 0.0
            g.hc[d] = self.hc[d] if d < self.ndim else 1.0
        return g

    def _show(self):
        """src/meshes.jl:226-242."""
        return "\n".join([f"CartesianGrid in ℝ{_superscript(self.ndim)}"] + _grid_fields(self))

    __repr__ = _show


class CartesianCell:
    """A cell of a CartesianGrid: the box between the nodes at `lc` and `hc` — src/meshes.jl:171-181."""

    def __init__(self, lc, hc):
        self.lc, self.hc = tuple(lc), tuple(hc)

    def __iter__(self):
        return iter((self.lc, self.hc))

    def _show(self):
        """src/meshes.jl:243-250"""
        f = lambda c: "(" + ", ".join(_sig4(x) for x in c) + ")"
        return f"CartesianCell in ℝ{_superscript(len(self.lc))}\n  ├─ lower corner: {f(self.lc)}\n  └─ upper corner: {f(self.hc)}"

    __repr__ = _show


def _cartesian_indices(shape):
    """index tuples of an array of `shape`, first index fastest (Julia's CartesianIndices order)"""
    import itertools
    return [tuple(reversed(t)) for t in itertools.product(*[range(k) for k in reversed(shape)])]


def nodeindices(g):
    return g.nodeindices()


def cellindices(g):
    return g.cellindices()


def getnode(g, *I):
    return g.getnode(I[0] if len(I) == 1 and not isinstance(I[0], int) else I)


def getcell(g, *I):
    return g.getcell(I[0] if len(I) == 1 and not isinstance(I[0], int) else I)


def active_nodeindices(phi):
    """active_nodeindices(ϕ) — src/meshfield.jl:134 (dense: every node) / :462 (band: the stored nodes)."""
    return phi.active_nodeindices() if hasattr(phi, "active_nodeindices") else phi.mesh.nodeindices()


def active_cellindices(phi):
    return phi.active_cellindices() if hasattr(phi, "active_cellindices") else phi.mesh.cellindices()


def update_band_(phi):
    """update_band!(ϕ) — src/meshfield.jl:553 (dense: no-op) / :555-588 (band)."""
    if isinstance(phi, LevelSetEquation):
        return phi.update_band()
    return phi.rebuild() if hasattr(phi, "rebuild") else phi


# ----------------------------------------------------------------------------- boundaryconditions.jl

class BoundaryCondition:
    pass


class PeriodicBC(BoundaryCondition):
    kind, degree = L.BC_PERIODIC, 0

    def __repr__(self):
        return "Periodic"


class ExtrapolationBC(BoundaryCondition):
    """ExtrapolationBC{P} — src/boundaryconditions.jl:40-46."""
    kind = L.BC_EXTRAPOLATION

    def __init__(self, P=0):
        if P < 0:
            raise ValueError("extrapolation order P must be at least 0")
        self.degree = int(P)

    def __repr__(self):
        return {0: "Neumann", 1: "Linear extrapolation"}.get(self.degree, f"Degree {self.degree} extrapolation")


def NeumannBC():
    return ExtrapolationBC(0)


def LinearExtrapolationBC():
    return ExtrapolationBC(1)


class SymmetryBC(BoundaryCondition):
    kind, degree = L.BC_SYMMETRY, 0

    def __repr__(self):
        return "Symmetry"


def _same_bc(a, b):
    return a.kind == b.kind and a.degree == b.degree


def _normalize_bc(bc, dim):
    """src/boundaryconditions.jl:166-188."""
    if isinstance(bc, BoundaryCondition):
        return tuple((bc, bc) for _ in range(dim))
    if len(bc) != dim:
        raise ValueError("invalid number of boundary conditions")
    out = []
    for i, b in enumerate(bc):
        if isinstance(b, BoundaryCondition):
            out.append((b, b))
            continue
        if not (len(b) == 2 and all(isinstance(x, BoundaryCondition) for x in b)):
            raise ValueError(f"invalid boundary condition for dimension {i + 1}")
        left, right = b
        if isinstance(left, PeriodicBC) != isinstance(right, PeriodicBC):
            raise ValueError(f"periodic boundary conditions cannot be mixed with others in dimension {i + 1}")
        out.append((left, right))
    return tuple(out)


def _bc_c(bcs, ndim, slab_faces=(False, False)):
    arr = L.BcArray()
    for d in range(3):
        for s in range(2):
            if d < ndim:
                arr[d][s].kind, arr[d][s].degree = bcs[d][s].kind, bcs[d][s].degree
                if d == ndim - 1 and slab_faces[s]:
                    arr[d][s].kind, arr[d][s].degree = L.BC_NONE, 0
            else:
                arr[d][s].kind, arr[d][s].degree = L.BC_EXTRAPOLATION, 0
    return arr


# ----------------------------------------------------------------------------- show (text/plain) helpers
# The reference prints trees (src/meshes.jl:226-242, src/meshfield.jl:294-312,395-414, src/levelsetequation.jl:91-116,
# src/boundaryconditions.jl:197-211, src/timestepping.jl:94-97; checked by test/test-show.jl): `show(x)` returns the
# same text, `repr(x)` of an equation the compact one-line form.

def _superscript(n):
    return "".join("⁰¹²³⁴⁵⁶⁷⁸⁹"[int(d)] for d in str(int(n)))


def _jl_float(x):
    """A Float64 as Julia prints it: shortest round-trip digits, exponent form below 1e-4 and from 1e6 on."""
    x = float(x)
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "Inf" if x > 0 else "-Inf"
    if x == 0.0:
        return "-0.0" if math.copysign(1.0, x) < 0 else "0.0"
    r = repr(x)
    mant, _, ex = r.partition("e")
    if "e" in r:
        e = int(ex)
        digits = mant.replace("-", "").replace(".", "")
        point = (mant.replace("-", "").index(".") if "." in mant else len(mant.replace("-", ""))) + e
    else:
        digits = mant.replace("-", "").replace(".", "")
        point = mant.replace("-", "").index(".") if "." in mant else len(digits)
    lead = len(digits) - len(digits.lstrip("0"))
    digits, point = digits[lead:].rstrip("0") or "0", point - lead
    sign = "-" if x < 0 else ""
    e10 = point - 1                                  # decimal exponent of the leading digit
    if -5 < e10 < 6:                                 # Julia: plain notation for 1e-4 <= |x| < 1e6
        if point <= 0:
            return f"{sign}0.{'0' * (-point)}{digits}"
        if point >= len(digits):
            return f"{sign}{digits}{'0' * (point - len(digits))}.0"
        return f"{sign}{digits[:point]}.{digits[point:]}"
    return f"{sign}{digits[0]}.{digits[1:] or '0'}e{e10}"


def _sig4(x):
    """round(x; sigdigits = 4) printed as Julia prints the result."""
    x = float(x)
    if x == 0.0 or math.isnan(x) or math.isinf(x):
        return _jl_float(x)
    return _jl_float(float(f"{x:.3e}"))


def _bc_str(bcs):
    """src/boundaryconditions.jl:197-211."""
    allb = [b for pair in bcs for b in pair]
    if all(_same_bc(b, allb[0]) for b in allb):
        return f"{allb[0]!r} (all)"
    names = ("x", "y", "z") if len(bcs) <= 3 else tuple(f"d{i + 1}" for i in range(len(bcs)))
    return ", ".join(f"{names[d]}: " + (repr(l) if _same_bc(l, r) else f"{l!r} ↔ {r!r}") for d, (l, r) in enumerate(bcs))


def _grid_fields(g, prefix="  ", last=True):
    dom = " × ".join(f"[{_jl_float(a)}, {_jl_float(b)}]" for a, b in zip(g.lc, g.hc))
    h = "(" + ", ".join(_sig4(x) for x in g.meshsize()) + ")"
    return [f"{prefix}├─ domain:  {dom}", f"{prefix}├─ nodes:   {' × '.join(map(str, g.n))}",
            f"{prefix}{'└─' if last else '├─'} spacing: h = {h}"]


def _field_fields(mesh, bcs, valtype, extrema, prefix="  ", active=None):
    lines = _grid_fields(mesh, prefix, last=False)
    if bcs is not None:
        lines.append(f"{prefix}├─ bc:     {_bc_str(bcs)}")
    if active is not None:
        lines.append(f"{prefix}├─ active:  {active[0]} nodes ({active[1]}-layer halo)")
    if extrema is None:
        lines.append(f"{prefix}└─ valtype: {valtype}")
    else:
        lines.append(f"{prefix}├─ valtype: {valtype}")
        lines.append(f"{prefix}└─ values:  min = {_sig4(extrema[0])},  max = {_sig4(extrema[1])}")
    return lines


def _embed_show(label, text, indent="  "):
    """src/levelsetequation.jl:91-99."""
    parts = text.split("\n")
    return [f"{indent}├─ {label}: {parts[0]}"] + [f"{indent}│{line}" for line in parts[1:]]


def show(x):
    """The text/plain `show` of the reference for grids, fields, integrators and equations (test/test-show.jl)."""
    return x._show() if hasattr(x, "_show") else repr(x)


# ----------------------------------------------------------------------------- meshfield.jl

class MeshField:
    """Host-resident dense field (src/meshfield.jl:51-55): `vals` is a numpy array of shape
    grid.n in Fortran order (the reference's column-major Array), or shape (ncomp, *grid.n) for
    vector-valued coefficient fields."""

    def __init__(self, vals_or_f, grid, bc=None, dtype=None):
        # element type as in the reference (the eltype of `vals`): float64, or float32 storage when asked for / given
        if dtype is None:
            dtype = np.float32 if getattr(vals_or_f, "dtype", None) == np.float32 else np.float64
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError(f"unsupported element type {dtype}: float64 or float32")
        if callable(vals_or_f):
            xs = np.meshgrid(*grid.coords(), indexing="ij", sparse=True)
            v = vals_or_f(tuple(xs))
            if isinstance(v, (tuple, list)):
                v = np.stack([np.broadcast_to(np.asarray(c, dtype=np.float64), grid.n) for c in v])
            else:
                v = np.broadcast_to(np.asarray(v, dtype=np.float64), grid.n)
            vals = np.array(v, dtype=dtype)
        else:
            vals = np.array(vals_or_f, dtype=dtype)
        if vals.shape[-grid.ndim:] != grid.n:
            raise ValueError(f"values of shape {vals.shape} do not match the grid {grid.n}")
        self.vals = np.asfortranarray(vals) if vals.ndim == grid.ndim else vals
        self.mesh = grid
        self.bcs = None if bc is None else _normalize_bc(bc, grid.ndim)

    def has_boundary_conditions(self):
        return self.bcs is not None

    def values(self):
        return self.vals

    def copy(self):
        m = MeshField(self.vals.copy(order="K"), self.mesh, dtype=self.vals.dtype)
        m.bcs = self.bcs
        return m

    def copy_(self, src):
        """copy!(dest, src) — src/meshfield.jl:282-292: the values of `src`; dest keeps its mesh and boundary conditions."""
        v = src.values() if hasattr(src, "values") else np.asarray(src)
        if v.shape != self.vals.shape:
            raise ValueError("copy!: the fields have different sizes")
        self.vals[...] = v
        return self

    def with_bc(self, bc):
        """_add_boundary_conditions(ϕ, bc) — src/meshfield.jl:104-113: a field over the SAME values array with boundary conditions."""
        m = MeshField.__new__(MeshField)
        m.vals, m.mesh, m.bcs = self.vals, self.mesh, _normalize_bc(bc, self.mesh.ndim)
        return m

    @property
    def ndim(self):
        return self.mesh.ndim

    def meshsize(self, dim=None):
        return self.mesh.meshsize(dim)

    def __getitem__(self, I):
        """ϕ[I] with 0-based I; out-of-grid indices go through the boundary conditions
        (_getindexbc, src/meshfield.jl:213-260)."""
        I = tuple(I) if not isinstance(I, int) else (I,)
        n = self.mesh.n
        if all(0 <= I[d] < n[d] for d in range(len(n))):
            return float(self.vals[I])
        if self.bcs is None:
            raise ValueError(f"index {I} lies outside the grid, but the field has no boundary conditions to resolve it.")
        return self._getindexbc(I, len(n))

    def _getindexbc(self, I, dim):
        if dim == 0:
            return float(self.vals[I])
        d = dim - 1
        n = self.mesh.n[d]
        if 0 <= I[d] < n:
            return self._getindexbc(I, dim - 1)
        left = I[d] < 0
        bc = self.bcs[d][0 if left else 1]
        k = -I[d] if left else I[d] - (n - 1)
        b, sgn = (0, 1) if left else (n - 1, -1)
        acc = 0.0
        if bc.kind == L.BC_PERIODIC:
            J = I[:d] + (((n - 1) - k) if left else k,) + I[d + 1:]
            acc += 1.0 * self._getindexbc(J, dim - 1)
        elif bc.kind == L.BC_EXTRAPOLATION:
            P = bc.degree
            for j in range(P + 1):
                w = 1.0
                for m in range(P + 1):
                    if m != j:
                        w *= (-k - m) / (j - m)
                acc += w * self._getindexbc(I[:d] + (b + sgn * j,) + I[d + 1:], dim - 1)
        else:
            acc += 1.0 * self._getindexbc(I[:d] + (b + sgn * k,) + I[d + 1:], dim - 1)
        return acc

    # ---- set operations on level sets (src/levelsetops.jl:246-325): union = min, intersection = max,
    #      complement = negation, difference = max(ϕ₁, -ϕ₂); the in-place forms return self, as the reference's `!` forms
    def _other(self, o):
        if not isinstance(o, MeshField) or o.mesh.n != self.mesh.n:
            raise ValueError("set operations need two MeshFields on the same grid")
        return o.vals

    def union_(self, o):
        np.minimum(self.vals, self._other(o), out=self.vals)
        return self

    def intersect_(self, o):
        np.maximum(self.vals, self._other(o), out=self.vals)
        return self

    def complement_(self):
        np.negative(self.vals, out=self.vals)
        return self

    def setdiff_(self, o):
        np.maximum(self.vals, -self._other(o), out=self.vals)
        return self

    def union(self, o):
        return self.copy().union_(o)

    def intersect(self, o):
        return self.copy().intersect_(o)

    def complement(self):
        return self.copy().complement_()

    def setdiff(self, o):
        return self.copy().setdiff_(o)

    __or__, __and__, __sub__, __neg__ = union, intersect, setdiff, complement     # ϕ₁ ∪ ϕ₂, ϕ₁ ∩ ϕ₂, setdiff, complement

    def _show(self):
        """src/meshfield.jl:294-312."""
        scalar = self.vals.ndim == self.mesh.ndim
        et = "Float32" if self.vals.dtype == np.float32 else "Float64"
        vt = et if scalar else f"SVector{{{self.vals.shape[0]}, {et}}}"
        ext = (self.vals.min(), self.vals.max()) if scalar else None
        return "\n".join([f"MeshField on CartesianGrid in ℝ{_superscript(self.mesh.ndim)}"] + _field_fields(self.mesh, self.bcs, vt, ext))

    __repr__ = _show


class LazyMeshField:
    """MeshField(f, grid) whose samples are only ever materialised slab by slab (large grids,
    multi-GPU): `f` receives a tuple of broadcastable coordinate arrays like MeshField's."""

    def __init__(self, f, grid, bc=None, dtype=np.float64):
        self.f, self.mesh = f, grid
        self.dtype = np.dtype(dtype)
        self.bcs = None if bc is None else _normalize_bc(bc, grid.ndim)

    def has_boundary_conditions(self):
        return self.bcs is not None

    def local_values(self, slab):
        cs = self.mesh.coords()
        if slab is not None:
            cs[-1] = cs[-1][slab[0]:slab[0] + slab[1]]
        xs = np.meshgrid(*cs, indexing="ij", sparse=True)
        shape = tuple(len(c) for c in cs)
        return np.asfortranarray(np.broadcast_to(np.asarray(self.f(tuple(xs)), dtype=np.float64), shape))


class ROCMeshField:
    """Device-resident field: a padded HBM buffer (layout from lsm_layout) + the grid + the
    normalised boundary conditions + the backend handle.  Plays the role of the
    `ROCMeshField <: AbstractMeshField` of SURVEY.md §8b."""

    def __init__(self, backend, mesh, bcs, buf=None):
        self.backend, self.mesh, self.bcs = backend, mesh, bcs
        self.buf = backend.alloc() if buf is None else buf
        self.ghosts_dirty = True    # set whenever the interior is rewritten from outside a step

    @classmethod
    def from_host(cls, backend, mf, bcs=None, local=None):
        f = cls(backend, mf.mesh, bcs if bcs is not None else mf.bcs)
        v = mf.vals if local is None else mf.vals[(Ellipsis, local)]
        backend.upload(f.buf, v)
        return f

    def values(self):
        """Host copy of the (local) interior, Fortran order."""
        return self.backend.download(self.buf)

    def to_host(self):
        m = MeshField(self.values(), self.mesh)
        m.bcs = self.bcs
        return m

    def copy(self):
        return ROCMeshField(self.backend, self.mesh, self.bcs, self.backend.clone(self.buf))

    def copy_(self, src):
        """copy!(dest, src) — src/meshfield.jl:289-292; accepts a device or a host field."""
        if isinstance(src, ROCMeshField):
            self.backend.copy_(self.buf, src.buf)
        else:
            self.backend.upload(self.buf, src.vals if isinstance(src, MeshField) else src)
        self.ghosts_dirty = True
        return self

    def extrema(self):
        return self.backend.extrema(self.buf)

    def _offset(self, I):
        I = tuple(I) if not isinstance(I, int) else (I,)
        lay = self.backend.lay
        if not all(0 <= I[d] < int(lay.n[d]) for d in range(len(I))):
            raise IndexError(f"index {I} is outside the (local) grid; ghost values are resolved inside the kernels")
        return int(lay.origin) + sum(int(I[d]) * int(lay.stride[d]) for d in range(len(I)))

    def __getitem__(self, I):
        """ϕ[I] (0-based, in-grid): scalar device read — slow, for tests and hooks (src/meshfield.jl:213-217)."""
        return float(self.buf[self._offset(I)].item())

    def __setitem__(self, I, val):
        """ϕ[I] = v (src/meshfield.jl:263-266): scalar device write; marks the ghost layers stale."""
        self.buf[self._offset(I)] = float(val)
        self.ghosts_dirty = True

    def _valtype(self):
        return "Float32" if np.dtype(getattr(self.backend, "dtype", np.float64)) == np.float32 else "Float64"

    def _show(self):
        """src/meshfield.jl:294-312 — the device field prints as the reference's MeshField does (extrema by lsm_extrema)."""
        return "\n".join([f"MeshField on CartesianGrid in ℝ{_superscript(self.mesh.ndim)}"] +
                         _field_fields(self.mesh, self.bcs, self._valtype(), self.extrema()))

    __repr__ = _show


class NarrowBandMeshField:
    """NarrowBandMeshField(ϕ::MeshField; nlayers = 3) / NarrowBandMeshField(f, grid; bc, nlayers)
    (src/meshfield.jl:411-440): host-side description of a field restricted to the topological band
    of `nlayers` nodes around the cut cells.  Passing it as `ic` makes the equation evolve the band only."""

    def __init__(self, phi_or_f, grid=None, bc=None, nlayers=3):
        base = phi_or_f if isinstance(phi_or_f, MeshField) else MeshField(phi_or_f, grid, bc=bc)
        if base.bcs is not None and any(b.kind == L.BC_PERIODIC for pair in base.bcs for b in pair):
            raise ValueError("PeriodicBC is not supported on a NarrowBandMeshField")   # src/meshfield.jl:339-340
        self.base, self.mesh, self.bcs, self.nlayers = base, base.mesh, base.bcs, int(nlayers)
        self.vals = base.vals

    def has_boundary_conditions(self):
        return self.bcs is not None


class ROCNarrowBandMeshField(ROCMeshField):
    """Device narrow band: dense padded values + a byte mask of active nodes (+ the halo mask and
    tile flags derived from it).  Non-band entries of the value array are scratch: before every
    stage they are refilled, within 3 nodes of the band, with the reference's affine extrapolant."""
    MC = int(os.environ.get("LSM_BAND_MC", "16"))   # planes per brick in band mode (tile = 32 x 8 x MC in 3-D; 16 measured best at 768^3: tools/band_probe.py)
    HALO = 3

    def __init__(self, backend, mesh, bcs, nlayers, buf=None):
        super().__init__(backend, mesh, bcs, buf)
        self.nlayers = int(nlayers)
        self.mask = backend.alloc_mask()
        self.halo = backend.alloc_mask()
        self.tiles = backend.alloc_tiles(self.MC)
        self._scratch = (backend.alloc_mask(), backend.alloc_mask())   # reused by every rebuild
        self._hlist = self._hcount = None

    def rebuild(self, from_dense=False):
        """update_band! (src/meshfield.jl:555-588) + what is derived from the new band: tile flags, the
        halo mask and the (halo node -> nearest band node) list every stage input is filled from."""
        b = self.backend
        if self._hlist is None:
            self._hlist, self._hcount = b.alloc_halo_list(1 << 16)
        b.band_update(self.buf, self.mask, from_dense, self.nlayers, self._scratch[0], self._scratch[1], self.halo,
                      self.tiles, self.MC, self._hlist, self._hcount)
        self._check_halo()
        self.ghosts_dirty = True

    def _check_halo(self):
        b = self.backend
        want, missed = b.band_status(self._hcount)
        if want > self._hlist.numel() // 2:          # list too short: grow it and search again
            self._hlist, self._hcount = b.alloc_halo_list(2 * want)
            b.band_halo(self.buf, self.mask, self.halo, self.tiles, self.MC, self._hlist, self._hcount)
            want, missed = b.band_status(self._hcount)
        if missed:                                   # src/meshfield.jl:499-500
            raise ValueError("a stencil of the band reads a node more than 6 nodes away from every band node")

    def prepare(self, buf):
        """Make `buf` readable by stencils: band halo (extrapolation) then out-of-grid ghosts (BCs)."""
        self.backend.band_prepare(buf, self.mask, self._hlist, self._hcount, self.tiles, self.MC)

    def active_mask(self):
        return self.backend.mask_to_host(self.mask)

    def active_count(self):
        return self.backend.band_count(self.mask)

    def _show(self):
        """src/meshfield.jl:395-414: the extrema are those of the band's values (off-band entries are scratch)."""
        vals = self.values()[self.active_mask()]
        ext = (vals.min(), vals.max()) if vals.size else (float("nan"), float("nan"))
        return "\n".join([f"NarrowBandMeshField on CartesianGrid in ℝ{_superscript(self.mesh.ndim)}"] +
                         _field_fields(self.mesh, self.bcs, self._valtype(), ext, active=(self.active_count(), self.nlayers)))

    __repr__ = _show

    def active_nodeindices(self):
        return [tuple(int(i) for i in I) for I in np.argwhere(self.active_mask())]

    def active_cellindices(self):
        """Cells whose 2^N corners are all band nodes (src/meshfield.jl:364-369), by their lower corner (0-based)."""
        m = self.active_mask()
        N = m.ndim
        ok = np.ones(tuple(k - 1 for k in m.shape), dtype=bool)
        for off in np.ndindex(*(2,) * N):
            ok &= m[tuple(slice(o, m.shape[d] - 1 + o) for d, o in enumerate(off))]
        return [tuple(int(i) for i in I) for I in np.argwhere(ok)]

    def values(self):
        """Host copy: stored values on the band, NaN elsewhere."""
        v = self.backend.download(self.buf)
        v[~self.active_mask()] = np.nan
        return v

    def copy(self):
        c = ROCNarrowBandMeshField(self.backend, self.mesh, self.bcs, self.nlayers, self.backend.clone(self.buf))
        c.mask.copy_(self.mask)
        c.halo.copy_(self.halo)
        c.tiles.copy_(self.tiles)
        if self._hlist is not None:
            c._hlist, c._hcount = self._hlist.clone(), self._hcount.clone()
        return c

    def copy_(self, src):
        """copy!(dest, src) (src/meshfield.jl:282-292): the values AND the active set of `src` (the Dict copy! syncs keys);
        dest keeps its own mesh, boundary conditions and nlayers."""
        if not isinstance(src, ROCNarrowBandMeshField):
            raise ValueError("copy! into a narrow-band field takes a narrow-band field")
        self.backend.copy_(self.buf, src.buf)
        self.mask.copy_(src.mask)
        self.halo.copy_(src.halo)
        self.tiles.copy_(src.tiles)
        if src._hlist is not None:
            self._hlist, self._hcount = src._hlist.clone(), src._hcount.clone()
        # the handle's compact tile lists (and a prefetched Δt) describe the band this buffer held before: rebuild them
        self.backend.band_retile(self.mask, self.tiles, self.MC)
        if self._hcount is not None:
            self.backend.band_status(self._hcount)
        self.ghosts_dirty = True
        return self

    def __getitem__(self, I):
        """ϕ[I] (src/meshfield.jl:441-445,475-511): stored value on the band, affine extrapolant from the
        nearest band node elsewhere in the grid, boundary conditions outside the grid.  Slow scalar path."""
        I = tuple(I) if not isinstance(I, int) else (I,)
        n = self.mesh.n
        if all(0 <= I[d] < n[d] for d in range(len(n))):
            off = self._offset(I)
            if not bool(self.mask[off].item()):
                t = self.backend.alloc_mask()
                t[off] = 1
                self.backend.band_fill(self.buf, self.mask, t, None, self.MC)   # whole grid: the node may be far from the band
                if self.backend.band_missed():
                    raise ValueError(f"index {I} is more than 6 nodes from the band")
            return float(self.buf[off].item())
        if self.bcs is None:
            raise ValueError(f"index {I} lies outside the grid, but the field has no boundary conditions to resolve it.")
        return self._bc_resolve(I, len(n))

    def _bc_resolve(self, I, dim):   # _getindexbc (src/meshfield.jl:248-260) over this field's own getindex
        if dim == 0:
            return self[I]
        d = dim - 1
        n = self.mesh.n[d]
        if 0 <= I[d] < n:
            return self._bc_resolve(I, dim - 1)
        left = I[d] < 0
        bc = self.bcs[d][0 if left else 1]
        k = -I[d] if left else I[d] - (n - 1)
        b, sgn = (0, 1) if left else (n - 1, -1)
        acc = 0.0
        if bc.kind == L.BC_EXTRAPOLATION:
            P = bc.degree
            for j in range(P + 1):
                w = 1.0
                for m in range(P + 1):
                    if m != j:
                        w *= (-k - m) / (j - m)
                acc += w * self._bc_resolve(I[:d] + (b + sgn * j,) + I[d + 1:], dim - 1)
        else:
            acc += 1.0 * self._bc_resolve(I[:d] + (b + sgn * k,) + I[d + 1:], dim - 1)
        return acc


# ----------------------------------------------------------------------------- derivatives.jl schemes

class SpatialScheme:
    pass


class Upwind(SpatialScheme):
    code = L.SCHEME_UPWIND


class WENO5(SpatialScheme):
    code = L.SCHEME_WENO5


# ----------------------------------------------------------------------------- coefficient catalogue
# Julia closures f(x,t) cannot run on the device (SURVEY.md §7 hard part 2).  Coefficients are:
#   numbers / tuples            -> CONST
#   RigidRotation               -> ROTATION   (the (x,t)->(-x₂,x₁) of the reference's tests/docs)
#   SeparableCoefficient        -> SEPARABLE  (per-axis tables × g(t), e.g. vortex deformation)
#   MeshField                   -> FIELD      (device arrays, SoA per component)
#   python callable f(x,t)      -> FIELD re-sampled on the host before the CFL and each stage (slow path)

class RigidRotation:
    """u = ω·(-(x₂-c₂), x₁-c₁ [, 0])."""

    def __init__(self, omega=1.0, center=(0.0, 0.0)):
        self.omega, self.center = float(omega), (float(center[0]), float(center[1]))


class SeparableCoefficient:
    """u_c(x,t) = ((T_c1[i1]·T_c2[i2])·T_c3[i3])·g(t); tables[c][axis] are 1-D arrays over the
    GLOBAL grid; time is None (g=1) or ('cos', T) for g = cos(πt/T)."""

    def __init__(self, tables, time=None):
        self.tables = [[np.asarray(t, dtype=np.float64) for t in comp] for comp in tables]
        self.time = time


def vortex_deformation(grid, period=3.0):
    """LeVeque's 3-D deformation field:
    u = 2 sin²(πx) sin(2πy) sin(2πz) g, v = -sin(2πx) sin²(πy) sin(2πz) g, w = -sin(2πx) sin(2πy) sin²(πz) g,
    g = cos(πt/period).  Tables are evaluated with the host libm at the node coordinates."""
    x, y, z = grid.coords()
    s2 = lambda a: np.sin(np.pi * a) * np.sin(np.pi * a)
    s = lambda a: np.sin(2 * np.pi * a)
    return SeparableCoefficient([[2 * s2(x), s(y), s(z)], [-s(x), s2(y), s(z)], [-s(x), s(y), s2(z)]], time=("cos", period))


class _Coeff:
    """Resolved coefficient bound to a backend: owns device buffers, fills an LsmCoeff."""

    def __init__(self, spec, ncomp, grid, backend, slab):
        self.spec, self.ncomp, self.grid, self.backend, self.slab = spec, ncomp, grid, backend, slab
        self.c = L.LsmCoeff()
        self._keep = []
        self.callable = None
        self.fields = None
        if isinstance(spec, RigidRotation):
            self.c.kind = L.COEFF_ROTATION
            self.c.value[0], self.c.value[1], self.c.value[2] = spec.omega, spec.center[0], spec.center[1]
        elif isinstance(spec, SeparableCoefficient):
            self.c.kind = L.COEFF_SEPARABLE
            if spec.time is not None:
                self.c.time_kind, self.c.time_param = L.TIME_COS, float(spec.time[1])
            for k in range(ncomp):
                t = backend.table(np.concatenate(spec.tables[k]))
                self._keep.append(t)
                self.c.sep[k] = t.data_ptr()
        elif isinstance(spec, (MeshField, ROCMeshField)) or callable(spec) or self._device_components(spec):
            self.c.kind = L.COEFF_FIELD
            self.fields = [getattr(backend, "alloc_side", backend.alloc)() for _ in range(ncomp)]
            for k, t in enumerate(self.fields):
                self.c.field[k] = t.data_ptr()
            if callable(spec):
                self.callable = spec
            else:
                self.set_values(spec)
        else:
            vals = spec if isinstance(spec, (tuple, list, np.ndarray)) else (spec,)
            if len(vals) != ncomp:
                raise ValueError(f"expected {ncomp} coefficient component(s), got {len(vals)}")
            self.c.kind = L.COEFF_CONST
            for k, v in enumerate(vals):
                self.c.value[k] = float(v)

    def _local(self, a):
        if self.slab is None:
            return a
        return a[..., self.slab[0]:self.slab[0] + self.slab[1]]

    @staticmethod
    def _device_components(spec):
        return isinstance(spec, (tuple, list)) and len(spec) > 0 and all(isinstance(x, ROCMeshField) for x in spec)

    def set_values(self, vals):
        """New values of a FIELD coefficient: a host array of shape grid.n or (ncomp, *grid.n) / a host MeshField
        (uploaded), or device fields — a ROCMeshField (scalar speed / b) or one per component (velocity) — copied
        device to device into the coefficient's float64 side arrays (a float32 field widens exactly)."""
        if isinstance(vals, ROCMeshField) or self._device_components(vals):
            comps = [vals] if isinstance(vals, ROCMeshField) else list(vals)
            if len(comps) != self.ncomp:
                raise ValueError(f"expected {self.ncomp} coefficient component(s), got {len(comps)}")
            for k, src in enumerate(comps):
                if src.buf.numel() == self.fields[k].numel() and src.buf.device == self.fields[k].device:
                    self.fields[k].copy_(src.buf)                      # same padded layout: HBM to HBM
                else:
                    v = src.values().astype(np.float64)                # another handle's layout (e.g. a whole-grid field on a slab)
                    getattr(self.backend, "upload_side", self.backend.upload)(self.fields[k], v if v.shape == tuple(self.backend.local_shape()) else self._local(v))
            return
        if isinstance(vals, MeshField):
            vals = vals.vals
        vals = np.asarray(vals, dtype=np.float64)
        if vals.ndim == self.grid.ndim:
            vals = vals[None]
        for k in range(self.ncomp):
            getattr(self.backend, "upload_side", self.backend.upload)(self.fields[k], self._local(vals[k]))

    def refresh(self, t):
        """Slow path: re-sample a python callable f(x, t) on the host."""
        if self.callable is None:
            return
        xs = tuple(np.meshgrid(*self.grid.coords(), indexing="ij", sparse=True))
        v = self.callable(xs, t)
        if not isinstance(v, (tuple, list)):
            v = (v,)
        self.set_values(np.stack([np.broadcast_to(np.asarray(c, dtype=np.float64), self.grid.n) for c in v]))


# ----------------------------------------------------------------------------- levelsetterms.jl

class LevelSetTerm:
    update_func = None
    coeff = None

    def _bind(self, grid, backend, slab):
        pass


class AdvectionTerm(LevelSetTerm):
    """AdvectionTerm(𝐮[, scheme = WENO5(), update_func]) — 𝐮 ⋅ ∇ϕ (src/levelsetterms.jl:45-63)."""
    kind = L.TERM_ADVECTION

    def __init__(self, velocity, scheme=None, update_func=None):
        self.velocity, self.scheme, self.update_func = velocity, scheme or WENO5(), update_func

    def _bind(self, grid, backend, slab):
        self.coeff = _Coeff(self.velocity, grid.ndim, grid, backend, slab)

    def __repr__(self):
        return "𝐮 ⋅ ∇ ϕ"


class CurvatureTerm(LevelSetTerm):
    """CurvatureTerm(b) — b κ|∇ϕ| (src/levelsetterms.jl:104-107)."""
    kind = L.TERM_CURVATURE

    def __init__(self, b):
        self.b = b

    def _bind(self, grid, backend, slab):
        self.coeff = _Coeff(self.b, 1, grid, backend, slab)

    def __repr__(self):
        return "b κ|∇ϕ|"


class NormalMotionTerm(LevelSetTerm):
    """NormalMotionTerm(v[, update_func]) — v|∇ϕ| (src/levelsetterms.jl:139-146)."""
    kind = L.TERM_NORMAL_MOTION

    def __init__(self, speed, update_func=None):
        self.speed, self.update_func = speed, update_func

    def _bind(self, grid, backend, slab):
        self.coeff = _Coeff(self.speed, 1, grid, backend, slab)

    def __repr__(self):
        return "v|∇ϕ|"


class EikonalReinitializationTerm(LevelSetTerm):
    """EikonalReinitializationTerm([ϕ₀]) — sign(ϕ)(|∇ϕ| - 1) (src/levelsetterms.jl:211-222).
    With ϕ₀ (host MeshField or device field) the smoothed sign S₀ = ϕ₀/√(ϕ₀²+Δx²) is frozen."""
    kind = L.TERM_EIKONAL

    def __init__(self, phi0=None):
        self.phi0 = phi0
        self.s0 = None

    def _bind(self, grid, backend, slab):
        if self.phi0 is None:
            return
        if isinstance(self.phi0, ROCMeshField):
            src = self.phi0.buf
        else:
            src = backend.alloc()
            v = self.phi0.vals if isinstance(self.phi0, MeshField) else np.asarray(self.phi0)
            backend.upload(src, v if slab is None else v[..., slab[0]:slab[0] + slab[1]])
        self.s0 = getattr(backend, "alloc_side", backend.alloc)()
        backend.eikonal_sign(src, self.s0)

    def __repr__(self):
        return "sign(ϕ) (|∇ϕ| - 1)" if self.phi0 is None else "sign(ϕ₀) (|∇ϕ| - 1)"


def _terms_c(terms):
    arr = (L.LsmTerm * max(1, len(terms)))()
    for i, t in enumerate(terms):
        arr[i].kind = t.kind
        arr[i].scheme = t.scheme.code if isinstance(t, AdvectionTerm) else 0
        if t.coeff is not None:
            C.memmove(C.byref(arr[i].coeff), C.byref(t.coeff.c), C.sizeof(L.LsmCoeff))
        if isinstance(t, EikonalReinitializationTerm) and t.s0 is not None:
            arr[i].s0 = t.s0.data_ptr()
    return arr


# ----------------------------------------------------------------------------- timestepping.jl

class TimeIntegrator:
    def __init__(self, cfl=0.5):
        self.cfl = float(cfl)

    def _show(self):
        """src/timestepping.jl:94-97."""
        return f"{self._describe}\n  └─ cfl: {_jl_float(self.cfl)}"

    __repr__ = _show


class ForwardEuler(TimeIntegrator):
    name, _describe = "fe", "ForwardEuler (1st order explicit)"


class RK2(TimeIntegrator):
    name, _describe = "rk2", "RK2 (2nd order TVD Runge-Kutta, Heun's method)"


class RK3(TimeIntegrator):
    name, _describe = "rk3", "RK3 (3rd order TVD Runge-Kutta)"


def _jl_min(*xs):
    m = xs[0]
    for x in xs[1:]:
        m = float("nan") if (math.isnan(m) or math.isnan(x)) else (x if x < m else m)
    return m


def _eps(x):
    return float(np.spacing(abs(float(x))))


# ----------------------------------------------------------------------------- slab groups inside one process

class LocalGroup:
    """Every rank of a slab decomposition as a handle of THIS process (include/lsm.h: LSM_COMM_LOCAL; any devices):
    `g = LocalGroup(world)`, then `LevelSetEquation(..., comm=g.rank(r), device=...)` on one host thread per rank.
    Ghost planes move by peer copies inside the library; no torch.distributed, no RCCL."""

    def __init__(self, world):
        import threading
        self.world = int(world)
        self._barrier = threading.Barrier(self.world)
        self._slots = [None] * self.world
        self._backends = []

    def rank(self, r):
        if not 0 <= r < self.world:
            raise ValueError("rank out of range")
        return _LocalRank(self, int(r))

    def abort(self):
        """A rank has failed: nobody may keep waiting for it — neither at this object's barrier nor inside the library
        (lsm_comm_abort: the exchanges and the Δt all-reduce of the other ranks return LSM_ERR_COMM)."""
        self._barrier.abort()
        for b in self._backends:
            b.comm_abort()

    def exchange(self, r, v):
        """all_gather_object among the ranks' threads."""
        self._slots[r] = v
        self._barrier.wait()
        out = list(self._slots)
        self._barrier.wait()
        return out


class _LocalRank:
    def __init__(self, group, r):
        self.group, self.r = group, r


# ----------------------------------------------------------------------------- levelsetequation.jl

class LevelSetEquation:
    BAND_OVERLAP = 10   # least number of planes of each neighbour a slab of a band keeps: nearest-band-node radius 6 + slope 1 + stencil 3

    """LevelSetEquation(; terms, integrator = RK2(), ic, bc = nothing, t = 0) — src/levelsetequation.jl:59-78.

    Extra keywords of this implementation: mode ('fast' | 'strict' arithmetic), device, and
    `comm` (a torch.distributed process group, or a rank of an in-process LocalGroup: the grid is then split into
    slabs of the last dimension, one per rank, with the ghost-plane exchange inside the library — RCCL or peer copies)."""

    def __init__(self, *, terms, ic, integrator=None, bc=None, t=0, mode="fast", device=0, comm=None, backend_factory=None, tuning=None):
        if isinstance(terms, LevelSetTerm):
            terms = (terms,)
        if not (isinstance(terms, tuple) and all(isinstance(x, LevelSetTerm) for x in terms)):
            raise ValueError(f"terms must be a LevelSetTerm or a tuple of them, got {type(terms)}")
        if len(terms) == 0 or len(terms) > L.MAX_TERMS:
            raise ValueError(f"between 1 and {L.MAX_TERMS} terms are supported")
        self.terms = terms
        self.integrator = integrator or RK2()
        if bc is None:
            if not ic.has_boundary_conditions():
                raise ValueError("no boundary conditions: pass `bc` or build `ic` with one")
            bcs = ic.bcs
        else:
            bcs = _normalize_bc(bc, ic.mesh.ndim)
        self.mesh_ = ic.mesh
        self.bcs = bcs
        self.t = t
        self.comm = comm
        grid = ic.mesh
        N = grid.ndim
        # slab decomposition of the last dimension
        self.rank, self.world = 0, 1
        self.slab = None
        slab_faces = (False, False)
        if comm is not None:
            if isinstance(comm, _LocalRank):
                self.rank, self.world = comm.r, comm.group.world
            else:
                import torch.distributed as dist
                self.rank, self.world = dist.get_rank(comm), dist.get_world_size(comm)
            nl = grid.n[N - 1]
            base, rem = divmod(nl, self.world)
            counts = [base + (1 if r < rem else 0) for r in range(self.world)]
            lo = sum(counts[:self.rank])
            self.slab = (lo, counts[self.rank])
            self.counts = counts
            self.own = (0, counts[self.rank])      # local plane range of the planes this rank owns
            if self.world > 1 and min(counts) < L.GHOST + 1:
                # every rank raises the same error (a rank failing alone would leave the others in their next collective):
                # the periodic wrap sends planes shifted by one node, and a SymmetryBC end face reads LSM_GHOST planes inwards
                raise ValueError(f"a slab needs at least {L.GHOST + 1} planes: {nl} planes over {self.world} ranks leave {min(counts)}")
            if isinstance(ic, NarrowBandMeshField) and self.world > 1:
                # a band needs its neighbours' mask AND values up to 7 planes deep (nearest band node within 6, its
                # slope neighbour) plus the stencil reach: every rank keeps BAND_OVERLAP planes of its neighbours as
                # ordinary planes of its own (extended) slab, computes everything on them redundantly — results are
                # right at least BAND_OVERLAP planes away from the cut faces, i.e. on the owned planes — and refreshes
                # them from their owners after every stage and every band update.
                # reinitialize! measures physical distances: a band node lies up to (nlayers + 2)·max(h) from the interface,
                # and the ball it searches for nearer samples must stay clear of the cut faces (2 planes of wrong patches)
                hs = grid.meshsize()
                W = max(self.BAND_OVERLAP, int(math.ceil((ic.nlayers + 2) * max(hs) / hs[N - 1])) + 3)
                self.BAND_OVERLAP = W
                if min(counts) < W or ic.nlayers + 1 > W:
                    raise ValueError(f"a slab-decomposed band needs at least {W} planes per rank and nlayers < {W}")
                wlo = W if self.rank > 0 else 0
                whi = W if self.rank < self.world - 1 else 0
                self.own = (wlo, counts[self.rank])
                self.slab = (lo - wlo, counts[self.rank] + wlo + whi)
            periodic = bcs[N - 1][0].kind == L.BC_PERIODIC
            slab_faces = (self.rank > 0 or (periodic and self.world > 1), self.rank < self.world - 1 or (periodic and self.world > 1))
            self.periodic_last = periodic
        # storage type of the state = element type of `ic` (side arrays and all arithmetic stay float64)
        if isinstance(ic, ROCMeshField):
            self.dtype = ic.backend.dtype if hasattr(ic.backend, "dtype") else np.dtype(np.float64)
        elif isinstance(ic, LazyMeshField):
            self.dtype = ic.dtype
        else:
            self.dtype = np.dtype(ic.vals.dtype)
        factory = backend_factory
        if factory is None:
            from .backend import HipBackend
            factory = lambda g, b, s: HipBackend(g, b, slab=s, mode=mode, device=device, dtype=self.dtype, tuning=tuning)   # tuning: {"LSM_...": value} (include/lsm.h)
        elif self.dtype != np.float64:
            raise ValueError("float32 fields need the HIP backend")
        self.backend = factory(grid._c(), _bc_c(bcs, N, slab_faces), self.slab)
        # dense slabs on the HIP backend: the plane exchange and the Δt all-reduce run inside the library
        # (lsm_comm_attach_rccl / _local); torch.distributed only carries the RCCL unique id to the ranks
        self.lib_comm = False
        import os as _os
        band_slab = isinstance(ic, NarrowBandMeshField) and comm is not None and self.world > 1
        if (comm is not None and self.world > 1 and hasattr(self.backend, "comm_attach_rccl")
                and (band_slab or _os.environ.get("LSM_LIB_COMM", "1") != "0")):
            self._attach_library_comm()
        if band_slab:
            # the overlap planes of a band travel inside the library only (lsm_band_overlap_mask / _values)
            if not self.lib_comm:
                raise ValueError("a slab-decomposed NarrowBandMeshField needs the library's communicator (lsm_comm_attach_rccl / _local)")
            self.backend.band_overlap_config(self.BAND_OVERLAP)
        # copy `ic` so the equation owns its state (src/levelsetequation.jl:67-76)
        self.band = isinstance(ic, NarrowBandMeshField)
        if self.band:
            self.state = ROCNarrowBandMeshField(self.backend, grid, bcs, ic.nlayers)
            v = ic.vals
            self.backend.upload(self.state.buf, v if self.slab is None else v[..., self.slab[0]:self.slab[0] + self.slab[1]])
            self.state.rebuild(from_dense=True)   # NarrowBandMeshField(ϕ; nlayers): seed every node, then update_band!
            if self.comm is not None and self.world > 1:
                self._band_sync_after_update()
        else:
            self.state = ROCMeshField(self.backend, grid, bcs)
        if self.band:
            pass
        elif isinstance(ic, ROCMeshField):
            self.backend.copy_(self.state.buf, ic.buf)
        elif isinstance(ic, LazyMeshField):
            self.backend.upload(self.state.buf, ic.local_values(self.slab))
        else:
            v = ic.vals
            self.backend.upload(self.state.buf, v if self.slab is None else v[..., self.slab[0]:self.slab[0] + self.slab[1]])
        for term in terms:
            term._bind(grid, self.backend, self.slab)
        self._range_checked_at = 0
        self._check_range()
        self._bufs = None
        self._hook_keep = None
        import os as _os
        self.overlap = _os.environ.get("LSM_SLAB_OVERLAP", "1") != "0"   # boundary-first stages overlapping the halo exchange
        if any(t.update_func is not None for t in terms) and hasattr(self.backend, "cfl_cache"):
            self.backend.cfl_cache(False)   # hooks may mutate coefficients in place

    # accessors (src/levelsetequation.jl:124-162)
    def current_state(self):
        return self.state

    def current_time(self):
        return self.t

    def mesh(self):
        return self.mesh_

    def time_integrator(self):
        return self.integrator

    def _pde(self):
        return "ϕₜ + " + " + ".join(repr(t) for t in self.terms) + " = 0"

    def _show(self):
        """src/levelsetequation.jl:101-110 (text/plain)."""
        lines = ["LevelSetEquation", f"  ├─ equation: {self._pde()}", f"  ├─ time:     {_jl_float(self.t)}"]
        lines += _embed_show("integrator", show(self.integrator)) + _embed_show("state", show(self.state))
        return "\n".join(lines + ["  ╰─"])

    def __repr__(self):
        """src/levelsetequation.jl:113-117 (compact form)."""
        return f"LevelSetEquation({self._pde()}, t={_jl_float(self.t)})"

    def _check_range(self):
        """FAST arithmetic has a domain (include/lsm.h, LSM_MODE_FAST): refuse to run outside it instead of returning
        Inf/NaN silently.  Checked when the equation is built and every 64 steps (one reduction pass)."""
        if not hasattr(self.backend, "check_range"):
            return
        ok, m = self.backend.check_range(self.state.buf)
        if self.comm is not None and self.world > 1:
            # every rank raises or none does: a rank failing alone would leave the others in their next exchange
            parts = self._all_gather_object((ok, m))
            ok, m = all(p[0] for p in parts), max(p[1] for p in parts)
        if not ok:
            raise ValueError(f"max|ϕ| = {m:.3g} is outside the domain of the FAST arithmetic mode (differences between neighbouring nodes "
                             f"must stay below 1e35: max|ϕ| <= {L.FAST_MAX_ABS:g}); rescale the field or build the equation with mode=\"strict\"")

    # ---- update_term! (src/levelsetterms.jl:14,65-69,148-152) + slow-path coefficient sampling
    def _needs_hook(self):
        return any(t.update_func is not None or (t.coeff is not None and t.coeff.callable is not None) for t in self.terms)

    def _update_terms(self, field, t):
        for term in self.terms:
            if term.coeff is not None:
                term.coeff.refresh(t)
            if term.update_func is not None:
                term.update_func(term.coeff, field, t)

    # ---- compute_cfl (src/levelsetterms.jl:22-28)
    def compute_cfl(self, t=None):
        t = self.t if t is None else t
        arr = _terms_c(self.terms)
        if self.band:   # minimum over active_nodeindices (src/levelsetterms.jl:31-38)
            dt = self.backend.compute_cfl_band(arr, len(self.terms), self.state.buf, self.state.mask, t, self.state.tiles, self.state.MC)
        else:
            dt = self.backend.compute_cfl_local(arr, len(self.terms), self.state.buf, t)
        if self.comm is not None and self.world > 1:
            dt = self._allreduce_min(dt)
        if not dt > 0:
            raise ValueError(f"invalid time-step based on CFL condition: Δt = {dt} (check for NaN/Inf in velocity or speed)")
        return dt

    def _attach_library_comm(self):
        b = self.backend
        if isinstance(self.comm, _LocalRank):
            g = self.comm.group
            backs = g.exchange(self.rank, b)
            if self.rank == 0:
                type(b).comm_attach_local(backs)
                g._backends = list(backs)
            g.exchange(self.rank, None)            # nobody runs ahead of the attachment
        else:
            import torch.distributed as dist
            # a rank whose library cannot open RCCL (or whose communicator does not come up) must not leave the others
            # exchanging with nobody: the ranks agree, and the group as a whole falls back to the stage-by-stage exchange
            # over torch.distributed (same planes, same order, same results — DESIGN.md §6) with a warning
            err = None
            try:
                box = [b.comm_unique_id() if self.rank == 0 else None]
            except L.LsmError as e:
                box, err = [None], e
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.comm, 0), group=self.comm)
            if box[0] is not None:
                try:
                    b.comm_attach_rccl(box[0], self.rank, self.world)
                except L.LsmError as e:
                    err = e
            else:
                err = err or L.LsmError("rank 0 could not create an RCCL unique id")
            oks = self._all_gather_object(err is None)
            if not all(oks):
                if err is None:
                    b.comm_detach()
                import warnings
                warnings.warn(f"libhiplsm's RCCL communicator is unavailable ({err or 'on another rank'}); "
                              "exchanging ghost planes through torch.distributed instead")
                return
        self.lib_comm = True

    def _all_gather_object(self, v):
        if isinstance(self.comm, _LocalRank):
            return self.comm.group.exchange(self.rank, v)
        import torch.distributed as dist
        parts = [None] * self.world
        dist.all_gather_object(parts, v, group=self.comm)
        return parts

    def _allreduce_min(self, dt):
        if self.lib_comm:
            return self.backend.allreduce_dt(dt)   # lsm_allreduce_dt: MIN over the ranks, NaN wins
        import torch
        import torch.distributed as dist
        dev = self.state.buf.device
        # on the current stream, behind the stages of the previous step: one collective in flight at a time per rank (the
        # local Δt may come early from the library's CFL stream, but overlapping this all-reduce with the plane exchange
        # of the previous step would run two communicators concurrently — not worth the 8 bytes)
        x = torch.tensor([-1.0 if math.isnan(dt) else dt], dtype=torch.float64, device=dev)   # NaN must win
        dist.all_reduce(x, op=dist.ReduceOp.MIN, group=self.comm)
        v = float(x.item())
        return float("nan") if v < 0 else v

    # ---- _advance! (src/timestepping.jl:126-202)
    def _advance(self, tc, dt):
        b = self.backend
        if self._bufs is None:
            self._bufs = (b.alloc(), b.alloc())   # cached across integrate! calls (the reference reallocates)
        b1, b2 = self._bufs
        arr = _terms_c(self.terms)
        n = len(self.terms)
        phi = self.state.buf
        name = self.integrator.name
        if self.band:
            return self._advance_band(tc, dt, b1, b2)
        if self.comm is None or self.lib_comm:
            if self.lib_comm and self.state.ghosts_dirty:   # a slab's lsm_advance_* expects valid ghosts on entry
                b.fill_ghosts(phi, 7)
                b.halo_exchange(phi)
                self.state.ghosts_dirty = False
            hook = None
            if self._needs_hook():
                def cb(_user, stage, field_ptr, t_stage):
                    try:
                        fld = self.state if stage == 0 else ROCMeshField(b, self.mesh_, self.bcs, b1 if stage == 1 else b2)
                        self._update_terms(fld, t_stage)
                        return 0
                    except Exception as e:   # surface python errors as an aborted step
                        self._hook_error = e
                        return 1
                hook = L.StageHook(cb)
                self._hook_keep = hook
            self._hook_error = None
            try:
                b.advance_single(name, arr, n, phi, b1, b2, tc, dt, hook)
            except L.LsmError:
                if self.lib_comm:
                    b.comm_abort()      # the other ranks get LSM_ERR_COMM instead of waiting for this one
                if self._hook_error is not None:
                    raise self._hook_error
                raise
            if not self.lib_comm:
                self.state.ghosts_dirty = True      # a whole-grid lsm_advance_* leaves ϕ's ghost layers stale (include/lsm.h)
            return
        # slab mode: stage by stage with ghost-plane exchange between stages
        fld = lambda buf: ROCMeshField(b, self.mesh_, self.bcs, buf)
        if self.state.ghosts_dirty:
            self._halo(phi)
            self.state.ghosts_dirty = False
        T = lambda: _terms_c(self.terms)
        if name == "fe":
            self._update_terms(self.state, tc)
            self._stage_slab(T(), n, phi, None, b1, None, L.BASE_PSI, dt, 0.0, tc)
            b.copy_(phi, b1)                       # copy!(ϕ, dst): ghosts travel with the padded buffer
        elif name == "rk2":
            self._update_terms(self.state, tc)
            self._stage_slab(T(), n, phi, None, b1, b2, L.BASE_PSI, dt, 0.5 * dt, tc)
            self._update_terms(fld(b1), tc + dt)
            self._stage_slab(T(), n, b1, b2, phi, None, L.BASE_OTHER, 0.5 * dt, 0.0, tc + dt)
        else:
            self._update_terms(self.state, tc)
            self._stage_slab(T(), n, phi, None, b1, None, L.BASE_PSI, dt, 0.0, tc)
            self._update_terms(fld(b1), tc + dt)
            self._stage_slab(T(), n, b1, phi, b2, None, L.BASE_RK3_S2, 0.25 * dt, 0.0, tc + dt)
            self._update_terms(fld(b2), tc + 0.5 * dt)
            self._stage_slab(T(), n, b2, phi, phi, None, L.BASE_RK3_S3, (2.0 / 3) * dt, 0.0, tc + 0.5 * dt)

    def _advance_band(self, tc, dt, b1, b2):
        """_advance! on a NarrowBandMeshField: lsm_advance_band_fe/rk2/rk3 (include/lsm.h) — the same stages, looping over
        active nodes only (src/timestepping.jl:128-202 with active_nodeindices = the band); every stage input is made
        readable first (band halo by affine extrapolation, then the boundary-condition ghosts) and, on a slab, the overlap
        planes of every stage result come from their owners."""
        b, st = self.backend, self.state
        if os.environ.get("LSM_BAND_PY") == "1":      # the stage-by-stage sequence through lsm_band_prepare / lsm_stage_band (tests)
            return self._advance_band_py(tc, dt, b1, b2)
        hook = None
        if self._needs_hook():
            def cb(_user, stage, field_ptr, t_stage):
                try:
                    fld = st if stage == 0 else ROCMeshField(b, self.mesh_, self.bcs, b1 if stage == 1 else b2)
                    self._update_terms(fld, t_stage)
                    return 0
                except Exception as e:   # surface python errors as an aborted step
                    self._hook_error = e
                    return 1
            hook = L.StageHook(cb)
            self._hook_keep = hook
        self._hook_error = None
        band_c = b.band_c(st.mask, st.tiles, st.MC, st._hlist, st._hcount)
        try:
            b.advance_band(self.integrator.name, _terms_c(self.terms), len(self.terms), band_c, st.buf, b1, b2, tc, dt, hook)
        except L.LsmError:
            if self.lib_comm:
                b.comm_abort()
            if self._hook_error is not None:
                raise self._hook_error
            raise
        st.ghosts_dirty = True

    def _advance_band_py(self, tc, dt, b1, b2):
        """The same step driven stage by stage from here (what lsm_advance_band_* does inside the library)."""
        b, st = self.backend, self.state
        n = len(self.terms)
        phi = st.buf
        name = self.integrator.name
        T = lambda: _terms_c(self.terms)
        sb = lambda psi, phin, out, out2, mode, c1, c2, t: b.stage_band(T(), n, psi, phin, out, out2, mode, c1, c2, t,
                                                                      st.mask, st.tiles, st.MC)
        fld = lambda buf: ROCMeshField(b, self.mesh_, self.bcs, buf)
        slabbed = self.comm is not None and self.world > 1
        sync = (lambda buf: b.band_overlap_values(buf)) if slabbed else (lambda buf: None)   # stencil inputs: owners' values
        st.prepare(phi)
        self._update_terms(st, tc)
        if name == "fe":
            b.copy_(b1, phi)                     # copy!(dst, ϕ): non-band entries keep ϕ's (scratch) values
            sb(phi, None, b1, None, L.BASE_PSI, dt, 0.0, tc)
            b.copy_(phi, b1)
        elif name == "rk2":
            sb(phi, None, b1, b2, L.BASE_PSI, dt, 0.5 * dt, tc)
            sync(b1)
            st.prepare(b1)
            self._update_terms(fld(b1), tc + dt)
            sb(b1, b2, phi, None, L.BASE_OTHER, 0.5 * dt, 0.0, tc + dt)
        else:
            sb(phi, None, b1, None, L.BASE_PSI, dt, 0.0, tc)
            sync(b1)
            st.prepare(b1)
            self._update_terms(fld(b1), tc + dt)
            sb(b1, phi, b2, None, L.BASE_RK3_S2, 0.25 * dt, 0.0, tc + dt)
            sync(b2)
            st.prepare(b2)
            self._update_terms(fld(b2), tc + 0.5 * dt)
            sb(b2, phi, phi, None, L.BASE_RK3_S3, (2.0 / 3) * dt, 0.0, tc + 0.5 * dt)
        sync(phi)
        st.ghosts_dirty = True

    def update_band(self):
        """update_band!(ϕ) after an accepted step (src/timestepping.jl:115): no-op on a dense field."""
        if self.band:
            self.state.rebuild(from_dense=False)
            if self.comm is not None and self.world > 1:
                self._band_sync_after_update()

    def _band_sync_after_update(self):
        """Slab of a band: the band set and the values of newly active nodes are only right away from the cut faces;
        take both from the owners on the overlap planes (lsm_band_overlap_mask: whole planes of mask bytes, after which
        only the band nodes' values travel), then re-derive tiles, lists and the halo from the full mask."""
        st, b = self.state, self.backend
        b.band_overlap_mask(st.mask)
        b.band_overlap_values(st.buf)
        b.band_retile(st.mask, st.tiles, st.MC)
        b.band_status(st._hcount)
        b.band_halo(st.buf, st.mask, st.halo, st.tiles, st.MC, st._hlist, st._hcount)
        st._check_halo()

    def _stage_slab(self, arr, n, psi, phin, out, out2, mode, cdt, cdt2, t):
        """One stage of a slab followed by its ghost resolution.  With overlap, the G+1 planes next to
        each slab interface are updated and ghost-filled first, their exchange is started, and the
        interior is updated while the planes travel over xGMI (the results are identical: every node is
        computed by the same kernel from the same inputs)."""
        b = self.backend
        N = self.mesh_.ndim
        nloc = int(b.lay.n[N - 1])
        B = L.GHOST + 1                      # +1: the periodic wrap sends planes shifted by one node
        if not (self.overlap and (self.world > 1 or getattr(self, "_force_overlap", False)) and N >= 2 and nloc >= 2 * B + 1
                and hasattr(b, "stage_planes")):
            b.stage(arr, n, psi, phin, out, out2, mode, cdt, cdt2, t)
            self._halo(out)
            return
        for m0, m1 in ((0, B), (nloc - B, nloc)):
            b.stage_planes(arr, n, psi, phin, out, out2, mode, cdt, cdt2, t, m0, m1)
            b.fill_ghosts_planes(out, m0, m1)
        reqs = self._exchange_start(out)
        b.stage_planes(arr, n, psi, phin, out, out2, mode, cdt, cdt2, t, B, nloc - B)
        b.fill_ghosts_planes(out, B, nloc - B)
        for w in reqs:
            w.wait()
        b.fill_ghosts(out, 1 << (N - 1))     # physical BC ghost planes of the end ranks (slab interfaces are skipped)

    def _halo(self, buf):
        """Ghost resolution for a slab: BC fill of every dimension (slab interfaces are skipped by
        the library), then exchange of LSM_GHOST full padded planes with the neighbouring ranks.
        Because the exchanged planes carry their own dim-1..N-1 ghosts, the corner composition of
        _getindexbc (src/meshfield.jl:248-260) is preserved."""
        self.backend.fill_ghosts(buf, 7)
        for w in self._exchange_start(buf):
            w.wait()

    def _exchange_start(self, buf):
        """Post the ghost-plane sends/receives of one field; returns the pending works."""
        import torch.distributed as dist
        if self.world == 1:
            return []
        b = self.backend
        N = self.mesh_.ndim
        G = L.GHOST
        sl = int(b.lay.stride[N - 1])          # elements per padded plane
        nloc = int(b.lay.n[N - 1])
        flat = b.flat(buf)
        plane = lambda k0, k1: flat[(k0 + G) * sl:(k1 + G) * sl]   # local plane range [k0, k1)
        up = self.rank + 1 if self.rank < self.world - 1 else (0 if self.periodic_last else None)
        dn = self.rank - 1 if self.rank > 0 else (self.world - 1 if self.periodic_last else None)
        # periodic wrap has period n-1 (nodes 1 and n coincide, src/boundaryconditions.jl:107-119):
        # across the wrap the sender skips its duplicate end node.
        wrap_up = self.rank == self.world - 1
        wrap_dn = self.rank == 0
        # op order [send up, recv dn, send dn, recv up]: messages between one pair of ranks match in
        # posting order (RCCL and gloo alike), which matters when up == dn (2 ranks, periodic ring).
        ops = []
        if up is not None:
            s0 = nloc - G - (1 if wrap_up else 0)
            ops.append(dist.P2POp(dist.isend, plane(s0, s0 + G), up, group=self.comm))
        if dn is not None:
            ops.append(dist.P2POp(dist.irecv, plane(-G, 0), dn, group=self.comm))
            s0 = 1 if wrap_dn else 0
            ops.append(dist.P2POp(dist.isend, plane(s0, s0 + G), dn, group=self.comm))
        if up is not None:
            ops.append(dist.P2POp(dist.irecv, plane(nloc, nloc + G), up, group=self.comm))
        return dist.batch_isend_irecv(ops) if ops else []

    def gather_state(self):
        """Full-grid host copy of the state on every rank (tests / diagnostics)."""
        v = self.state.values()
        if self.comm is None or self.world == 1:
            return v
        v = v[..., self.own[0]:self.own[0] + self.own[1]]       # a band slab also holds copies of its neighbours' planes
        N = self.mesh_.ndim
        return np.asfortranarray(np.concatenate(self._all_gather_object(v), axis=N - 1))


def integrate_(ls, tf, dt=float("inf"), prehook=None, posthook=None):
    """integrate!(ls, tf, Δt = Inf; prehook, posthook) — src/levelsetequation.jl:194-203 and the
    step loop _integrate! of src/timestepping.jl:101-122, on the host."""
    tc = ls.current_time()
    if not tf >= tc:
        raise ValueError(f"final time {tf} must be ≥ initial time {tc}: the level-set equation cannot be solved back in time")
    alpha = ls.integrator.cfl
    while tc <= tf - _eps(tc):
        if prehook is not None:
            prehook(ls)
        ls._update_terms(ls.state, tc)
        step = _jl_min(dt, alpha * ls.compute_cfl(tc), tf - tc)
        ls._advance(tc, step)
        tc += step
        ls.t = tc
        ls._range_checked_at += 1
        if ls._range_checked_at % 64 == 0:
            ls._check_range()
        ls.update_band()   # re-tube before the posthook (no-op on a full grid) — src/timestepping.jl:115
        if posthook is not None:
            posthook(ls)
    ls.t = tf
    return ls


def _sum_over_ranks(ls, x):
    if ls.comm is None or ls.world == 1:
        return x
    if isinstance(ls.comm, _LocalRank):
        return float(sum(ls._all_gather_object(float(x))))   # rank order: the same sum on every rank
    import torch
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=ls.state.buf.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=ls.comm)
    return float(t.item())


def _band_single_device(ls, what):
    if ls.comm is not None and ls.world > 1:
        raise ValueError(f"{what} of a slab-decomposed NarrowBandMeshField is not built (the band-free grid lines are "
                         "classified by the nearest band node of the whole band)")


def volume(ls):
    """volume(eq) — measure of {ϕ ≤ 0} with the smoothed Heaviside (src/levelsetops.jl:27-33,
    src/levelsetequation.jl:165); the standard posthook diagnostic (docs/src/levelset-equation.md:142-149).
    NarrowBandMeshField states: from the band alone (src/levelsetops.jl:34-116)."""
    if getattr(ls, "band", False):
        _band_single_device(ls, "volume")
        return ls.backend.band_volume(ls.state.buf, ls.state.mask)
    return _sum_over_ranks(ls, ls.backend.volume_local(ls.state.buf))


def perimeter(ls):
    """perimeter(eq) — measure of {ϕ = 0} with the smoothed Dirac delta (src/levelsetops.jl:139-149); for a
    NarrowBandMeshField state the sum over the active nodes (:150-166)."""
    if getattr(ls, "band", False):
        _band_single_device(ls, "perimeter")
        ls.state.prepare(ls.state.buf)      # ϕ[I] off the band (extrapolation) and outside the grid (BCs) for the centred gradient
        return ls.backend.band_perimeter(ls.state.buf, ls.state.mask)
    if ls.comm is not None and ls.world > 1:
        ls._halo(ls.state.buf)          # slab interfaces: the centred gradient needs the neighbours' planes
    return _sum_over_ranks(ls, ls.backend.perimeter_local(ls.state.buf))


def extend_along_normals_(F, phi, nb_iters=50, cfl=0.45, frozen=None, interface_band=1.5, min_norm=1.0e-14):
    """extend_along_normals!(F, ϕ; nb_iters, cfl, frozen, interface_band, min_norm) —
    src/velocityextension.jl:20-76.  `F` and `phi` are device fields (ROCMeshField) on the same mesh;
    `frozen` is None (band rule) or a boolean/0-1 host array or device field.  F is updated in place."""
    if not (isinstance(F, ROCMeshField) and isinstance(phi, ROCMeshField)):
        raise ValueError("F and ϕ must be device fields (ROCMeshField) of the same equation/backend")
    if F.mesh.n != phi.mesh.n:
        raise ValueError("F and ϕ must be defined on the same mesh")
    if nb_iters < 0:
        raise ValueError("nb_iters must be non-negative")
    if not cfl > 0:
        raise ValueError("cfl must be strictly positive")
    if not interface_band >= 0:
        raise ValueError("interface_band must be non-negative")
    if not min_norm >= 0:
        raise ValueError("min_norm must be non-negative")
    b = phi.backend
    fz = None
    if frozen is not None:
        if isinstance(frozen, (ROCMeshField, SideField)):
            fz = frozen.buf if str(frozen.buf.dtype) == "torch.float64" else frozen.buf.double()   # side arrays are float64
        else:
            a = np.asarray(frozen.vals if isinstance(frozen, MeshField) else frozen)
            if a.shape != phi.mesh.n:
                raise ValueError("frozen mask must have the same size as ϕ")
            fz = getattr(b, "alloc_side", b.alloc)()
            getattr(b, "upload_side", b.upload)(fz, a.astype(np.float64))
    b.extend_along_normals(F.buf, phi.buf, fz, int(nb_iters), float(cfl), float(interface_band), float(min_norm))
    F.ghosts_dirty = True
    return F


class SideField:
    """A float64 device array in the padded layout of a backend (coefficient fields, frozen masks, the outputs of
    curvature_field / gradient_field / normal_field)."""

    def __init__(self, backend, mesh, buf=None):
        self.backend, self.mesh = backend, mesh
        self.buf = backend.alloc_side() if buf is None else buf

    def values(self):
        return self.backend.download_side(self.buf)


def _geometry(phi, what, ncomp, scale, band, fill, out, frozen_out):
    if not isinstance(phi, ROCMeshField):
        raise ValueError("ϕ must be a device field (ROCMeshField)")
    b = phi.backend
    mask = None
    if isinstance(phi, ROCNarrowBandMeshField):     # the queries work on both field types (docs/src/geometry-queries.md)
        phi.prepare(phi.buf)                        # ϕ[I] off the band (extrapolation) and outside the grid (BCs)
        mask = phi.mask
    outs = out if out is not None else [SideField(b, phi.mesh) for _ in range(ncomp)]
    if len(outs) != ncomp:
        raise ValueError(f"expected {ncomp} output field(s)")
    b.geometry(what, phi.buf, [o.buf if isinstance(o, SideField) else o for o in outs], scale=scale,
               band_width=-1.0 if band is None else float(band), fill=fill,
               frozen_out=None if frozen_out is None else (frozen_out.buf if isinstance(frozen_out, SideField) else frozen_out), mask=mask)
    phi.ghosts_dirty = False
    return outs


def curvature_field(phi, scale=1.0, band=None, fill=0.0, out=None, frozen_out=None):
    """scale·curvature(ϕ, I) (src/levelsetops.jl:197-205) at every node, on the device.  `band`: only nodes with
    |ϕ[I]| <= band are evaluated (the others get `fill`) and `frozen_out` (a SideField) marks them with 1.0 — the
    seed-and-freeze loop of the reference's speed update functions (test/test-velocityextension.jl:118-131)."""
    return _geometry(phi, L.GEOM_CURVATURE, 1, scale, band, fill, None if out is None else [out], frozen_out)[0]


def gradient_field(phi, scale=1.0):
    """gradient(ϕ, I) (src/levelsetops.jl:212-215) at every node: one SideField per dimension."""
    return _geometry(phi, L.GEOM_GRADIENT, phi.mesh.ndim, scale, None, 0.0, None, None)


def normal_field(phi, scale=1.0):
    """normal(ϕ, I) = ∇ϕ/‖∇ϕ‖ (src/levelsetops.jl:222-226) at every node: one SideField per dimension."""
    return _geometry(phi, L.GEOM_NORMAL, phi.mesh.ndim, scale, None, 0.0, None, None)


def curvature(phi, I):
    """curvature(ϕ, I) — scalar convenience (slow: evaluates the field); 0-based I."""
    return float(curvature_field(phi).values()[tuple(I)])


def gradient(phi, I):
    return np.array([float(g.values()[tuple(I)]) for g in gradient_field(phi)])


def normal(phi, I):
    return np.array([float(g.values()[tuple(I)]) for g in normal_field(phi)])


class InterpolatedField:
    """InterpolatedField(ϕ, order) (src/interpolation.jl:117-151): the piecewise polynomial interpolant of a device field
    (dense, or a narrow band near its active nodes), evaluated on the device.  `itp(x)` for one point or an (npts, ndim) array; `gradient`, `hessian`,
    `value_and_gradient`, `value_gradient_hessian` as in the reference (:228-260)."""

    def __init__(self, phi, order=3):
        if not isinstance(phi, ROCMeshField):
            raise ValueError("InterpolatedField wraps a device field (ROCMeshField / ROCNarrowBandMeshField)")
        if not 1 <= int(order) <= 5:
            raise ValueError("interpolation order must be in 1..5")
        self.phi, self.order = phi, int(order)

    def _eval(self, x, grad, hess):
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        pts = x[None, :] if single else x
        if pts.ndim != 2 or pts.shape[1] != self.phi.mesh.ndim:
            raise ValueError(f"points must have {self.phi.mesh.ndim} coordinates")
        if isinstance(self.phi, ROCNarrowBandMeshField):
            # a band field interpolates wherever the patch's stencil stays on band or halo nodes (test/test-narrow-band.jl:91-103):
            # the halo holds the extrapolated values, exactly what the reference's nb[I] returns there
            self.phi.prepare(self.phi.buf)
        v, g, H = self.phi.backend.interpolate(self.phi.buf, self.order, pts, grad, hess)
        self.phi.ghosts_dirty = False
        if single:
            return float(v[0]), (g[0] if grad else None), (H[0] if hess else None)
        return v, g, H

    def __call__(self, x):
        return self._eval(x, False, False)[0]

    def gradient(self, x):
        return self._eval(x, True, False)[1]

    def hessian(self, x):
        return self._eval(x, False, True)[2]

    def value_and_gradient(self, x):
        return self._eval(x, True, False)[:2]

    def value_gradient_hessian(self, x):
        return self._eval(x, True, True)


class NewtonSDF:
    """NewtonSDF(ϕ; order = 3, upsample = 2, maxiters = 10, xtol, ftol) (src/sdf.jl:57-78) on the device: samples the
    interface of a private copy of ϕ once; `sdf(x)` is the signed distance at a point or an (npts, ndim) array of
    points (src/sdf.jl:80-84); `get_sample_points()`; `closest_point(x)`.  ϕ: a dense or narrow-band device field."""

    def __init__(self, phi, order=3, upsample=2, maxiters=10, xtol=None, ftol=None):
        if not isinstance(phi, ROCMeshField):
            raise ValueError("NewtonSDF takes a device field (ROCMeshField / ROCNarrowBandMeshField)")
        eps = float(np.sqrt(np.finfo(np.float64).eps))
        b = phi.backend
        mask = None
        if isinstance(phi, ROCNarrowBandMeshField):
            phi.prepare(phi.buf)
            mask = phi.mask
        else:
            b.fill_ghosts(phi.buf)
            phi.ghosts_dirty = False
        self.backend, self.ndim = b, phi.mesh.ndim
        self._h, self.nsamples = b.sdf_create(phi.buf, mask, order, upsample, maxiters, eps if xtol is None else xtol, eps if ftol is None else ftol)

    def _eval(self, x, want_cp):
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        pts = x[None, :] if single else x
        if pts.ndim != 2 or pts.shape[1] != self.ndim:
            raise ValueError(f"points must have {self.ndim} coordinates")
        d, cp, nfail = self.backend.sdf_eval(self._h, pts, want_cp)
        if single:
            return float(d[0]), (cp[0] if want_cp else None), nfail
        return d, cp, nfail

    def __call__(self, x):
        return self._eval(x, False)[0]

    def closest_point(self, x):
        """(closest point(s), number of non-converged solves) — _closest_point_on_interface, src/sdf.jl:113-127"""
        _, cp, nfail = self._eval(x, True)
        return cp, nfail

    def get_sample_points(self):
        return self.backend.sdf_samples(self._h, self.nsamples)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None:
            try:
                self.backend.sdf_destroy(h)
            except Exception:
                pass


def hausdorff_distance(sdf1, sdf2):
    """hausdorff_distance(sdf₁, sdf₂) (src/sdf.jl:129-150): the larger of the two one-sided maxima over the sample points of
    one interface of the distance to the other."""
    def one_sided(a, b):
        pts = a.get_sample_points()
        cp, _ = b.closest_point(pts)
        return float(np.sqrt(((pts - cp) ** 2).sum(axis=1)).max())
    return max(one_sided(sdf1, sdf2), one_sided(sdf2, sdf1))


def reinitialize_(phi, order=3, upsample=2, maxiters=20, xtol=None, ftol=None):
    """reinitialize!(ϕ; order = 3, upsample = 2, maxiters = 20, xtol = nothing, ftol = nothing)
    (src/reinitializer.jl:12-42): overwrite every active node of ϕ with its signed distance to the interface,
    by the Newton closest-point method on the piecewise-polynomial interpolant of ϕ (NewtonSDF, src/sdf.jl).
    `phi` is a device field (or an equation: its current state).  Returns ϕ; warns, like the reference, when
    the closest-point solver did not converge at some nodes."""
    import warnings
    eq = phi if isinstance(phi, LevelSetEquation) else None
    if eq is not None:
        phi = eq.current_state()
    if not isinstance(phi, ROCMeshField):
        raise TypeError("reinitialize_ expects a device field (ROCMeshField / ROCNarrowBandMeshField) or a LevelSetEquation")
    if phi.bcs is None:
        raise ValueError("the field needs boundary conditions: the interpolation stencils reach outside the grid")
    b = phi.backend
    eps = float(np.finfo(getattr(b, "dtype", np.dtype(np.float64))).eps)
    xtol = math.sqrt(eps) if xtol is None else float(xtol)      # sqrt(eps(T)), T = float(valtype(ϕ)) — src/sdf.jl:66-67
    ftol = math.sqrt(eps) if ftol is None else float(ftol)
    band = isinstance(phi, ROCNarrowBandMeshField)
    if band:
        phi.prepare(phi.buf)              # stencil nodes off the band: the affine extrapolant, then the BC ghosts
    else:
        b.fill_ghosts(phi.buf, 7)
    ncand, nfail, nfar = b.reinitialize(phi.buf, phi.mask if band else None, order, upsample, maxiters, xtol, ftol)
    phi.ghosts_dirty = True
    if eq is not None and eq.comm is not None and eq.world > 1:
        if not band:
            raise ValueError("reinitialize! of a slab-decomposed dense field is not supported")
        eq.backend.band_overlap_values(phi.buf)     # the neighbours' planes: their owners' values
        nfail = nfar = 0                 # counted on the extended slab, cut faces included: not meaningful per rank
    if nfar:
        warnings.warn(f"reinitialize!: no interface sample was found ({nfar} nodes left unchanged)")
    if nfail:
        n_active = phi.active_count() if band else int(np.prod(phi.mesh.n))
        warnings.warn(f"reinitialize!: closest-point solver did not converge for {nfail} / {n_active} points")
    return phi


def current_state(ls):
    return ls.current_state()


def current_time(ls):
    return ls.current_time()

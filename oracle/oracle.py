"""ctypes front-end of the CPU ORACLE (oracle/lsm_oracle.c) — test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It deliberately shares nothing with the product package except the POD layouts of include/lsm.h.

Arrays are numpy float64 in Fortran (column-major) order with shape (n1[, n2[, n3]]), i.e. the
memory layout of the reference's `Array{Float64,N}` (src/meshfield.jl:209).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSM_ORACLE_LIB: another build of the same source (the sanitizer build of `make -C oracle asan`, tools/oracle_asan.sh)
_LIB_PATH = os.environ.get("LSM_ORACLE_LIB") or os.path.join(_HERE, "liblsm_oracle.so")

GHOST = 3
BC_PERIODIC, BC_EXTRAPOLATION, BC_SYMMETRY, BC_NONE = 0, 1, 2, 3
TERM_ADVECTION, TERM_NORMAL_MOTION, TERM_CURVATURE, TERM_EIKONAL = 0, 1, 2, 3
SCHEME_UPWIND, SCHEME_WENO5 = 0, 1
COEFF_CONST, COEFF_ROTATION, COEFF_SEPARABLE, COEFF_FIELD = 0, 1, 2, 3
TIME_ONE, TIME_COS = 0, 1
BASE_PSI, BASE_RK3_S2, BASE_RK3_S3, BASE_OTHER = 0, 1, 2, 3
FE, RK2, RK3 = 0, 1, 2


class LsmGrid(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("_pad", C.c_int32), ("n", C.c_int64 * 3), ("lc", C.c_double * 3),
                ("hc", C.c_double * 3)]


class LsmBc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("degree", C.c_int32)]


class LsmSlab(C.Structure):
    _fields_ = [("lo", C.c_int64), ("n", C.c_int64)]


class LsmLayout(C.Structure):
    _fields_ = [("n", C.c_int64 * 3), ("g", C.c_int64 * 3), ("stride", C.c_int64 * 3), ("origin", C.c_int64),
                ("total", C.c_int64)]


class LsmCoeff(C.Structure):
    _fields_ = [("kind", C.c_int32), ("time_kind", C.c_int32), ("time_param", C.c_double), ("value", C.c_double * 4),
                ("field", C.c_void_p * 3), ("sep", C.c_void_p * 3)]


class LsmTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("scheme", C.c_int32), ("coeff", LsmCoeff), ("s0", C.c_void_p)]


BcArray = (LsmBc * 2) * 3


def build():
    """Compile the oracle with gcc (oracle/Makefile)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
                os.path.join(_HERE, "lsm_oracle.c")):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        i3 = C.POINTER(C.c_int64)
        L.orc_get.restype = C.c_double
        L.orc_get.argtypes = [C.POINTER(LsmGrid), BcArray, C.c_int, dp, i3]
        L.orc_deriv.restype = C.c_double
        L.orc_deriv.argtypes = [C.POINTER(LsmGrid), BcArray, C.c_int, dp, C.c_int, i3, C.c_int, C.c_int]
        L.orc_weno5_core.restype = C.c_double
        L.orc_weno5_core.argtypes = [C.c_double] * 5
        L.orc_term.restype = C.c_double
        L.orc_term.argtypes = [C.POINTER(LsmGrid), BcArray, dp, C.POINTER(LsmTerm), i3, C.c_double]
        L.orc_eikonal_sign.restype = None
        L.orc_eikonal_sign.argtypes = [C.POINTER(LsmGrid), dp, dp]
        L.orc_compute_cfl.restype = C.c_double
        L.orc_compute_cfl.argtypes = [C.POINTER(LsmGrid), BcArray, dp, C.POINTER(LsmTerm), C.c_int, C.c_double]
        L.orc_advance.restype = None
        L.orc_advance.argtypes = [C.c_int, C.POINTER(LsmGrid), BcArray, dp, dp, dp, C.POINTER(LsmTerm), C.c_int,
                                  C.c_double, C.c_double]
        L.orc_integrate.restype = C.c_int64
        L.orc_integrate.argtypes = [C.c_int, C.c_double, C.POINTER(LsmGrid), BcArray, dp, C.POINTER(LsmTerm), C.c_int,
                                    C.c_double, C.c_double, C.c_double, C.c_int64, dp, dp]
        L.orc_layout.restype = None
        L.orc_layout.argtypes = [C.POINTER(LsmGrid), C.POINTER(LsmSlab), C.POINTER(LsmLayout)]
        L.orc_fill_ghosts_padded.restype = None
        L.orc_fill_ghosts_padded.argtypes = [C.POINTER(LsmGrid), BcArray, C.POINTER(LsmSlab), C.POINTER(LsmLayout), dp,
                                             C.c_int]
        L.orc_stage_padded.restype = None
        L.orc_stage_padded.argtypes = [C.POINTER(LsmGrid), BcArray, C.POINTER(LsmSlab), C.POINTER(LsmLayout),
                                       C.POINTER(LsmTerm), C.c_int, dp, dp, dp, dp, C.c_int, C.c_double, C.c_double,
                                       C.c_double]
        L.orc_cfl_padded.restype = C.c_double
        L.orc_cfl_padded.argtypes = [C.POINTER(LsmGrid), BcArray, C.POINTER(LsmSlab), C.POINTER(LsmLayout),
                                     C.POINTER(LsmTerm), C.c_int, dp, C.c_double]
        L.orc_extend_along_normals.restype = None
        L.orc_extend_along_normals.argtypes = [C.POINTER(LsmGrid), BcArray, dp, dp, dp, C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_measure.restype = C.c_double
        L.orc_measure.argtypes = [C.c_int, C.POINTER(LsmGrid), BcArray, dp]
        L.orc_geometry.restype = None
        L.orc_geometry.argtypes = [C.c_int, C.POINTER(LsmGrid), BcArray, dp, dp, dp, dp]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def set_threads(n):
    lib().orc_set_threads(int(n))


def max_threads():
    return lib().orc_max_threads()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _i3(I):
    I = list(I) + [0] * (3 - len(I))
    return (C.c_int64 * 3)(*I)


class Grid:
    """CartesianGrid(lc, hc, n) — src/meshes.jl:34-42."""

    def __init__(self, lc, hc, n):
        lc, hc, n = list(lc), list(hc), list(n)
        assert len(lc) == len(hc) == len(n)
        self.ndim = len(n)
        self.lc = [float(x) for x in lc]
        self.hc = [float(x) for x in hc]
        self.n = [int(x) for x in n]
        self.c = LsmGrid()
        self.c.ndim = self.ndim
        for d in range(3):
            self.c.n[d] = self.n[d] if d < self.ndim else 1
            self.c.lc[d] = self.lc[d] if d < self.ndim else 0.0
            self.c.hc[d] = self.hc[d] if d < self.ndim else 1.0

    @property
    def shape(self):
        return tuple(self.n)

    def meshsize(self, d=None):
        """(hc - lc)/(n - 1) — src/meshes.jl:109-110."""
        hs = [(self.hc[k] - self.lc[k]) / (self.n[k] - 1) for k in range(self.ndim)]
        return hs if d is None else hs[d]

    def node(self, I):
        """lc + (I-1)*h with 0-based I — src/meshes.jl:114-117."""
        return [self.lc[d] + float(I[d]) * self.meshsize(d) for d in range(self.ndim)]

    def coords(self):
        """Per-axis coordinate vectors, lc + i*h."""
        return [self.lc[d] + np.arange(self.n[d], dtype=np.float64) * self.meshsize(d) for d in range(self.ndim)]

    def sample(self, f):
        """MeshField(f, grid): f receives broadcastable coordinate arrays — src/meshfield.jl:208-211."""
        xs = np.meshgrid(*self.coords(), indexing="ij", sparse=True)
        return np.asfortranarray(np.broadcast_to(f(*xs), self.shape).astype(np.float64))


def make_bc(spec, ndim):
    """_normalize_bc — src/boundaryconditions.jl:166-188.

    spec: a single bc, a per-dimension list, or per-dimension (left, right) pairs, where a bc is
    'periodic' | 'neumann' | 'linear' | 'symmetry' | ('extrapolation', P) | 'none'."""

    def one(b):
        if isinstance(b, str):
            b = {"periodic": (BC_PERIODIC, 0), "neumann": (BC_EXTRAPOLATION, 0), "linear": (BC_EXTRAPOLATION, 1),
                 "symmetry": (BC_SYMMETRY, 0), "none": (BC_NONE, 0)}[b]
        elif b[0] == "extrapolation":
            b = (BC_EXTRAPOLATION, int(b[1]))
        return b

    def is_single(b):
        return isinstance(b, str) or (isinstance(b, tuple) and len(b) == 2 and (b[0] == "extrapolation" or isinstance(b[0], int)))

    if is_single(spec):
        pairs = [(one(spec), one(spec))] * ndim
    else:
        if len(spec) != ndim:
            raise ValueError("invalid number of boundary conditions")
        pairs = []
        for d, b in enumerate(spec):
            if is_single(b):
                pairs.append((one(b), one(b)))
            else:
                if len(b) != 2:
                    raise ValueError(f"invalid boundary condition for dimension {d + 1}")
                l, r = one(b[0]), one(b[1])
                if (l[0] == BC_PERIODIC) != (r[0] == BC_PERIODIC):
                    raise ValueError(f"periodic boundary conditions cannot be mixed with others in dimension {d + 1}")
                pairs.append((l, r))
    arr = BcArray()
    for d in range(3):
        for s in range(2):
            k, p = pairs[d][s] if d < ndim else (BC_EXTRAPOLATION, 0)
            arr[d][s].kind = k
            arr[d][s].degree = p
    return arr


class Coeff:
    def __init__(self, kind, value=(), fields=(), sep=(), time_kind=TIME_ONE, time_param=1.0):
        self.c = LsmCoeff()
        self.c.kind = kind
        self.c.time_kind = time_kind
        self.c.time_param = time_param
        for i, v in enumerate(value):
            self.c.value[i] = float(v)
        self._keep = []
        for i, a in enumerate(fields):
            a = np.asfortranarray(a, dtype=np.float64)
            self._keep.append(a)
            self.c.field[i] = a.ctypes.data
        for i, a in enumerate(sep):
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._keep.append(a)
            self.c.sep[i] = a.ctypes.data


def const(*values):
    return Coeff(COEFF_CONST, value=values)


def rotation(w=1.0, c1=0.0, c2=0.0):
    return Coeff(COEFF_ROTATION, value=(w, c1, c2))


def separable(tables, time_kind=TIME_ONE, time_param=1.0):
    """tables[c] = list of per-axis 1-D arrays for component c."""
    return Coeff(COEFF_SEPARABLE, sep=[np.concatenate([np.asarray(t, dtype=np.float64) for t in comp]) for comp in tables],
                 time_kind=time_kind, time_param=time_param)


def field(*arrays):
    return Coeff(COEFF_FIELD, fields=arrays)


class Term:
    def __init__(self, kind, coeff=None, scheme=SCHEME_WENO5, s0=None):
        self.kind, self.coeff, self.scheme = kind, coeff, scheme
        self.s0 = None if s0 is None else np.asfortranarray(s0, dtype=np.float64)

    def fill(self, t):
        t.kind = self.kind
        t.scheme = self.scheme
        if self.coeff is not None:
            C.memmove(C.byref(t.coeff), C.byref(self.coeff.c), C.sizeof(LsmCoeff))
        t.s0 = self.s0.ctypes.data if self.s0 is not None else None


def advection(coeff, scheme=SCHEME_WENO5):
    return Term(TERM_ADVECTION, coeff, scheme)


def normal_motion(coeff):
    return Term(TERM_NORMAL_MOTION, coeff)


def curvature(coeff):
    return Term(TERM_CURVATURE, coeff)


def eikonal(s0=None):
    return Term(TERM_EIKONAL, None, s0=s0)


def term_array(terms):
    arr = (LsmTerm * max(1, len(terms)))()
    for i, t in enumerate(terms):
        t.fill(arr[i])
    return arr


# ------------------------------------------------------------------ dense (reference-layout) API

def get(grid, bc, v, I):
    """ϕ[I] with ghost resolution (0-based I, may be out of grid) — src/meshfield.jl:213-260."""
    return lib().orc_get(C.byref(grid.c), bc if bc is not None else BcArray(), 1 if bc is not None else 0, _dp(v), _i3(I))


_DERIV = {"D0": 0, "Dp": 1, "Dm": 2, "weno5m": 3, "weno5p": 4, "D20": 5, "D2": 6, "D2pp": 7, "D2mm": 8}


def deriv(grid, bc, v, which, I, dim, dim2=0):
    return lib().orc_deriv(C.byref(grid.c), bc if bc is not None else BcArray(), 1 if bc is not None else 0, _dp(v),
                           _DERIV[which], _i3(I), dim, dim2)


def weno5_core(v1, v2, v3, v4, v5):
    return lib().orc_weno5_core(v1, v2, v3, v4, v5)


def term_value(grid, bc, v, term, I, t=0.0):
    arr = term_array([term])
    return lib().orc_term(C.byref(grid.c), bc, _dp(v), arr, _i3(I), t)


def eikonal_sign(grid, v):
    s0 = np.empty_like(v, order="F")
    lib().orc_eikonal_sign(C.byref(grid.c), _dp(v), _dp(s0))
    return s0


def extend_along_normals(grid, bc, F, phi, nb_iters=50, cfl=0.45, frozen=None, interface_band=1.5, min_norm=1.0e-14):
    """extend_along_normals!(F, ϕ; ...) in place on F — src/velocityextension.jl:20-67."""
    fz = None if frozen is None else np.asfortranarray(np.asarray(frozen, dtype=np.float64))
    lib().orc_extend_along_normals(C.byref(grid.c), bc, _dp(F), _dp(phi), _dp(fz), nb_iters, cfl, interface_band, min_norm)
    return F


def volume(grid, v):
    """volume(ϕ) — src/levelsetops.jl:27-33."""
    return lib().orc_measure(0, C.byref(grid.c), make_bc("linear", grid.ndim), _dp(v))


def perimeter(grid, v, bc=None):
    """perimeter(ϕ) — src/levelsetops.jl:139-149 (LinearExtrapolationBC when the field has none)."""
    return lib().orc_measure(1, C.byref(grid.c), bc if bc is not None else make_bc("linear", grid.ndim), _dp(v))


def geometry(grid, bc, v, what):
    """what = 'curvature' | 'gradient' | 'normal' at every node — src/levelsetops.jl:197-226.
    Returns one dense array (curvature) or a list of ndim arrays."""
    k = {"curvature": 0, "gradient": 1, "normal": 2}[what]
    outs = [np.zeros(grid.n, dtype=np.float64, order="F") for _ in range(1 if k == 0 else grid.ndim)]
    ptr = [_dp(o) for o in outs] + [None] * (3 - len(outs))
    lib().orc_geometry(k, C.byref(grid.c), bc, _dp(np.asfortranarray(v)), ptr[0], ptr[1], ptr[2])
    return outs[0] if k == 0 else outs


def compute_cfl(grid, bc, v, terms, t=0.0):
    """Raw minimum over terms and nodes; raises like src/levelsetterms.jl:26 when not > 0."""
    arr = term_array(terms)
    dt = lib().orc_compute_cfl(C.byref(grid.c), bc, _dp(v), arr, len(terms), t)
    if not dt > 0:
        raise ValueError(f"invalid time-step based on CFL condition: Δt = {dt} (check for NaN/Inf in velocity or speed)")
    return dt


def advance(integrator, grid, bc, phi, terms, tc, dt, bufs=None):
    """_advance! in place on phi — src/timestepping.jl:128-202."""
    if bufs is None:
        bufs = (phi.copy(order="F"), phi.copy(order="F"))
    arr = term_array(terms)
    lib().orc_advance(integrator, C.byref(grid.c), bc, _dp(phi), _dp(bufs[0]), _dp(bufs[1]), arr, len(terms), tc, dt)
    return phi


def integrate(integrator, grid, bc, phi, terms, tf, t0=0.0, cfl=0.5, dt_max=float("inf"), max_steps=-1):
    """_integrate! without hooks — src/timestepping.jl:101-122. Returns (steps, t, last_dt)."""
    if tf < t0:
        raise ValueError(f"final time {tf} must be ≥ initial time {t0}: the level-set equation cannot be solved back in time")
    arr = term_array(terms)
    t_out = C.c_double(0.0)
    last = C.c_double(0.0)
    steps = lib().orc_integrate(integrator, cfl, C.byref(grid.c), bc, _dp(phi), arr, len(terms), t0, tf, dt_max,
                                max_steps, C.byref(t_out), C.byref(last))
    if steps < 0:
        raise ValueError("invalid time-step based on CFL condition (check for NaN/Inf in velocity or speed)")
    return steps, t_out.value, last.value


# ------------------------------------------------------------------ padded-layout API

def layout(grid, slab=None):
    lay = LsmLayout()
    s = None
    if slab is not None:
        s = LsmSlab(slab[0], slab[1])
    lib().orc_layout(C.byref(grid.c), C.byref(s) if s is not None else None, C.byref(lay))
    return lay


def padded_shape(lay, ndim):
    return tuple(int(lay.n[d] + 2 * lay.g[d]) for d in range(ndim))


def to_padded(lay, ndim, dense):
    """Dense local interior -> padded array (ghosts NaN)."""
    p = np.full(padded_shape(lay, ndim), np.nan, dtype=np.float64, order="F")
    sl = tuple(slice(int(lay.g[d]), int(lay.g[d] + lay.n[d])) for d in range(ndim))
    p[sl] = dense
    return p


def from_padded(lay, ndim, p):
    sl = tuple(slice(int(lay.g[d]), int(lay.g[d] + lay.n[d])) for d in range(ndim))
    return np.asfortranarray(p[sl])


def _slab(slab):
    return C.byref(LsmSlab(slab[0], slab[1])) if slab is not None else None


def fill_ghosts_padded(grid, bc, lay, p, slab=None, dim_mask=7):
    lib().orc_fill_ghosts_padded(C.byref(grid.c), bc, _slab(slab), C.byref(lay), _dp(p), dim_mask)
    return p


def stage_padded(grid, bc, lay, terms, psi, phin, out, out2, base_mode, cdt, cdt2, t, slab=None):
    arr = term_array(terms)
    lib().orc_stage_padded(C.byref(grid.c), bc, _slab(slab), C.byref(lay), arr, len(terms), _dp(psi), _dp(phin), _dp(out),
                           _dp(out2), base_mode, cdt, cdt2, t)
    return out


def cfl_padded(grid, bc, lay, terms, phi, t, slab=None):
    arr = term_array(terms)
    return lib().orc_cfl_padded(C.byref(grid.c), bc, _slab(slab), C.byref(lay), arr, len(terms), _dp(phi), t)

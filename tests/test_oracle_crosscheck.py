"""Second, independent restatement (numpy, vectorised) of parts of the path, compared bit for bit
with the C oracle — two implementations written from the same reference text
(src/derivatives.jl:61-121, src/levelsetterms.jl:73-82,156-170,234-265, src/timestepping.jl:170-202) —
plus the proof that filling ghost layers dimension 1 → N reproduces the _getindexbc recursion
(src/meshfield.jl:248-260) exactly, corners included.
"""
import numpy as np
import pytest


def _np_weno5(v1, v2, v3, v4, v5):
    d1 = (1 / 3) * v1 - (7 / 6) * v2 + (11 / 6) * v3
    d2 = -(1 / 6) * v2 + (5 / 6) * v3 + (1 / 3) * v4
    d3 = (1 / 3) * v3 + (5 / 6) * v4 - (1 / 6) * v5
    S1 = (13 / 12) * (v1 - 2 * v2 + v3) ** 2 + (1 / 4) * (v1 - 4 * v2 + 3 * v3) ** 2
    S2 = (13 / 12) * (v2 - 2 * v3 + v4) ** 2 + (1 / 4) * (v2 - v4) ** 2
    S3 = (13 / 12) * (v3 - 2 * v4 + v5) ** 2 + (1 / 4) * (3 * v3 - 4 * v4 + v5) ** 2
    m = np.maximum(np.maximum(np.maximum(np.maximum(v1 * v1, v2 * v2), v3 * v3), v4 * v4), v5 * v5)
    eps = 1.0e-6 * m + 1.0e-99
    a1 = 0.1 / ((S1 + eps) * (S1 + eps))
    a2 = 0.6 / ((S2 + eps) * (S2 + eps))
    a3 = 0.3 / ((S3 + eps) * (S3 + eps))
    w1 = a1 / (a1 + a2 + a3)
    w2 = a2 / (a1 + a2 + a3)
    w3 = a3 / (a1 + a2 + a3)
    return w1 * d1 + w2 * d2 + w3 * d3


def _periodic_pad(a, g):
    """ghost i<0 -> n-1+i ; i>n-1 -> i-n+1 (period n-1), every dimension."""
    for ax in range(a.ndim):
        n = a.shape[ax]
        idx = [(n - 1 + i) if i < 0 else (i - n + 1 if i > n - 1 else i) for i in range(-g, n + g)]
        a = np.take(a, idx, axis=ax)
    return a


def _np_advection_term(phi, hs, us, g=3):
    """Σ_d u_d * (u_d>0 ? weno5⁻ : weno5⁺) for constant u, periodic BCs."""
    P = _periodic_pad(phi, g)
    out = None
    for d, (h, u) in enumerate(zip(hs, us)):
        def sh(k):  # phi[I + k e_d] on the interior
            sl = [slice(g, g + n) for n in phi.shape]
            sl[d] = slice(g + k, g + k + phi.shape[d])
            return P[tuple(sl)]
        Dm = lambda k: (sh(k) - sh(k - 1)) / h   # D⁻ at I+k
        Dp = lambda k: (sh(k + 1) - sh(k)) / h   # D⁺ at I+k
        der = _np_weno5(Dm(-2), Dm(-1), Dm(0), Dm(1), Dm(2)) if u > 0 else _np_weno5(Dp(2), Dp(1), Dp(0), Dp(-1), Dp(-2))
        c = u * der
        out = c if out is None else out + c
    return out


@pytest.mark.parametrize("shape,us", [((37,), (0.7,)), ((21, 17), (1.0, -0.5)), ((12, 11, 10), (-0.3, 0.0, 0.9))])
def test_numpy_rk3_advection_matches_c_oracle_bitwise(orc, shape, us):
    rng = np.random.default_rng(1)
    nd = len(shape)
    grid = orc.Grid((-1.0,) * nd, (1.0,) * nd, shape)
    phi = np.asfortranarray(rng.standard_normal(shape))
    bc = orc.make_bc("periodic", nd)
    hs = grid.meshsize()
    dt = 0.01
    L = lambda a: _np_advection_term(a, hs, us)
    b1 = phi - dt * L(phi)
    b2 = 0.75 * phi + 0.25 * b1
    b2 = b2 - (0.25 * dt) * L(b1)
    b3 = (phi + 2 * b2) / 3
    b3 = b3 - ((2 / 3) * dt) * L(b2)
    got = phi.copy(order="F")
    orc.advance(orc.RK3, grid, bc, got, [orc.advection(orc.const(*us))], 0.0, dt)
    assert np.array_equal(got, b3)


def test_weno5_core_matches_numpy(orc):
    rng = np.random.default_rng(2)
    for _ in range(200):
        v = rng.standard_normal(5) * 10.0 ** rng.integers(-8, 8)
        assert orc.weno5_core(*v) == float(_np_weno5(*[np.float64(x) for x in v]))
    assert orc.weno5_core(0, 0, 0, 0, 0) == 0.0  # flat field: ε floor 1e-99 keeps the weights finite


BCS = ["periodic", "neumann", "linear", "symmetry", ("extrapolation", 2), ("extrapolation", 4),
       [("neumann", ("extrapolation", 3)), "periodic", ("symmetry", "linear")]]


@pytest.mark.parametrize("bcspec", BCS, ids=[str(b) for b in BCS])
@pytest.mark.parametrize("shape", [(9,), (8, 7), (7, 6, 8)])
def test_dimension_ordered_ghost_fill_equals_recursion(orc, bcspec, shape):
    nd = len(shape)
    if isinstance(bcspec, list):
        bcspec = bcspec[:nd]
    rng = np.random.default_rng(3)
    grid = orc.Grid((0.0,) * nd, (1.0,) * nd, shape)
    phi = np.asfortranarray(rng.standard_normal(shape))
    bc = orc.make_bc(bcspec, nd)
    lay = orc.layout(grid)
    p = orc.to_padded(lay, nd, phi)
    orc.fill_ghosts_padded(grid, bc, lay, p)
    assert not np.isnan(p).any()
    g = orc.GHOST
    for I in np.ndindex(*p.shape):
        J = tuple(i - g for i in I)
        assert p[I] == orc.get(grid, bc, phi, J), (I, J)


def test_padded_stage_equals_dense_advance(orc):
    """orc_stage_padded (the lsm_stage contract) chained as RK3 == the literal _advance!."""
    rng = np.random.default_rng(4)
    shape = (10, 9, 8)
    grid = orc.Grid((-1.0,) * 3, (1.0,) * 3, shape)
    bc = orc.make_bc([("extrapolation", 2), "neumann", "periodic"], 3)
    phi = np.asfortranarray(rng.standard_normal(shape))
    terms = [orc.advection(orc.rotation(1.0)), orc.eikonal(), orc.normal_motion(orc.const(0.3)),
             orc.curvature(orc.const(-0.05))]
    dt, tc = 1e-3, 0.2
    ref = phi.copy(order="F")
    orc.advance(orc.RK3, grid, bc, ref, terms, tc, dt)
    lay = orc.layout(grid)
    P = orc.fill_ghosts_padded(grid, bc, lay, orc.to_padded(lay, 3, phi))
    b1 = np.full_like(P, np.nan)
    b2 = np.full_like(P, np.nan)
    orc.stage_padded(grid, bc, lay, terms, P, None, b1, None, orc.BASE_PSI, dt, 0.0, tc)
    orc.fill_ghosts_padded(grid, bc, lay, b1)
    orc.stage_padded(grid, bc, lay, terms, b1, P, b2, None, orc.BASE_RK3_S2, 0.25 * dt, 0.0, tc + dt)
    orc.fill_ghosts_padded(grid, bc, lay, b2)
    orc.stage_padded(grid, bc, lay, terms, b2, P, P, None, orc.BASE_RK3_S3, (2 / 3) * dt, 0.0, tc + 0.5 * dt)
    assert np.array_equal(orc.from_padded(lay, 3, P), ref)


def test_oracle_under_address_and_ub_sanitizers():
    """the CPU restatement runs clean under ASan + UBSan (`make -C oracle asan`; tools/oracle_asan.sh runs the whole
    oracle-only suite against that build).  Here: a fresh interpreter with the sanitizer runtime preloaded takes one RK3 step of the
    headline equation with partial tiles and one ghost fill with every boundary-condition kind — any out-of-bounds read of the
    padded array or signed overflow in the index arithmetic aborts the child."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "asan"])
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    code = (
        "import numpy as np\n"
        "from oracle import oracle as orc\n"
        "n = (13, 9, 11)\n"
        "g = orc.Grid((0, 0, 0), (1, 1, 1), n)\n"
        "x, y, z = g.coords()\n"
        "s2 = lambda a: np.sin(np.pi * a) ** 2\n"
        "s = lambda a: np.sin(2 * np.pi * a)\n"
        "terms = [orc.advection(orc.separable([[2 * s2(x), s(y), s(z)], [-s(x), s2(y), s(z)], [-s(x), s(y), s2(z)]], orc.TIME_COS, 3.0)), orc.eikonal()]\n"
        "phi = g.sample(lambda X, Y, Z: np.sqrt((X - 0.35) ** 2 + (Y - 0.35) ** 2 + (Z - 0.35) ** 2) - 0.15)\n"
        "for kind in ('neumann', 'periodic'):\n"
        "    bc = orc.make_bc(kind, 3)\n"
        "    p = phi.copy(order='F')\n"
        "    dt = 0.5 * orc.compute_cfl(g, bc, p, terms, 0.0)\n"
        "    orc.advance(orc.RK3, g, bc, p, terms, 0.0, dt)\n"
        "    assert np.isfinite(p).all()\n"
        "print('ok')\n")
    env = dict(os.environ, LSM_ORACLE_LIB=os.path.join(root, "oracle", "liblsm_oracle_asan.so"), LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1", PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]

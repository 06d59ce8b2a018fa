"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars:
  * STRICT arithmetic mode: bit-for-bit for ghosts, CFL, advection/upwind/WENO5, NormalMotion and
    Eikonal stages; ≤ 1e-13·max|ϕ| for curvature (pow / dot order are tolerance-level in the reference).
  * FAST mode (the measured one): ≤ 1e-13·max|ϕ| per stage, ≤ 1e-10·max|ϕ| after 100 RK3 steps.
  * Δt from the CFL reduction: bitwise (min is exact).
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_STAGE = 1e-13
TOL_100 = 1e-10


@pytest.fixture(scope="module")
def hip():
    import _hip
    return _hip


def _rand_field(shape, seed=0, smooth=True):
    rng = np.random.default_rng(seed)
    nd = len(shape)
    xs = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij", sparse=True)
    r = np.sqrt(sum(x * x for x in xs))
    f = r - 0.5 + 0.05 * np.sin(3 * sum((i + 1) * x for i, x in enumerate(xs)))
    if not smooth:
        f = f + 0.02 * rng.standard_normal(shape)
    return np.asfortranarray(f)


BCS = ["periodic", "neumann", ("extrapolation", 2), "symmetry",
       [("neumann", ("extrapolation", 3)), "periodic", ("symmetry", "linear")]]


@pytest.mark.parametrize("shape", [(37,), (23, 19), (13, 11, 9), (300, 50), (270, 45, 5)])   # rows longer than a block, more than 42 rows
@pytest.mark.parametrize("bcspec", BCS, ids=[str(b) for b in BCS])
def test_ghost_fill_bitwise(hip, orc, shape, bcspec):
    nd = len(shape)
    if isinstance(bcspec, list):
        bcspec = bcspec[:nd]
    c = hip.Case(shape, bcspec)
    phi = _rand_field(shape, 1, smooth=False)
    # signed zeros on the faces: the reference's `acc = 0; acc += w·value` turns a copied -0.0 into +0.0 — bit patterns below
    phi[(0,) * nd] = -0.0
    phi[tuple(n - 1 for n in shape)] = -0.0
    phi[(0,) + tuple(n // 2 for n in shape[1:])] = -0.0
    want = c.pad(phi)
    t = c.to_dev(np.nan_to_num(c.pad(phi, fill=False), nan=-7.0))
    c.be.fill_ghosts(t)
    got = c.to_host(t)
    assert np.array_equal(got, want)
    assert np.array_equal(np.ascontiguousarray(got).view(np.int64), np.ascontiguousarray(want).view(np.int64))


SINGLE = {
    "adv_weno_const": [("adv", ("const", (0.7, -0.4, 0.9)), "weno5")],
    "adv_upwind_rot": [("adv", ("rot", 1.0, 0.0, 0.0), "upwind")],
    "adv_weno_rot": [("adv", ("rot", 1.3, 0.1, -0.2), "weno5")],
    "nm_const": [("nm", ("const", (0.6,)))],
    "nm_neg": [("nm", ("const", (-0.6,)))],
    "eik_current": [("eik", None)],
    "eik_frozen": [("eik", "phi0")],
    "curv_const": [("curv", ("const", (-0.1,)))],
}
FUSED = {
    "adv+eik": [("adv", ("rot", 1.0, 0.0, 0.0), "weno5"), ("eik", None)],
    "eik+adv": [("eik", None), ("adv", ("const", (0.5, 0.25, -1.0)), "weno5")],
    "adv+eikfrozen": [("adv", ("const", (0.5, 0.25, -1.0)), "weno5"), ("eik", "phi0")],
    "nm+curv": [("nm", ("const", (0.1,))), ("curv", ("const", (-0.1,)))],
    "adv+curv": [("adv", ("const", (0.5, 0.25, -1.0)), "weno5"), ("curv", ("const", (-0.05,)))],
    "adv+nm": [("adv", ("const", (0.5, 0.25, -1.0)), "weno5"), ("nm", ("const", (0.3,)))],
    "all4_multipass": [("adv", ("rot", 1.0, 0.0, 0.0), "weno5"), ("eik", None), ("nm", ("const", (0.3,))),
                       ("curv", ("const", (-0.05,)))],
    "two_adv_multipass": [("adv", ("const", (0.5, 0.25, -1.0)), "weno5"), ("adv", ("const", (0.1, 0.2, 0.3)), "upwind")],
}


def _fix_specs(specs, nd, phi):
    out = []
    for s in specs:
        if s[0] == "adv" and s[1][0] == "const":
            s = ("adv", ("const", s[1][1][:nd]), s[2])
        if s[0] == "adv" and s[1][0] == "rot" and nd == 1:
            s = ("adv", ("const", (0.8,)), s[2])
        if s[0] == "eik" and isinstance(s[1], str):
            s = ("eik", phi)
        out.append(s)
    return out


def _run_stage(c, orc, specs, phi, base_mode, with_out2=False, t=0.3):
    """One lsm_stage on the GPU and the same contract evaluated by the oracle."""
    from lsm_amd import _lib as L
    nd = c.nd
    ot, arr = c.terms(specs)
    psi = c.pad(phi)
    phin = c.pad(np.asfortranarray(phi * 0.9 + 0.01))
    cdt, cdt2 = 1.7e-3, 0.85e-3
    want = np.full_like(psi, np.nan)
    want2 = np.full_like(psi, np.nan) if with_out2 else None
    orc.stage_padded(c.grid, c.bc, c.olay, ot, psi, phin, want, want2, base_mode, cdt, cdt2, t)
    d_psi, d_phin = c.to_dev(psi), c.to_dev(phin)
    d_out = c.to_dev(np.zeros_like(psi))
    d_out2 = c.to_dev(np.zeros_like(psi)) if with_out2 else None
    c.be.stage(arr, len(specs), d_psi, d_phin, d_out, d_out2, base_mode, cdt, cdt2, t)
    got = c.interior(c.to_host(d_out))
    got2 = c.interior(c.to_host(d_out2)) if with_out2 else None
    return got, c.interior(want), got2, (c.interior(want2) if with_out2 else None)


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("shape,bcspec", [((41,), "periodic"), ((300,), ("extrapolation", 2)), ((37, 21), "neumann"),
                                          ((270, 40), "periodic"), ((13, 11, 9), ("extrapolation", 2)),
                                          ((37, 21, 45), "periodic"), ((64, 16, 33), "neumann")])
@pytest.mark.parametrize("name", list(SINGLE) + list(FUSED))
def test_stage_matches_oracle(hip, orc, mode, shape, bcspec, name):
    nd = len(shape)
    c = hip.Case(shape, bcspec, mode=mode)
    phi = _rand_field(shape, 2)
    specs = _fix_specs((SINGLE | FUSED)[name], nd, phi)
    for base_mode in (0, 1, 2, 3):
        got, want, _, _ = _run_stage(c, orc, specs, phi, base_mode)
        scale = np.abs(want).max()
        has_curv = any(s[0] == "curv" for s in specs)
        if mode == "strict" and not has_curv:
            assert np.array_equal(got, want), f"{name} base {base_mode}: max diff {np.abs(got - want).max():.3e}"
        else:
            assert np.abs(got - want).max() <= TOL_STAGE * scale, f"{name} base {base_mode}: {np.abs(got - want).max():.3e}"


def test_stage_second_output_rk2(hip, orc):
    c = hip.Case((29, 17, 15), "neumann", mode="strict")
    phi = _rand_field((29, 17, 15), 3)
    specs = _fix_specs(FUSED["all4_multipass"], 3, phi)
    got, want, got2, want2 = _run_stage(c, orc, specs[:2], phi, 0, with_out2=True)
    assert np.array_equal(got, want) and np.array_equal(got2, want2)
    got, want, got2, want2 = _run_stage(c, orc, specs, phi, 0, with_out2=True)   # multi-pass accumulates out2
    s = np.abs(want).max()
    assert np.abs(got - want).max() <= TOL_STAGE * s and np.abs(got2 - want2).max() <= TOL_STAGE * s


def test_rough_and_flat_fields_fast_mode(hip, orc):
    """Kinks (non-smooth WENO weights) and exactly flat regions (ε floor) in FAST mode."""
    shape = (33, 31, 29)
    c = hip.Case(shape, "neumann", mode="fast")
    phi = _rand_field(shape, 4, smooth=False)
    phi[:, :, :8] = 0.25          # exactly flat slab
    phi[10:14, :, :] = -0.125
    specs = [("adv", ("const", (0.7, -0.4, 0.9)), "weno5"), ("eik", None)]
    got, want, _, _ = _run_stage(c, orc, specs, phi, 0)
    assert not np.isnan(got).any()
    assert np.abs(got - want).max() <= TOL_STAGE * np.abs(want).max()


def test_separable_and_field_coefficients(hip, orc):
    shape = (24, 20, 18)
    c = hip.Case(shape, "neumann", lc=(0.0, 0.0, 0.0), hc=(1.0, 1.0, 1.0), mode="strict")
    x, y, z = c.grid.coords()
    s2 = lambda a: np.sin(np.pi * a) * np.sin(np.pi * a)
    s = lambda a: np.sin(2 * np.pi * a)
    tables = [[2 * s2(x), s(y), s(z)], [-s(x), s2(y), s(z)], [-s(x), s(y), s2(z)]]
    phi = _rand_field(shape, 5)
    got, want, _, _ = _run_stage(c, orc, [("adv", ("sep", tables, ("cos", 3.0)), "weno5"), ("eik", None)], phi, 1, t=0.4)
    assert np.array_equal(got, want)
    rng = np.random.default_rng(6)
    u = [np.asfortranarray(rng.standard_normal(shape)) for _ in range(3)]
    sp = np.asfortranarray(rng.standard_normal(shape))
    got, want, _, _ = _run_stage(c, orc, [("adv", ("field", u), "weno5"), ("nm", ("field", [sp]))], phi, 2)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shape", [(300,), (270, 40), (70, 21, 45)])
def test_plain_and_general_kernel_variants_agree(hip, orc, shape, monkeypatch):
    """FAST mode launches a compile-time 'plain' variant of the stage kernel when the call allows it (dense field, one
    output, terms in slot order, catalogued coefficients) and the general variant otherwise (LSM_STAGE_GENERIC=1
    forces it).  Both evaluate the same expressions: results must agree to rounding, and each with the oracle.  The
    fields change sign of u_d and of ϕ inside waves and keep it across others, so the wave-uniform sign paths and the
    per-lane path are both exercised."""
    nd = len(shape)
    c = hip.Case(shape, "neumann", lc=(0.0,) * nd, hc=(1.0,) * nd, mode="fast")
    coords = c.grid.coords()
    s2 = lambda a: np.sin(np.pi * a) * np.sin(np.pi * a)
    s = lambda a: np.sin(2 * np.pi * a)
    tables = [[(2 * s2(x) if k == d else (-1) ** d * s(x)) for k, x in enumerate(coords)] for d in range(nd)]
    phi = _rand_field(shape, 12)
    cases = [[("adv", ("sep", tables, ("cos", 3.0)), "weno5"), ("eik", None)],
             [("adv", ("const", (0.7, -0.4, 0.9)[:nd]), "weno5"), ("nm", ("const", (-0.3,)))],
             [("nm", ("const", (0.2,))), ("curv", ("const", (-0.05,)))],
             [("adv", ("const", (0.7, -0.4, 0.9)[:nd]), "upwind")]]
    if nd >= 2:
        cases.append([("adv", ("rot", 1.3, 0.45, 0.55), "weno5"), ("eik", phi)])
    for specs in cases:
        for base_mode in (0, 1, 2, 3):
            c.be.set_tuning("LSM_STAGE_GENERIC", 0)
            plain, want, _, _ = _run_stage(c, orc, specs, phi, base_mode, t=0.4)
            c.be.set_tuning("LSM_STAGE_GENERIC", 1)
            general, _, _, _ = _run_stage(c, orc, specs, phi, base_mode, t=0.4)
            c.be.set_tuning("LSM_STAGE_GENERIC", 0)
            scale = np.abs(want).max()
            assert np.abs(plain - want).max() <= TOL_STAGE * scale, (specs[0][:1], base_mode, np.abs(plain - want).max())
            assert np.abs(general - want).max() <= TOL_STAGE * scale, (specs[0][:1], base_mode, np.abs(general - want).max())
            assert np.abs(plain - general).max() <= 4e-16 * scale, (specs[0][:1], base_mode, np.abs(plain - general).max())


@pytest.mark.parametrize("shape", [(50,), (40, 30), (20, 18, 16)])
def test_cfl_bitwise(hip, orc, shape):
    from lsm_amd import _lib as L
    nd = len(shape)
    c = hip.Case(shape, "neumann", mode="fast")
    phi = _rand_field(shape, 7)
    rng = np.random.default_rng(8)
    u = [np.asfortranarray(rng.standard_normal(shape)) for _ in range(nd)]
    cases = [[("adv", ("const", (0.7, -0.4, 0.9)[:nd]), "weno5")], [("adv", ("field", u), "weno5")],
             [("nm", ("field", [u[0]])), ("eik", None)], [("curv", ("const", (-0.3,))), ("adv", ("field", u), "upwind")]]
    if nd >= 2:
        cases.append([("adv", ("rot", 1.0, 0.0, 0.0), "upwind")])
    for specs in cases:
        ot, arr = c.terms(specs)
        p = c.pad(phi)
        want = orc.cfl_padded(c.grid, c.bc, c.olay, ot, p, 0.2)
        got = c.be.compute_cfl_local(arr, len(specs), c.to_dev(p), 0.2)
        assert got == want, (specs, got, want)
    # NaN velocity must survive the reduction (Julia min propagates NaN; the host then throws)
    u[0][tuple(n // 2 for n in shape)] = np.nan
    ot, arr = c.terms([("adv", ("field", u), "weno5")])
    assert math.isnan(c.be.compute_cfl_local(arr, 1, c.to_dev(c.pad(phi)), 0.0))


@pytest.mark.parametrize("shape", [(60,), (40, 30), (24, 20, 18)])
def test_cfl_time_separable_coefficient_bitwise_at_every_time(hip, orc, shape):
    """u(x)·cos(πt/T): the arg-max candidates are recorded on the first call and only those nodes are
    evaluated afterwards — Δt must stay bitwise equal to the full reduction at every t (sign change,
    near-zero and zero factor included), for advection and normal motion, and a NaN table entry or a flat
    (all-equal) field must fall back to the sweep."""
    nd = len(shape)
    c = hip.Case(shape, "neumann", mode="fast")
    phi = _rand_field(shape, 3)
    rng = np.random.default_rng(11)
    tabs = lambda ncomp: [[rng.standard_normal(n) for n in shape] for _ in range(ncomp)]
    p = c.pad(phi)
    d = c.to_dev(p)
    T = 3.0
    for specs in ([("adv", ("sep", tabs(nd), ("cos", T)), "weno5")], [("nm", ("sep", tabs(1), ("cos", T))), ("eik", None)]):
        ot, arr = c.terms(specs)
        for t in (0.0, 0.3, 1.4999999, 1.5, 2.0, 2.9, 3.0, 7.7, 0.3):
            want = orc.cfl_padded(c.grid, c.bc, c.olay, ot, p, t)
            got = c.be.compute_cfl_local(arr, len(specs), d, t)
            assert got == want or (math.isinf(got) and math.isinf(want)), (specs[0][0], t, got, want)
    flat = [[np.ones(n) for n in shape] for _ in range(nd)]          # every node attains the maximum
    ot, arr = c.terms([("adv", ("sep", flat, ("cos", T)), "upwind")])
    for t in (0.1, 2.0):
        assert c.be.compute_cfl_local(arr, 1, d, t) == orc.cfl_padded(c.grid, c.bc, c.olay, ot, p, t)
    bad = tabs(nd)
    bad[0][0][shape[0] // 2] = np.nan
    ot, arr = c.terms([("adv", ("sep", bad, ("cos", T)), "upwind")])
    for t in (0.1, 2.0):
        assert math.isnan(c.be.compute_cfl_local(arr, 1, d, t))


INTEG = {"fe": 0, "rk2": 1, "rk3": 2}


@pytest.mark.parametrize("mode", ["strict", "fast"])
@pytest.mark.parametrize("integ", ["fe", "rk2", "rk3"])
@pytest.mark.parametrize("shape,bcspec", [
    ((64,), "periodic"), ((33, 30), ("extrapolation", 2)), ((21, 19, 17), "neumann"),
    # FAST steps serve the ghosts of copy-type x / y faces from the loads (include/lsm.h): every kind, per face, with partial tiles
    ((70, 20), "neumann"), ((45, 33), "periodic"), ((40, 18, 12), "periodic"), ((37, 21, 9), "symmetry"),
    ((36, 17, 11), [("neumann", "symmetry"), "periodic", ("extrapolation", 2)]),      # x and y from the loads, z weighted
    ((36, 17, 11), [("extrapolation", 2), ("symmetry", "neumann"), "neumann"]),         # x weighted: nothing redirected
    ((67, 13, 10), ["periodic", ("extrapolation", 3), "symmetry"]),                     # x only
], ids=lambda v: str(v).replace(" ", ""))
def test_advance_matches_literal_reference_loop(hip, orc, mode, integ, shape, bcspec):
    """lsm_advance_* (fused stages, in-place final stage, materialised ghosts) == the literal
    term-by-term _advance! of the oracle (dense arrays, recursive ghost resolution)."""
    nd = len(shape)
    c = hip.Case(shape, bcspec, mode=mode)
    phi = _rand_field(shape, 9)
    specs = _fix_specs(FUSED["adv+eik"], nd, phi)
    ref = phi.copy(order="F")
    tc, dt = 0.1, 2.0e-3
    orc.advance(INTEG[integ], c.grid, c.bc, ref, c.dense_terms(specs), tc, dt)
    _, arr = c.terms(specs)
    d_phi = c.to_dev(np.nan_to_num(c.pad(phi, fill=False), nan=0.0))
    b1, b2 = c.be.alloc(), c.be.alloc()
    c.be.advance_single(integ, arr, len(specs), d_phi, b1, b2, tc, dt, None)
    got = c.interior(c.to_host(d_phi))
    if mode == "strict":
        assert np.array_equal(got, ref), np.abs(got - ref).max()
    else:
        assert np.abs(got - ref).max() <= 3 * TOL_STAGE * np.abs(ref).max()
    # ghosts of phi are stale on return (include/lsm.h: the next reader fills them); one fill makes the padded array whole again
    c.be.fill_ghosts(d_phi)
    assert np.array_equal(c.to_host(d_phi), c.pad(got)) if mode == "strict" else True


def test_100_rk3_steps_fast_mode(hip, orc):
    shape = (40, 36, 32)
    c = hip.Case(shape, "neumann", lc=(0, 0, 0), hc=(1, 1, 1), mode="fast")
    x, y, z = c.grid.coords()
    s2 = lambda a: np.sin(np.pi * a) * np.sin(np.pi * a)
    s = lambda a: np.sin(2 * np.pi * a)
    tables = [[2 * s2(x), s(y), s(z)], [-s(x), s2(y), s(z)], [-s(x), s(y), s2(z)]]
    X, Y, Z = np.meshgrid(x, y, z, indexing="ij")
    phi = np.asfortranarray(np.sqrt((X - 0.35) ** 2 + (Y - 0.35) ** 2 + (Z - 0.35) ** 2) - 0.15)
    specs = [("adv", ("sep", tables, ("cos", 3.0)), "weno5"), ("eik", None)]
    ref = phi.copy(order="F")
    dense = c.dense_terms(specs)
    _, arr = c.terms(specs)
    d_phi = c.to_dev(np.nan_to_num(c.pad(phi, fill=False), nan=0.0))
    b1, b2 = c.be.alloc(), c.be.alloc()
    tc = 0.0
    for _ in range(100):
        dtc = orc.compute_cfl(c.grid, c.bc, ref, dense, tc)
        got_dt = c.be.compute_cfl_local(arr, 2, d_phi, tc)
        assert got_dt == dtc                      # Δt bitwise
        dt = 0.5 * dtc
        orc.advance(orc.RK3, c.grid, c.bc, ref, dense, tc, dt)
        c.be.advance_single("rk3", arr, 2, d_phi, b1, b2, tc, dt, None)
        tc += dt
    got = c.interior(c.to_host(d_phi))
    assert np.abs(got - ref).max() <= TOL_100 * np.abs(ref).max()


def test_extrema_and_eikonal_sign(hip, orc):
    shape = (30, 20, 10)
    c = hip.Case(shape, "neumann")
    phi = _rand_field(shape, 11)
    d = c.to_dev(c.pad(phi))
    lo, hi = c.be.extrema(d)
    assert lo == phi.min() and hi == phi.max()
    s0 = c.be.alloc()
    c.be.eikonal_sign(d, s0)
    assert np.array_equal(c.interior(c.to_host(s0)), orc.eikonal_sign(c.grid, phi))


def test_upload_download_roundtrip(hip):
    shape = (17, 13, 11)
    c = hip.Case(shape, "neumann")
    phi = _rand_field(shape, 12)
    t = c.be.alloc()
    c.be.upload(t, phi)
    assert np.array_equal(c.be.download(t), phi)
    assert np.array_equal(c.interior(c.to_host(t)), phi)


def test_errors_are_reported_not_thrown(hip):
    from lsm_amd import _lib as L
    c = hip.Case((16, 16), "neumann")
    _, arr = c.terms([("adv", ("const", (1.0, 0.0)), "weno5")])
    t = c.be.alloc()
    with pytest.raises(L.LsmError):
        c.be.stage(arr, 1, t, None, t, None, 0, 1e-3, 0.0, 0.0)   # out aliases psi
    with pytest.raises(L.LsmError):
        c.be.stage(arr, 1, t, None, c.be.alloc(), None, 1, 1e-3, 0.0, 0.0)   # RK3_S2 without phin


@pytest.mark.parametrize("shape,bcspec", [((40, 33), "periodic"), ((21, 19, 23), ("extrapolation", 2))])
def test_plane_range_stages_compose_to_the_full_stage(hip, orc, shape, bcspec):
    """lsm_stage_planes / lsm_fill_ghosts_planes (used to overlap the halo exchange): boundary planes
    first, interior afterwards == one full stage + full ghost fill, bit for bit."""
    nd = len(shape)
    c = hip.Case(shape, bcspec, mode="fast")
    phi = _rand_field(shape, 21)
    specs = _fix_specs(FUSED["all4_multipass"], nd, phi)
    _, arr = c.terms(specs)
    psi = c.to_dev(c.pad(phi))
    phin = c.to_dev(c.pad(np.asfortranarray(0.9 * phi)))
    full, part = c.be.alloc(), c.be.alloc()
    args = (2, 1.3e-3, 0.0, 0.25)
    c.be.stage(arr, len(specs), psi, phin, full, None, *args)
    c.be.fill_ghosts(full)
    n, B = shape[-1], 4
    for m0, m1 in ((0, B), (n - B, n)):
        c.be.stage_planes(arr, len(specs), psi, phin, part, None, *args, m0, m1)
        c.be.fill_ghosts_planes(part, m0, m1)
    c.be.stage_planes(arr, len(specs), psi, phin, part, None, *args, B, n - B)
    c.be.fill_ghosts_planes(part, B, n - B)
    c.be.fill_ghosts(part, 1 << (nd - 1))
    assert np.array_equal(c.to_host(part), c.to_host(full))


PAIRS = {
    "upwind": [("adv", ("const", (0.7, -0.4, 0.9)), "upwind")],
    "nm": [("nm", ("const", (0.6,)))],
    "nm_neg": [("nm", ("const", (-0.6,)))],
    "eik_current": [("eik", None)],
    "eik_frozen": [("eik", "phi0")],
}


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("shape,bcspec", [((128, 16, 20), "neumann"), ((256, 11, 70), "periodic"), ((130, 20, 9), ("extrapolation", 2)),
                                          ((384, 8, 8), "symmetry"), ((200, 13, 30), [("neumann", ("extrapolation", 3)), "periodic", ("symmetry", "linear")])])
@pytest.mark.parametrize("name", list(PAIRS))
def test_pair_kernels_match_the_oracle_and_the_one_node_kernels(hip, orc, monkeypatch, name, shape, bcspec, dtype):
    """The two-nodes-per-thread kernels (stage_tile2: 16-byte accesses, tiles 128 nodes wide) serve the dense FAST
    launches of a single upwind / NormalMotion / Eikonal term when n1 is even and
    >= 128.  They call the same node_update on the same LDS tile as the one-node kernels (LSM_PAIRS=0): results bit for bit
    equal to theirs, and within the stage tolerance of the oracle — whole tiles, partial tiles in x (130, 200, 384 = 3 x 128),
    partial tiles in y, every base mode, float32 storage."""
    dt = np.dtype(dtype)
    c = hip.Case(shape, bcspec, mode="fast", dtype=dt)
    phi = _rand_field(shape, 7)
    if dt == np.float32:
        phi = np.asfortranarray(phi.astype(np.float32).astype(np.float64))
    specs = _fix_specs(PAIRS[name], 3, phi)
    for base_mode in (0, 1, 2, 3):
        c.be.set_tuning("LSM_PAIRS", 1)
        got, want, _, _ = _run_stage(c, orc, specs, phi, base_mode)
        c.be.set_tuning("LSM_PAIRS", 0)
        one, _, _, _ = _run_stage(c, orc, specs, phi, base_mode)
        assert np.array_equal(got, one), (name, base_mode, np.abs(got - one).max())
        if dt == np.float64:
            assert np.abs(got - want).max() <= TOL_STAGE * np.abs(want).max(), (name, base_mode)
        else:
            assert np.abs(got - want.astype(np.float32).astype(np.float64)).max() <= 2.4e-7 * np.abs(want).max(), (name, base_mode)


WEIGHTED_TERMS = {
    "nm+curv": [("nm", ("const", (0.1,))), ("curv", ("const", (-0.1,)))],          # BASELINE config 3's pair
    "upwind": [("adv", ("const", (0.5, 0.25, -1.0)), "upwind")],
    "eik": [("eik", None)],
    "nm": [("nm", ("const", (-0.4,)))],
    "curv": [("curv", ("const", (-0.05,)))],
    "nm+eik_multipass": [("nm", ("const", (0.3,))), ("eik", None), ("curv", ("const", (-0.02,)))],
}


@pytest.mark.parametrize("terms", list(WEIGHTED_TERMS))
@pytest.mark.parametrize("bcspec", [
    ("extrapolation", 2),                                                            # config 3's boundary condition, every face
    [(("extrapolation", 1), ("extrapolation", 3)), (("extrapolation", 2), "neumann"), "neumann"],      # a degree per side, a copy-type side beside a weighted one
    [("symmetry", ("extrapolation", 2)), ("extrapolation", 3), ("extrapolation", 1)],
], ids=lambda v: str(v).replace(" ", ""))
@pytest.mark.parametrize("integ", ["rk3", "fe"])
def test_weighted_and_copy_type_faces_side_by_side(hip, orc, terms, bcspec, integ):
    """ExtrapolationBC{P}, P = 1..3, on x / y faces of a 3-D grid whose extents are multiples of every tile shape (128 | n1, 8 | n2), with
    every face weighted, a degree per side, and copy-type sides (served by the stage kernel's loads in FAST steps) beside weighted
    ones (materialised by the fill) — the two mechanisms meet in the tile corners.  The step must equal the oracle's literal
    loop (dense arrays, recursive ghost resolution: src/boundaryconditions.jl:134-144, src/meshfield.jl:248-260) to FAST's tolerance.
    (Round 4 resolved the weighted faces inside the kernel as well; these cases caught its bugs and stay, the code did not:
    DESIGN.md §3.1.)"""
    shape = (128, 16, 11)
    c = hip.Case(shape, bcspec, mode="fast")
    phi = _rand_field(shape, 21)
    specs = _fix_specs(WEIGHTED_TERMS[terms], 3, phi)
    ref = phi.copy(order="F")
    tc, dt = 0.2, 1.5e-3
    orc.advance(INTEG[integ], c.grid, c.bc, ref, c.dense_terms(specs), tc, dt)
    _, arr = c.terms(specs)
    d_phi = c.to_dev(np.nan_to_num(c.pad(phi, fill=False), nan=0.0))       # ghost layers hold zeros: a kernel that read them would show
    b1, b2 = c.be.alloc(), c.be.alloc()
    c.be.advance_single(integ, arr, len(specs), d_phi, b1, b2, tc, dt, None)
    got = c.interior(c.to_host(d_phi))
    assert np.abs(got - ref).max() <= 3 * TOL_STAGE * np.abs(ref).max(), float(np.abs(got - ref).max())

"""The reference's own hot-path tests, run through the drop-in host API (lsm_amd) on the GPU.
Same grids, boundary conditions, integrators and thresholds as the reference test files cited."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsm():
    import lsm_amd
    return lsm_amd


def _advection_error_1d(lsm, integrator, N, u=1.0, tf=0.5, scheme=None):
    """test/test-timestepping.jl:8-22"""
    grid = lsm.CartesianGrid((-1.0,), (1.0,), (N,))
    phi = lsm.MeshField(lambda x: np.sin(np.pi * x[0]), grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((u,), scheme or lsm.WENO5()),), ic=phi, bc=lsm.PeriodicBC(),
                              integrator=integrator)
    lsm.integrate_(eq, tf)
    out = lsm.current_state(eq).values()
    x = grid.coords()[0]
    return np.abs(out - np.sin(np.pi * (x - u * tf))).max()


def test_integrator_accuracy_1d(lsm):
    """test/test-timestepping.jl:24-34"""
    assert _advection_error_1d(lsm, lsm.ForwardEuler(), 200) < 0.05
    assert _advection_error_1d(lsm, lsm.RK2(), 200) < 1.0e-3
    assert _advection_error_1d(lsm, lsm.RK3(), 200) < 1.0e-5


def test_weno5_spatial_order(lsm):
    """test/test-levelsetequation.jl:26-45"""
    Ns = [20, 40, 80]
    e = [_advection_error_1d(lsm, lsm.RK3(cfl=1.0e-2), N) for N in Ns]
    orders = [math.log(e[i] / e[i + 1]) / math.log(Ns[i + 1] / Ns[i]) for i in range(2)]
    assert all(o >= 4.5 for o in orders), orders


def test_dumbbell_rotation_792_steps(lsm):
    """docs/src/time-integrators.md:92-94: one revolution at cfl 0.5 costs exactly 792 steps and
    integrate! lands exactly on tf."""
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (64, 64))
    disk = lambda c: lsm.MeshField(lambda x: np.hypot(x[0] - c[0], x[1] - c[1]) - 0.25, grid).vals
    bar = lsm.MeshField(lambda x: np.maximum(np.abs(x[0]) - 0.5, np.abs(x[1]) - 0.1), grid).vals
    phi0 = lsm.MeshField(np.minimum(np.minimum(disk((-0.5, 0.0)), disk((0.5, 0.0))), bar), grid)
    eq = lsm.LevelSetEquation(terms=lsm.AdvectionTerm(lsm.RigidRotation()), ic=phi0, bc=lsm.NeumannBC(), integrator=lsm.RK3())
    steps = []
    lsm.integrate_(eq, 2 * math.pi, posthook=lambda e: steps.append(e.current_time()))
    assert len(steps) == 792
    assert eq.current_time() == 2 * math.pi
    out = eq.current_state().values()
    assert np.abs(out - phi0.vals)[np.abs(phi0.vals) < 0.1].max() < 0.05   # the shape came back


def test_eikonal_term_converges_to_sdf(lsm):
    """test/test-levelsetterms.jl:33-51"""
    grid = lsm.CartesianGrid((-1.0,), (1.0,), (101,))
    phi = lsm.MeshField(lambda x: 2 * (x[0] - 0.3), grid)
    eq = lsm.LevelSetEquation(terms=(lsm.EikonalReinitializationTerm(phi),), ic=phi, bc=lsm.LinearExtrapolationBC())
    lsm.integrate_(eq, 2.0)
    out = eq.current_state().values()
    exact = grid.coords()[0] - 0.3
    assert np.where(np.abs(out) > 0.5, 0.0, np.abs(out - exact)).max() < 0.05


def test_nan_robustness(lsm):
    """test/test-levelsetterms.jl:53-77"""
    grid = lsm.CartesianGrid((-2.0, -2.0), (2.0, 2.0), (31, 31))
    phi = lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - 0.7, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.CurvatureTerm(-0.1),), ic=phi, bc=lsm.NeumannBC(), integrator=lsm.RK2())
    lsm.integrate_(eq, 0.1)
    assert not np.isnan(eq.current_state().values()).any()
    g1 = lsm.CartesianGrid((-1.0,), (1.0,), (31,))
    flat = lsm.MeshField(lambda x: 0.0 * x[0], g1)
    eq2 = lsm.LevelSetEquation(terms=(lsm.EikonalReinitializationTerm(),), ic=flat, bc=lsm.NeumannBC(), integrator=lsm.RK2())
    lsm.integrate_(eq2, 0.1)
    assert not np.isnan(eq2.current_state().values()).any()


def test_normal_motion_and_curvature_orders(lsm):
    """test/test-levelsetequation.jl:67-119"""
    def run(term, exact, r0, tf):
        Ns, errs = [30, 60, 120], []
        for N in Ns:
            grid = lsm.CartesianGrid((-2.0, -2.0), (2.0, 2.0), (N, N))
            phi = lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - r0, grid)
            eq = lsm.LevelSetEquation(terms=(term,), ic=phi, bc=lsm.ExtrapolationBC(2), integrator=lsm.RK3())
            lsm.integrate_(eq, tf)
            out = eq.current_state().values()
            X, Y = np.meshgrid(*grid.coords(), indexing="ij")
            r = np.hypot(X, Y)
            errs.append(np.abs(out - exact(r))[(r >= 0.5) & (r <= 1.5)].max())
        return [math.log(errs[i] / errs[i + 1]) / math.log(2) for i in range(2)]
    assert all(o >= 1.5 for o in run(lsm.NormalMotionTerm(0.5), lambda r: r - 0.5 - 0.5 * 0.2, 0.5, 0.2))
    assert all(o >= 1.5 for o in run(lsm.CurvatureTerm(-0.1), lambda r: np.sqrt(r ** 2 + 0.2 * 0.2) - 0.7, 0.7, 0.2))


def test_cfl_error_and_backward_time(lsm):
    """src/levelsetterms.jl:26, src/levelsetequation.jl:196"""
    grid = lsm.CartesianGrid((-1.0,), (1.0,), (32,))
    phi = lsm.MeshField(lambda x: x[0], grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((float("nan"),)),), ic=phi, bc=lsm.NeumannBC())
    with pytest.raises(ValueError, match="invalid time-step based on CFL condition"):
        lsm.integrate_(eq, 1.0)
    eq2 = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((1.0,)),), ic=phi, bc=lsm.NeumannBC(), t=1.0)
    with pytest.raises(ValueError, match="cannot be solved back in time"):
        lsm.integrate_(eq2, 0.5)
    with pytest.raises(ValueError, match="no boundary conditions"):
        lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((1.0,)),), ic=phi)


def test_callable_velocity_and_update_func_hooks(lsm, orc):
    """A python closure velocity u(x,t) (re-sampled per stage: slow path) and an update_func hook
    (src/levelsetterms.jl:65-69) give the same result as the catalogued rotation."""
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (48, 40))
    ic = lsm.MeshField(lambda x: np.hypot(x[0] - 0.3, x[1]) - 0.4, grid)
    mk = lambda vel, **kw: lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(vel, lsm.WENO5(), **kw),), ic=ic, bc=lsm.NeumannBC(),
                                                integrator=lsm.RK3(), mode="strict")
    a = mk(lsm.RigidRotation())
    calls = []
    b = mk(lambda x, t: (-x[1], x[0]), update_func=lambda coeff, field, t: calls.append(t))
    lsm.integrate_(a, 0.2)
    lsm.integrate_(b, 0.2)
    assert np.array_equal(a.current_state().values(), b.current_state().values())
    assert len(calls) > 0 and len(calls) % 4 == 0   # once before the CFL + once per RK3 stage


def test_device_fields_as_coefficients(lsm):
    """A ROCMeshField as NormalMotionTerm speed and one per component as AdvectionTerm velocity (the reference's
    `MeshField` coefficients, src/levelsetterms.jl:42, resident on the device): copied HBM to HBM into the term's
    coefficient arrays at construction and by `coeff.set_values(...)` inside an update_func — equal to the host-array
    route bit for bit; a float32 device field widens exactly."""
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (48, 40))
    X, Y = np.meshgrid(*grid.coords(), indexing="ij")
    ic = lsm.MeshField(lambda x: np.hypot(x[0] - 0.1, x[1]) - 0.5, grid)
    sp = np.asfortranarray(0.3 + 0.2 * X * Y)
    u = [np.asfortranarray(0.5 - Y), np.asfortranarray(0.25 + X * X)]
    mk = lambda speed, vel, **kw: lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(speed, **kw), lsm.AdvectionTerm(vel, lsm.WENO5())),
                                                        ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK2())
    ref = mk(lsm.MeshField(sp, grid), lsm.MeshField(np.stack(u), grid))
    lsm.integrate_(ref, 0.05)
    want = ref.current_state().values()
    donor = mk(0.0, (0.0, 0.0))                         # a handle of the same layout to put device fields on
    dev = lambda a: lsm.ROCMeshField.from_host(donor.backend, lsm.MeshField(a, grid), donor.bcs)
    eq = mk(dev(sp), (dev(u[0]), dev(u[1])))
    lsm.integrate_(eq, 0.05)
    assert np.array_equal(eq.current_state().values(), want)
    # refreshed from a device field inside the hook (the natural update_func-driven speed)
    calls = []
    eq2 = mk(dev(0 * sp), (dev(u[0]), dev(u[1])), update_func=lambda coeff, field, t: (calls.append(t), coeff.set_values(dev(sp))))
    lsm.integrate_(eq2, 0.05)
    assert calls and np.array_equal(eq2.current_state().values(), want)
    with pytest.raises(ValueError, match="component"):
        mk(dev(sp), (dev(u[0]),))


def test_state_is_a_copy_and_hooks_can_mutate(lsm):
    """src/levelsetequation.jl:67-76 (ic is copied) and :180-185 (hooks may mutate the state)."""
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (32, 32))
    ic = lsm.MeshField(lambda x: np.hypot(x[0], x[1]) - 0.5, grid)
    keep = ic.vals.copy()
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(1.0),), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.ForwardEuler())
    times = [0.0]
    lsm.integrate_(eq, 0.05, prehook=lambda e: e.current_state().copy_(ic), posthook=lambda e: times.append(e.current_time()))
    assert np.array_equal(ic.vals, keep)
    last = times[-1] - times[-2]
    one = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(1.0),), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.ForwardEuler())
    lsm.integrate_(one, last)
    # with the state reset before every step, the final state is ONE step of size `last` from ic
    assert np.allclose(eq.current_state().values(), one.current_state().values(), atol=1e-12)


def test_slab_code_path_on_one_gpu_matches_single_device_path(lsm):
    """The slab driver (stage-by-stage with lsm_stage_planes / lsm_fill_ghosts_planes, RCCL group of
    one rank) must reproduce lsm_advance_rk3 bit for bit — with and without the boundary-first split."""
    import os
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1)
        created = True
    try:
        grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (40, 36, 44))
        ic = lsm.MeshField(lambda x: np.sqrt((x[0] - 0.35) ** 2 + (x[1] - 0.35) ** 2 + (x[2] - 0.35) ** 2) - 0.15, grid)
        mk = lambda **kw: lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.vortex_deformation(grid), lsm.WENO5()),
                                                      lsm.EikonalReinitializationTerm()), ic=ic, bc=lsm.NeumannBC(),
                                               integrator=lsm.RK3(), **kw)
        ref = mk()
        lsm.integrate_(ref, 0.02)
        want = ref.current_state().values()
        for force in (False, True):
            eq = mk(comm=dist.group.WORLD)
            eq._force_overlap = force
            lsm.integrate_(eq, 0.02)
            assert np.array_equal(eq.gather_state(), want), force
    finally:
        if created:
            dist.destroy_process_group()


def test_volume_perimeter_known_answers_and_parity(lsm, orc):
    """jldoctests src/levelsetops.jl:14-25,126-137 (200² circle) through the device reductions, plus
    oracle parity on a 3-D field; sums are order-dependent in the reference itself: 1e-12 relative."""
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (200, 200))
    ic = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2) - 0.5, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=ic, bc=lsm.LinearExtrapolationBC())
    assert lsm.volume(eq) == pytest.approx(0.7854362890190668, rel=1e-12)
    assert lsm.perimeter(eq) == pytest.approx(3.1426415491430384, rel=1e-12)
    g3 = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (40, 36, 32))
    f3 = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.55, g3)
    e3 = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=f3, bc=lsm.NeumannBC())
    og = orc.Grid((-1, -1, -1), (1, 1, 1), (40, 36, 32))
    assert lsm.volume(e3) == pytest.approx(orc.volume(og, f3.vals), rel=1e-12)
    assert lsm.perimeter(e3) == pytest.approx(orc.perimeter(og, f3.vals, orc.make_bc("neumann", 3)), rel=1e-12)
    # scalar getindex / setindex! on the device field (src/meshfield.jl:213-217,263-266)
    st = e3.current_state()
    assert st[(3, 4, 5)] == f3.vals[3, 4, 5]
    st[(3, 4, 5)] = 7.5
    assert st.values()[3, 4, 5] == 7.5 and st.ghosts_dirty


@pytest.mark.parametrize("world,bc,overlap,integ", [(2, "neumann", True, "rk3"), (3, "neumann", False, "rk3"), (3, "periodic", True, "rk3"),
                                                   (2, "periodic", False, "rk3"), (3, "periodic", True, "rk2"), (2, "neumann", True, "fe"),
                                                   (4, "extrap2", True, "rk2")])
def test_slab_handles_on_one_gpu_compose_to_the_single_device_result(lsm, monkeypatch, world, bc, overlap, integ):
    """True slab handles (plane offset, LSM_BC_NONE interfaces) driven through the library's own slab step: `world`
    ranks run as threads of this process over an LSM_COMM_LOCAL group (include/lsm.h, "multi-GPU"), i.e.
    lsm_comm_attach_local + lsm_advance_* with lsm_halo_start / lsm_halo_wait / lsm_allreduce_dt inside.  The
    concatenated slabs must equal the single-device run bit for bit, Δt reduction included."""
    import threading
    monkeypatch.setenv("LSM_SLAB_OVERLAP", "1" if overlap else "0")
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (24, 20, 41))
    ic = lsm.MeshField(lambda x: np.sqrt((x[0] - 0.35) ** 2 + (x[1] - 0.35) ** 2 + (x[2] - 0.4) ** 2) - 0.2, grid)
    bcs = {"periodic": lsm.PeriodicBC(), "neumann": lsm.NeumannBC(), "extrap2": (lsm.NeumannBC(), lsm.SymmetryBC(), lsm.ExtrapolationBC(2))}[bc]
    I = {"rk3": lsm.RK3, "rk2": lsm.RK2, "fe": lsm.ForwardEuler}[integ]
    mk = lambda **kw: lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.vortex_deformation(grid), lsm.WENO5()),
                                                  lsm.EikonalReinitializationTerm()), ic=ic, bc=bcs, integrator=I(), **kw)
    ref = mk()
    lsm.integrate_(ref, 0.03)
    want = ref.current_state().values()
    g = lsm.LocalGroup(world)
    got, errs, info = [None] * world, [], [None] * world

    def run(r):
        try:
            eq = mk(comm=g.rank(r))
            assert eq.lib_comm
            info[r] = eq.backend.comm_info()
            lsm.integrate_(eq, 0.03)
            got[r] = eq.current_state().values()
            full = eq.gather_state()
            assert np.array_equal(full, want)
        except BaseException as e:   # noqa: BLE001 - reported by the main thread
            import traceback
            errs.append((r, traceback.format_exc()))
            g.abort()

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, errs
    assert info == [(r, world, 2) for r in range(world)]            # LSM_COMM_LOCAL
    full = np.concatenate(got, axis=2)
    assert full.shape == want.shape
    assert np.array_equal(full, want), np.abs(full - want).max()


@pytest.mark.parametrize("world,integ,reinit", [(2, "rk3", False), (3, "rk2", False), (3, "fe", False), (2, "rk2", True), (3, "rk3", True)])
def test_slab_decomposed_narrow_band_matches_single_device(lsm, monkeypatch, world, integ, reinit):
    """BASELINE config 5's decomposition: a narrow band cut into slabs.  Each rank carries BAND_OVERLAP planes of its
    neighbours and refreshes them after every stage and band update; on the planes a rank owns, the band set and
    the values must equal the single-device band run bit for bit.  Ranks = threads over an LSM_COMM_LOCAL group: the mask
    planes, the sparse values and Δt all move inside the library (lsm_band_overlap_mask / _values, lsm_advance_band_*,
    lsm_allreduce_dt) — the calls a Julia host would make."""
    import threading
    n = (28, 24, 66)
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), n)
    phi = lsm.MeshField(lambda x: np.sqrt((x[0] - 0.05) ** 2 + x[1] ** 2 + (x[2] + 0.02) ** 2) - 0.62, grid)
    I = {"rk3": lsm.RK3, "rk2": lsm.RK2, "fe": lsm.ForwardEuler}[integ]
    mk = lambda **kw: lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.02)),
                                           ic=lsm.NarrowBandMeshField(phi, nlayers=3), bc=lsm.ExtrapolationBC(2), integrator=I(), **kw)
    import warnings
    hook = (lambda e: lsm.reinitialize_(e)) if reinit else None     # the reference's workflow: reinitialize! every step
    ref = mk()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lsm.integrate_(ref, 0.04, posthook=hook)
    st = ref.current_state()
    want_m, want_v = st.active_mask(), st.values()
    assert want_m[:, :, 20:46].any() and 2000 < want_m.sum()       # the band crosses every slab interface
    g = lsm.LocalGroup(world)
    got, errs = [None] * world, []

    def run(r):
        try:
            eq = mk(comm=g.rank(r))
            assert eq.lib_comm and eq.backend.comm_info() == (r, world, 2)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                lsm.integrate_(eq, 0.04, posthook=hook)
            o0, on = eq.own
            s = eq.current_state()
            got[r] = (s.active_mask()[..., o0:o0 + on], s.values()[..., o0:o0 + on])
        except BaseException as e:   # noqa: BLE001 - reported by the main thread
            import traceback
            errs.append((r, traceback.format_exc()))
            g.abort()

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, errs
    m = np.concatenate([g[0] for g in got], axis=2)
    v = np.concatenate([g[1] for g in got], axis=2)
    assert np.array_equal(m, want_m)
    d = np.abs(np.where(m, v - want_v, 0.0))
    assert np.array_equal(v[m], want_v[m]), (float(d.max()), np.argwhere(d > 0)[:8].tolist(), int((d > 0).sum()))


def test_fast_mode_refuses_fields_outside_its_domain(lsm):
    """include/lsm.h, LSM_MODE_FAST: beyond |Δϕ| ≈ 1e35 the one-reciprocal WENO5 weights leave the fp64 range.  The host
    layer asks lsm_check_range when the equation is built (and every 64 steps) and refuses — no silent Inf/NaN; STRICT
    mode has no such limit and reproduces the oracle on the same 1e40-scaled field."""
    from oracle import oracle as orc
    grid = lsm.CartesianGrid((-1, -1), (1, 1), (40, 36))
    base = lsm.MeshField(lambda x: np.hypot(x[0] - 0.1, x[1]) - 0.5, grid).vals
    mk = lambda scale, mode: lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((0.7, -0.4), lsm.WENO5()),), ic=lsm.MeshField(scale * base, grid),
                                                  bc=lsm.NeumannBC(), integrator=lsm.RK3(), mode=mode)
    with pytest.raises(ValueError, match="outside the domain of the FAST arithmetic mode"):
        mk(1e40, "fast")
    ok_eq = mk(1e30, "fast")                                   # large but inside: still within the stated tolerance
    assert ok_eq.backend.check_range(ok_eq.state.buf) == (True, float(np.abs(1e30 * base).max()))
    eq = mk(1e40, "strict")
    assert eq.backend.check_range(eq.state.buf)[0]
    dt = 0.5 * eq.compute_cfl(0.0)
    eq._advance(0.0, dt)
    og = orc.Grid((-1, -1), (1, 1), (40, 36))
    ref = np.asfortranarray(1e40 * base)
    orc.advance(orc.RK3, og, orc.make_bc("neumann", 2), ref, [orc.advection(orc.const(0.7, -0.4))], 0.0, dt)
    assert np.array_equal(eq.current_state().values(), ref)
    ok_eq._advance(0.0, dt)
    ref30 = np.asfortranarray(1e30 * base)
    orc.advance(orc.RK3, og, orc.make_bc("neumann", 2), ref30, [orc.advection(orc.const(0.7, -0.4))], 0.0, dt)
    assert np.abs(ok_eq.current_state().values() - ref30).max() <= 3e-13 * np.abs(ref30).max()
    # a field that grows out of the domain during a run is caught by the periodic check
    ok_eq.state.copy_(lsm.MeshField(1e36 * base, grid))
    ok_eq._range_checked_at = 63
    with pytest.raises(ValueError, match="outside the domain"):
        lsm.integrate_(ok_eq, ok_eq.current_time() + 2 * dt)


def test_whole_grid_stages_of_one_handle_on_two_streams(lsm):
    """The dynamic tail of the dense 3-D stage kernel hands its short chunks out through a ticket counter.  Every launch is
    self-contained (it resets its counter, consecutive launches take different counters of a ring), and a launch on a
    caller's stream — which may run concurrently with one on the handle's — gets the static tail: two whole-grid stages of
    one handle issued back to back on two streams produce exactly what the serial launches produce, and so do the serial
    launches that follow them."""
    import ctypes as C
    import torch
    from lsm_amd import _lib as L
    from lsm_amd.api import _terms_c
    n = 256                                            # 8 chunk layers of 32 planes: graded + dynamic tail are on
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (n, n, n))
    ic = lsm.LazyMeshField(lambda x: np.sqrt((x[0] - 0.35) ** 2 + (x[1] - 0.35) ** 2 + (x[2] - 0.35) ** 2) - 0.15, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((0.7, -0.4, 1.1), lsm.WENO5()), lsm.EikonalReinitializationTerm()),
                              ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3())
    b = eq.backend
    psi = eq.state.buf
    b.fill_ghosts(psi, 7)
    arr, nt = _terms_c(eq.terms), len(eq.terms)
    ref = [b.alloc() for _ in range(2)]
    for k, o in enumerate(ref):
        b.stage(arr, nt, psi, None, o, None, L.BASE_PSI, 1e-3 * (k + 1), 0.0, 0.0)
    b.sync()
    outs = [b.alloc() for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for rep in range(3):
        for k, (o, s) in enumerate(zip(outs, streams)):
            L.check(b.h, b.lib.lsm_stage(b.h, arr, nt, b.ptr(psi), None, b.ptr(o), None, L.BASE_PSI, 1e-3 * (k + 1), 0.0, 0.0,
                                         C.c_void_p(s.cuda_stream)), "lsm_stage")
        # ... and one on the handle's own stream at the same time (it uses the ticket counters)
        own = b.alloc()
        b.stage(arr, nt, psi, None, own, None, L.BASE_PSI, 1e-3, 0.0, 0.0)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1]) and torch.equal(own, ref[0]), rep
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()              # the zeroing ran on torch's stream: order it before the next round on the side streams
    # many serial launches in a row: every slot of the counter ring is taken more than once
    for rep in range(40):
        o = outs[rep % 2]
        b.stage(arr, nt, psi, None, o, None, L.BASE_PSI, 1e-3 * (rep % 2 + 1), 0.0, 0.0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], ref[0]) and torch.equal(outs[1], ref[1])


def test_profile_sampling_counts_every_launch_and_scales_the_time(lsm):
    """lsm_profile_enable(h, N) (include/lsm.h): HIP-event pairs around every N-th stage launch only — the timing must not pay for
    itself — while lsm_profile_read still reports every launch and a total time scaled from the sampled ones."""
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (96, 96, 96))
    phi = lsm.MeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((1.0, 0.5, -0.25), lsm.WENO5()),), ic=phi, bc=lsm.NeumannBC(), integrator=lsm.RK3())
    be = eq.backend
    out = {}
    for _ in range(30):                                 # past the first launches' one-off costs
        eq._advance(0.0, 1e-3)
    for every in (1, 4):
        be.profile_enable(every)
        for _ in range(8):
            eq._advance(0.0, 1e-3)
        n, ms = be.profile_read()
        out[every] = (n, ms)
    be.profile_enable(False)
    eq._advance(0.0, 1e-3)
    assert be.profile_read() == (0, 0.0)
    assert out[1][0] == 24 and out[4][0] == 24          # three stage launches per RK3 step, sampled or not
    assert out[1][1] > 0 and 0.4 < out[4][1] / out[1][1] < 2.5

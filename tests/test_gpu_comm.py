"""The multi-GPU entry points of the C ABI (include/lsm.h, "multi-GPU"): lsm_comm_attach_rccl / _local, lsm_halo_*,
lsm_allreduce_dt and lsm_advance_* on slab handles — on ONE GPU: a one-rank RCCL communicator (the real librccl), and
LSM_COMM_LOCAL groups whose ranks are handles of this process (one thread per rank in test_gpu_api.py; here one thread
driving every rank stage by stage).  Real multi-GPU runs are the driver's SCALE bench."""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lsm():
    import lsm_amd
    return lsm_amd


def _equation(lsm, grid, ic, **kw):
    return lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.vortex_deformation(grid), lsm.WENO5()), lsm.EikonalReinitializationTerm()),
                                ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3(), **kw)


def test_one_rank_rccl_communicator(lsm):
    """ncclCommInitRank through the library (librccl opened at run time), a one-rank group on a whole-grid handle:
    attach, info, Δt all-reduce (NaN survives), an exchange with no neighbours, overlap switch, detach; the step
    through lsm_advance_rk3 with the communicator attached equals the plain one bit for bit."""
    from lsm_amd import _lib as L
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (40, 36, 44))
    ic = lsm.MeshField(lambda x: np.sqrt((x[0] - 0.35) ** 2 + (x[1] - 0.35) ** 2 + (x[2] - 0.35) ** 2) - 0.15, grid)
    ref = _equation(lsm, grid, ic)
    lsm.integrate_(ref, 0.02)
    eq = _equation(lsm, grid, ic)
    b = eq.backend
    uid = b.comm_unique_id()
    assert len(uid) == L.COMM_ID_BYTES and any(uid)
    b.comm_attach_rccl(uid, 0, 1)
    assert b.comm_info() == (0, 1, L.COMM_RCCL)
    with pytest.raises(L.LsmError, match="already attached"):
        b.comm_attach_rccl(uid, 0, 1)
    assert b.allreduce_dt(0.125) == 0.125 and math.isnan(b.allreduce_dt(float("nan"))) and b.allreduce_dt(float("inf")) == float("inf")
    b.halo_exchange(eq.state.buf)
    b.comm_set_overlap(False)
    b.comm_set_overlap(True)
    lsm.integrate_(eq, 0.02)
    assert np.array_equal(eq.current_state().values(), ref.current_state().values())
    L.check(b.h, b.lib.lsm_comm_detach(b.h), "lsm_comm_detach")
    assert b.comm_info() == (0, 1, L.COMM_NONE)
    with pytest.raises(L.LsmError, match="no communicator"):
        b.halo_exchange(eq.state.buf)


def test_attach_validates_the_slab_faces(lsm):
    from lsm_amd import _lib as L
    from lsm_amd.api import _bc_c, _normalize_bc
    from lsm_amd.backend import HipBackend
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (16, 16, 24))
    bcs = _normalize_bc(lsm.NeumannBC(), 3)
    whole = HipBackend(grid._c(), _bc_c(bcs, 3, (False, False)))
    slab0 = HipBackend(grid._c(), _bc_c(bcs, 3, (False, True)), slab=(0, 12))
    slab1 = HipBackend(grid._c(), _bc_c(bcs, 3, (True, False)), slab=(12, 12))
    with pytest.raises(L.LsmError, match="LSM_BC_NONE"):
        HipBackend.comm_attach_local([slab0, whole])          # rank 1 lacks the interface face
    with pytest.raises(L.LsmError, match="one-rank group"):
        HipBackend.comm_attach_local([slab0])
    HipBackend.comm_attach_local([slab0, slab1])
    assert slab0.comm_info() == (0, 2, L.COMM_LOCAL) and slab1.comm_info() == (1, 2, L.COMM_LOCAL)
    t = slab0.alloc()
    slab0.halo_start(t)
    with pytest.raises(L.LsmError, match="has not been waited for"):
        slab0.halo_start(t)
    slab1.halo_start(slab1.alloc())
    slab0.halo_wait()
    slab1.halo_wait()
    for b in (slab0, slab1, whole):
        b.close()


@pytest.mark.parametrize("periodic", [False, True])
def test_local_group_driven_stage_by_stage_from_one_thread(lsm, periodic):
    """One host thread, every rank: the interface planes of each slab first (lsm_stage_planes + lsm_fill_ghosts_planes),
    lsm_halo_start on every handle, the interiors, lsm_halo_wait on every handle, then the physical ghost planes —
    three RK3 stages by hand.  Equal to the single-device step bit for bit."""
    from lsm_amd import _lib as L
    from lsm_amd.api import _bc_c, _normalize_bc, _terms_c
    from lsm_amd.backend import HipBackend
    n, world = (20, 24, 37), 3
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), n)
    bc = lsm.PeriodicBC() if periodic else (lsm.NeumannBC(), lsm.ExtrapolationBC(2), lsm.SymmetryBC())
    ic = lsm.MeshField(lambda x: np.sqrt((x[0] - 0.4) ** 2 + (x[1] - 0.5) ** 2 + (x[2] - 0.45) ** 2) - 0.25, grid)
    mk = lambda: lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((0.7, -0.4, 1.1), lsm.WENO5()), lsm.EikonalReinitializationTerm()),
                                      ic=ic, bc=bc, integrator=lsm.RK3())
    ref = mk()
    dt = 0.5 * ref.compute_cfl(0.0)
    ref._advance(0.0, dt)
    want = ref.current_state().values()

    bcs = _normalize_bc(bc, 3)
    counts = [13, 12, 12]
    los = [0, 13, 25]
    backs = [HipBackend(grid._c(), _bc_c(bcs, 3, (r > 0 or periodic, r < world - 1 or periodic)), slab=(los[r], counts[r])) for r in range(world)]
    HipBackend.comm_attach_local(backs)
    terms = mk().terms                      # constant coefficients: the same LsmTerm array serves every rank
    for t in terms:
        t._bind(grid, backs[0], None)
    arr, nt = _terms_c(terms), len(terms)
    phi = [b.alloc() for b in backs]
    b1 = [b.alloc() for b in backs]
    b2 = [b.alloc() for b in backs]
    for r, b in enumerate(backs):
        b.upload(phi[r], ic.vals[..., los[r]:los[r] + counts[r]])
        b.fill_ghosts(phi[r], 7)
    for r, b in enumerate(backs):
        b.halo_start(phi[r])
    for b in backs:
        b.halo_wait()
    B = L.GHOST + 1

    def stage(psi, phin, out, mode, cdt, t):
        for r, b in enumerate(backs):
            for m0, m1 in ((0, B), (counts[r] - B, counts[r])):
                b.stage_planes(arr, nt, psi[r], phin[r] if phin else None, out[r], None, mode, cdt, 0.0, t, m0, m1)
                b.fill_ghosts_planes(out[r], m0, m1)
            b.halo_start(out[r])
        for r, b in enumerate(backs):
            b.stage_planes(arr, nt, psi[r], phin[r] if phin else None, out[r], None, mode, cdt, 0.0, t, B, counts[r] - B)
            b.fill_ghosts_planes(out[r], B, counts[r] - B)
        for r, b in enumerate(backs):
            b.halo_wait()
            b.fill_ghosts(out[r], 1 << 2)

    stage(phi, None, b1, L.BASE_PSI, dt, 0.0)
    stage(b1, phi, b2, L.BASE_RK3_S2, 0.25 * dt, dt)
    stage(b2, phi, phi, L.BASE_RK3_S3, (2.0 / 3) * dt, 0.5 * dt)
    got = np.concatenate([b.download(phi[r]) for r, b in enumerate(backs)], axis=2)
    assert np.array_equal(got, want), np.abs(got - want).max()
    for b in backs:
        b.close()


@pytest.mark.parametrize("ndim,dtype,integ,world", [(2, "float64", "rk2", 3), (2, "float32", "rk3", 2), (3, "float32", "rk3", 3), (2, "float64", "fe", 4)])
def test_local_groups_in_two_dimensions_and_float32(lsm, ndim, dtype, integ, world):
    """The slab step of the library for the other storage type and dimension: 2-D slabs are rows, a "plane" is one padded
    row; float32 planes travel as 4-byte elements.  Mixed terms (no WENO5) so that the 64x8 / 256x1 light kernels serve the
    plane ranges too.  One thread per rank over an LSM_COMM_LOCAL group; bitwise against the single-device run."""
    import threading
    n = (40, 53) if ndim == 2 else (24, 20, 29)
    grid = lsm.CartesianGrid((-1,) * ndim, (1,) * ndim, n)
    dt = np.dtype(dtype)
    ic = lsm.MeshField(lambda x: np.sqrt(sum((x[d] - 0.1 * d) ** 2 for d in range(ndim))) - 0.55, grid, dtype=dt)
    I = {"rk3": lsm.RK3, "rk2": lsm.RK2, "fe": lsm.ForwardEuler}[integ]
    bc = (lsm.ExtrapolationBC(2),) + (lsm.NeumannBC(),) * (ndim - 1)
    mk = lambda **kw: lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.3), lsm.CurvatureTerm(-0.02), lsm.AdvectionTerm((0.5, -0.25, 0.4)[:ndim], lsm.Upwind())),
                                           ic=ic, bc=bc, integrator=I(), **kw)
    ref = mk()
    lsm.integrate_(ref, 0.02)
    want = ref.current_state().values()
    assert want.dtype == dt
    g = lsm.LocalGroup(world)
    got, errs = [None] * world, []

    def run(r):
        try:
            eq = mk(comm=g.rank(r))
            lsm.integrate_(eq, 0.02)
            got[r] = eq.current_state().values()
        except BaseException:   # noqa: BLE001 - reported by the main thread
            import traceback
            errs.append((r, traceback.format_exc()))
            g.abort()

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, errs
    full = np.concatenate(got, axis=ndim - 1)
    assert np.array_equal(full, want), np.abs(full.astype(np.float64) - want).max()


def _three_slabs(lsm):
    from lsm_amd.api import _bc_c, _normalize_bc
    from lsm_amd.backend import HipBackend
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (16, 16, 36))
    bcs = _normalize_bc(lsm.NeumannBC(), 3)
    backs = [HipBackend(grid._c(), _bc_c(bcs, 3, (r > 0, r < 2)), slab=(12 * r, 12)) for r in range(3)]
    HipBackend.comm_attach_local(backs)
    return backs


@pytest.mark.parametrize("how", ["abort", "leave"])
def test_a_failed_rank_fails_its_peers_instead_of_hanging_them(lsm, how):
    """include/lsm.h, "Failure": rank 1 of three never posts its exchange — it calls lsm_comm_abort, or is destroyed — while
    ranks 0 and 2 sit in lsm_halo_wait / lsm_allreduce_dt: they return LSM_ERR_COMM within a second, and every later call on
    their communicators fails the same way."""
    import threading
    import time
    from lsm_amd import _lib as L
    backs = _three_slabs(lsm)
    fields = [b.alloc() for b in backs]
    out, t_done = {}, {}

    def waiter(r, what):
        try:
            if what == "halo":
                backs[r].halo_start(fields[r])
                backs[r].halo_wait()
            else:
                backs[r].allreduce_dt(0.1)
            out[r] = "returned"
        except L.LsmCommError as e:
            out[r] = str(e)
        t_done[r] = time.perf_counter()

    ts = [threading.Thread(target=waiter, args=(0, "halo"), daemon=True), threading.Thread(target=waiter, args=(2, "halo"), daemon=True)]
    for t in ts:
        t.start()
    time.sleep(0.3)                      # both are blocked now: rank 1 has not posted
    assert not out
    t0 = time.perf_counter()
    if how == "abort":
        backs[1].comm_abort()
    else:
        backs[1].close()
    for t in ts:
        t.join(5)
    assert set(out) == {0, 2} and all("aborted by rank 1" in v or "rank 1 has left" in v for v in out.values()), out
    assert max(t_done.values()) - t0 < 1.0
    # the communicators stay failed
    for r in (0, 2):
        with pytest.raises(L.LsmCommError):
            backs[r].allreduce_dt(0.5)
        with pytest.raises(L.LsmCommError):
            backs[r].halo_exchange(fields[r])
    for b in backs:
        b.close()


def test_finished_rank_may_be_destroyed_while_a_neighbour_still_orders_its_stream(lsm):
    """The group owns the exchange streams and events: rank 0 completes an exchange and is destroyed at once; rank 1, which
    posted the same exchange, still waits for it successfully afterwards (its stream is ordered behind events that must
    outlive rank 0's handle), and only its NEXT exchange reports the missing peer."""
    from lsm_amd import _lib as L
    from lsm_amd.api import _bc_c, _normalize_bc
    from lsm_amd.backend import HipBackend
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (16, 16, 24))
    bcs = _normalize_bc(lsm.NeumannBC(), 3)
    b0 = HipBackend(grid._c(), _bc_c(bcs, 3, (False, True)), slab=(0, 12))
    b1 = HipBackend(grid._c(), _bc_c(bcs, 3, (True, False)), slab=(12, 12))
    HipBackend.comm_attach_local([b0, b1])
    f0, f1 = b0.alloc(), b1.alloc()
    f0.fill_(1.0)
    f1.fill_(2.0)
    b0.halo_start(f0)
    b1.halo_start(f1)
    b0.halo_wait()
    b0.sync()
    b0.close()                           # rank 0 is gone; its planes were read by rank 1's copies already (close waits for them)
    b1.halo_wait()                       # complete exchange: succeeds
    b1.sync()
    lay = b1.lay
    sl = int(lay.stride[2])
    ghosts = f1[:3 * sl].cpu().numpy()   # the three ghost planes below rank 1's slab hold rank 0's values
    assert (ghosts == 1.0).all()
    with pytest.raises(L.LsmCommError, match="rank 0 has left"):
        b1.halo_exchange(f1)
    b1.close()


def test_comm_timeout_in_a_fresh_process(lsm, tmp_path):
    """LSM_COMM_TIMEOUT_MS (read once per process): a rank that waits for a peer which never posts gives up by itself."""
    import os
    import subprocess
    import sys
    code = '''
import sys, time
sys.path.insert(0, %r)
import lsm_amd as lsm
from lsm_amd import _lib as L
from lsm_amd.api import _bc_c, _normalize_bc
from lsm_amd.backend import HipBackend
grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), (16, 16, 24))
bcs = _normalize_bc(lsm.NeumannBC(), 3)
b0 = HipBackend(grid._c(), _bc_c(bcs, 3, (False, True)), slab=(0, 12))
b1 = HipBackend(grid._c(), _bc_c(bcs, 3, (True, False)), slab=(12, 12))
HipBackend.comm_attach_local([b0, b1])
t0 = time.perf_counter()
try:
    b0.halo_exchange(b0.alloc())
    print("RETURNED")
except L.LsmCommError as e:
    print("COMM_ERROR %%.2f %%s" %% (time.perf_counter() - t0, e))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LSM_COMM_TIMEOUT_MS="400")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "COMM_ERROR" in r.stdout and "timed out" in r.stdout, (r.stdout, r.stderr[-2000:])
    assert 0.3 < float(r.stdout.split()[1]) < 3.0


def _run_ranks(lsm, world, body, timeout=600):
    """One thread per rank over an in-process group; returns what body(rank, comm) returned, rank by rank."""
    import threading
    import traceback
    g = lsm.LocalGroup(world)
    got, errs = [None] * world, []

    def run(r):
        try:
            got[r] = body(r, g.rank(r))
        except BaseException:   # noqa: BLE001 - reported by the main thread
            errs.append((r, traceback.format_exc()))
            g.abort()

    ts = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout)
    assert not errs, errs
    return got


@pytest.mark.parametrize("case", ["upwind_periodic", "curvature", "normal_motion", "eikonal_periodic", "nm_curv_mixed"])
@pytest.mark.parametrize("integ", ["rk3", "fe"])
def test_slab_exchange_sends_only_the_planes_the_stencils_read(lsm, case, integ):
    """the exchange depth is the reach of the step's stencils — 1 plane for upwind and curvature
    (src/levelsetops.jl:234-244), 2 for the ENO2 terms (src/levelsetterms.jl:156-170), 3 for WENO5 (src/derivatives.jl:89-121).
    Slab steps through the library (lsm_advance_* on LOCAL groups of 3 ranks, periodic ring included) with depth-1 and depth-2
    term lists equal the single-device run bit for bit, over several steps (the ghost planes beyond the depth stay stale and
    must never be read)."""
    from lsm_amd import _lib as L
    n, world = (22, 18, 41), 3
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), n)
    ic = lsm.MeshField(lambda x: np.sqrt((x[0] - 0.45) ** 2 + (x[1] - 0.5) ** 2 + (x[2] - 0.4) ** 2) - 0.27, grid)
    terms, bc = {
        "upwind_periodic": (lambda: (lsm.AdvectionTerm((0.6, -0.3, 0.9), lsm.Upwind()),), lsm.PeriodicBC()),
        "curvature": (lambda: (lsm.CurvatureTerm(-0.05),), lsm.ExtrapolationBC(1)),
        "normal_motion": (lambda: (lsm.NormalMotionTerm(0.8),), lsm.NeumannBC()),
        "eikonal_periodic": (lambda: (lsm.EikonalReinitializationTerm(),), lsm.PeriodicBC()),
        "nm_curv_mixed": (lambda: (lsm.NormalMotionTerm(-0.5), lsm.CurvatureTerm(-0.03)), (lsm.NeumannBC(), lsm.ExtrapolationBC(2), lsm.SymmetryBC())),
    }[case]
    I = {"rk3": lsm.RK3, "fe": lsm.ForwardEuler}[integ]
    mk = lambda **kw: lsm.LevelSetEquation(terms=terms(), ic=ic, bc=bc, integrator=I(), **kw)
    ref = mk()
    tf = 4.2 * 0.5 * ref.compute_cfl(0.0)                 # a handful of steps, the last one cut
    lsm.integrate_(ref, tf)
    want = ref.current_state().values()

    def body(r, comm):
        eq = mk(comm=comm)
        assert eq.lib_comm and eq.backend.comm_info() == (r, world, L.COMM_LOCAL)
        lsm.integrate_(eq, tf)
        return eq.current_state().values()

    got = np.concatenate(_run_ranks(lsm, world, body), axis=2)
    assert np.array_equal(got, want), float(np.abs(got - want).max())


def test_slab_step_refreshes_its_ghosts_when_the_term_list_gets_deeper(lsm):
    """A step leaves ϕ with as many valid ghost planes as its own stencils read.  A host that then steps the same field with a
    deeper term list (curvature: 1 plane, then WENO5 advection: 3) must get the single-device result: the slab step notices
    and refreshes all layers first."""
    n, world = (20, 16, 36), 3
    grid = lsm.CartesianGrid((0, 0, 0), (1, 1, 1), n)
    ic = lsm.MeshField(lambda x: np.sqrt((x[0] - 0.5) ** 2 + (x[1] - 0.5) ** 2 + (x[2] - 0.45) ** 2) - 0.3, grid)
    shallow = lambda: (lsm.CurvatureTerm(-0.04),)
    deep = lambda: (lsm.AdvectionTerm((0.5, 0.2, -0.8), lsm.WENO5()),)

    # Δt of the two phases: node-independent (constant coefficients), the same on every rank
    dt1 = 0.5 * lsm.LevelSetEquation(terms=shallow(), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3()).compute_cfl(0.0)
    dt2 = 0.5 * lsm.LevelSetEquation(terms=deep(), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3()).compute_cfl(0.0)

    def two_phases(**kw):
        eq = lsm.LevelSetEquation(terms=shallow(), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3(), **kw)
        eq._advance(0.0, dt1)
        eq._advance(dt1, dt1)
        # the SAME handle and field step on with the deep list, as a host that edits the equation's terms would
        eq.terms = deep()
        for t in eq.terms:
            t._bind(grid, eq.backend, eq.slab)
        eq._advance(2 * dt1, dt2)
        return eq.current_state().values()

    want = two_phases()
    got = np.concatenate(_run_ranks(lsm, world, lambda r, comm: two_phases(comm=comm)), axis=2)
    assert np.array_equal(got, want), float(np.abs(got - want).max())


def test_eight_rank_local_group_on_the_scaling_runs_plane_shape(lsm):
    """The 8-rank decomposition the driver's scaling run uses (1024 x 1024 planes, `bench.py --gpus 8`), rehearsed on one device:
    eight rank threads over an in-process group, 12 planes each (the boundary-first order needs 2·(3+1)+1), two RK3 steps of the
    headline equation through lsm_advance_rk3 — bit for bit the single-device step, the all-reduced Δt included."""
    n, world = (1024, 1024, 96), 8
    h = 1.0 / (n[0] - 1)
    grid = lsm.CartesianGrid((0.0, 0.0, 0.0), tuple((k - 1) * h for k in n), n)
    ic = lsm.LazyMeshField(lambda x: np.sqrt((x[0] - 0.35) ** 2 + (x[1] - 0.35) ** 2 + (x[2] - 0.04) ** 2) - 0.15, grid)

    def steps(**kw):
        eq = _equation(lsm, grid, ic, **kw)
        tc, dts = 0.0, []
        for _ in range(2):
            eq._update_terms(eq.state, tc)
            dt = 0.5 * eq.compute_cfl(tc)
            eq._advance(tc, dt)
            tc += dt
            dts.append(dt)
        return eq, dts

    ref, want_dts = steps()
    want = ref.current_state().values()
    del ref

    def body(r, comm):
        eq, dts = steps(comm=comm)
        assert eq.lib_comm and dts == want_dts, (dts, want_dts)
        return eq.current_state().values()

    got = np.concatenate(_run_ranks(lsm, world, body), axis=2)
    assert got.shape == want.shape and np.array_equal(got, want), float(np.abs(got - want).max())

#!/usr/bin/env python
"""Timing of the non-headline BASELINE configs through the host API (GPU box), and a
FETCH_SIZE calibration run (mode 'calib': kernels with exactly known HBM traffic)."""
import json
import sys
import time

import numpy as np
import torch

import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import lsm_amd as lsm


def timed(eq, steps, warmup=2):
    tc = 0.0
    def one(tc):
        eq._update_terms(eq.state, tc)
        dt = eq.integrator.cfl * eq.compute_cfl(tc)
        eq._advance(tc, dt)
        return tc + dt
    for _ in range(warmup):
        tc = one(tc)
    # the device leaves its idle power state over ≈50 ms of load: untimed steps until 80 ms have gone by
    torch.cuda.synchronize()
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.08:
        for _ in range(4):
            tc = one(tc)
        torch.cuda.synchronize()
    # the step is timed WITHOUT the event pairs around its stage launches (an event costs the stream ≈3.7 µs: 22 µs of a six-event step);
    # the stage kernels' own duration comes from a second pass with them
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tc = one(tc)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    eq.backend.profile_enable(True)
    for _ in range(steps):
        tc = one(tc)
    n, ms = eq.backend.profile_read()
    eq.backend.profile_enable(False)
    return el / steps * 1e3, ms / max(n, 1), n // steps


def config2(n=2048, steps=20):
    grid = lsm.CartesianGrid((-1.5, -1.5), (1.5, 1.5), (n, n))
    ic = lsm.MeshField(lambda x: np.maximum(np.hypot(x[0] + 0.75, x[1]) - 0.5,
                                            -np.maximum(np.abs(x[0] + 0.75) - 0.1, np.abs(x[1] + 0.25) - 0.5)), grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()),), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3())
    ms, kms, nl = timed(eq, steps)
    re = lsm.LevelSetEquation(terms=(lsm.EikonalReinitializationTerm(eq.current_state()),), ic=eq.current_state(), bc=lsm.NeumannBC(),
                              integrator=lsm.RK3())
    ms2, kms2, nl2 = timed(re, steps)
    return {"config": f"2D {n}^2 Zalesak", "advect_ms_per_step": ms, "advect_stage_ms": kms, "advect_Mcells_s": n * n / ms / 1e3,
            "reinit_ms_per_step": ms2, "reinit_stage_ms": kms2, "reinit_Mcells_s": n * n / ms2 / 1e3}


def config3(n=512, steps=10):
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    ic = lsm.LazyMeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.1), lsm.CurvatureTerm(-0.1)), ic=ic, bc=lsm.ExtrapolationBC(2), integrator=lsm.RK3())
    ms, kms, nl = timed(eq, steps)
    return {"config": f"3D {n}^3 mean-curvature flow (NormalMotion+Curvature)", "ms_per_step": ms, "stage_ms": kms,
            "Mcells_s": n ** 3 / ms / 1e3, "GBs_algorithmic": n ** 3 * 64 / 3 / kms / 1e6}


def config5(n=768, steps=10, nlayers=3):
    """BASELINE config 5 on ONE device: float32 storage, sphere, rigid rotation (WENO5) + curvature, RK3, narrow band."""
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    f = lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5
    terms = lambda: (lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.01))
    res = {"config": f"3D {n}^3 float32 narrow band (nlayers {nlayers}), rotation WENO5 + curvature, RK3"}
    for dt in (np.float32, np.float64):
        vals = lsm.LazyMeshField(f, grid).local_values(None).astype(dt)
        eq = lsm.LevelSetEquation(terms=terms(), ic=lsm.NarrowBandMeshField(lsm.MeshField(vals, grid, dtype=dt), nlayers=nlayers),
                                  bc=lsm.NeumannBC(), integrator=lsm.RK3())
        del vals
        tc = 0.0

        def one(tc):
            eq._update_terms(eq.state, tc)
            step = eq.integrator.cfl * eq.compute_cfl(tc)
            eq._advance(tc, step)
            eq.update_band()
            return tc + step
        for _ in range(2):
            tc = one(tc)
        torch.cuda.synchronize()
        t_pre = time.perf_counter()                      # out of the idle power state first (see timed())
        while time.perf_counter() - t_pre < 0.08:
            for _ in range(4):
                tc = one(tc)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tc = one(tc)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        c = eq.state.active_count()
        res[np.dtype(dt).name] = {"ms_per_step": round(ms, 3), "active_nodes": c, "active_fraction": round(c / n ** 3, 4),
                                  "Mcells_s_grid": round(n ** 3 / ms / 1e3, 1), "Mcells_s_active": round(c / ms / 1e3, 1)}
        del eq
        torch.cuda.empty_cache()
    return res


def config5r(n=768, steps=40, nlayers=3, every=10):
    """Config 5's float32 band with reinitialize! every `every` steps — the workflow of the reference's own narrow-band tests
    (a posthook that reinitialises, test/test-narrow-band.jl:355-395).  `steps` is rounded to a multiple of `every`."""
    import warnings
    steps = max(every, steps // every * every)
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    f = lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5
    vals = lsm.LazyMeshField(f, grid).local_values(None).astype(np.float32)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.01)),
                              ic=lsm.NarrowBandMeshField(lsm.MeshField(vals, grid, dtype=np.float32), nlayers=nlayers),
                              bc=lsm.NeumannBC(), integrator=lsm.RK3())
    del vals
    state = {"tc": 0.0, "k": 0, "reinit_s": 0.0}

    def one(timed_reinit=False):
        tc = state["tc"]
        eq._update_terms(eq.state, tc)
        step = eq.integrator.cfl * eq.compute_cfl(tc)
        eq._advance(tc, step)
        eq.update_band()
        state["tc"] = tc + step
        state["k"] += 1
        if state["k"] % every == 0:
            if timed_reinit:
                torch.cuda.synchronize()
                t = time.perf_counter()
            lsm.reinitialize_(eq)
            if timed_reinit:
                torch.cuda.synchronize()
                state["reinit_s"] += time.perf_counter() - t
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(every):
            one()
        torch.cuda.synchronize()
        t_pre = time.perf_counter()                      # out of the idle power state first (see timed())
        while time.perf_counter() - t_pre < 0.08:
            for _ in range(every):
                one()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        for _ in range(steps):                            # a second pass with a synchronisation either side of every reinitialize!
            one(True)
    c = eq.state.active_count()
    return {"config": f"3D {n}^3 float32 narrow band (nlayers {nlayers}), rotation WENO5 + curvature, RK3, reinitialize! every {every} steps",
            "ms_per_step": round(ms, 3), "reinitialize_ms": round(state["reinit_s"] / (steps // every) * 1e3, 3), "reinitialize_every": every,
            "active_nodes": c, "Mcells_s_grid": round(n ** 3 / ms / 1e3, 1), "Mcells_s_active": round(c / ms / 1e3, 1)}


def upwind(n=512, steps=10):
    """An HBM-bound member of the family: first-order upwind advection, ForwardEuler, both storage types."""
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    res = {"config": f"3D {n}^3 upwind advection, ForwardEuler"}
    for dt in (np.float64, np.float32):
        ic = lsm.LazyMeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, grid, dtype=dt)
        eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm((1.0, 0.5, -0.25), lsm.Upwind()),), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.ForwardEuler())
        ms, kms, nl = timed(eq, steps)
        es = np.dtype(dt).itemsize
        res[np.dtype(dt).name] = {"ms_per_step": round(ms, 3), "stage_ms": round(kms, 4), "Mcells_s": round(n ** 3 / ms / 1e3, 1),
                                  "GBs_algorithmic_stage": round(n ** 3 * 2 * es / kms / 1e6, 1)}
        del eq
        torch.cuda.empty_cache()
    return res


def terms(n=512, steps=8):
    """Every single term and the fused pairs on a 512^3 sphere, ForwardEuler (one stage per step: read ψ, write): the
    stage kernel's duration and algorithmic GB/s (16 B per node) — which members of the family are HBM-bound."""
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    ic = lsm.LazyMeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, grid)
    cases = {
        "upwind adv (const)": (lsm.AdvectionTerm((1.0, 0.5, -0.25), lsm.Upwind()),),
        "WENO5 adv (const)": (lsm.AdvectionTerm((1.0, 0.5, -0.25), lsm.WENO5()),),
        "WENO5 adv (rotation)": (lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()),),
        "NormalMotion (const)": (lsm.NormalMotionTerm(0.1),),
        "Curvature (const)": (lsm.CurvatureTerm(-0.1),),
        "Eikonal (current sign)": (lsm.EikonalReinitializationTerm(),),
        "NormalMotion + Curvature": (lsm.NormalMotionTerm(0.1), lsm.CurvatureTerm(-0.1)),
        "WENO5 adv (rotation) + Curvature": (lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.01)),
        "WENO5 adv (vortex) + Eikonal": (lsm.AdvectionTerm(lsm.vortex_deformation(grid), lsm.WENO5()), lsm.EikonalReinitializationTerm()),
    }
    res = {"config": f"3D {n}^3 single stages (ForwardEuler)"}
    for name, tm in cases.items():
        eq = lsm.LevelSetEquation(terms=tm, ic=ic, bc=lsm.NeumannBC(), integrator=lsm.ForwardEuler())
        ms, kms, nl = timed(eq, steps)
        res[name] = {"stage_ms": round(kms, 4), "GBs_algorithmic": round(n ** 3 * 16 / kms / 1e6, 1), "frac_of_8TBs": round(n ** 3 * 16 / kms / 1e6 / 8000, 3)}
        del eq
        torch.cuda.empty_cache()
    return res


def calib(n=512):
    """Known-traffic kernels for the FETCH_SIZE/WRITE_SIZE calibration (run under rocprofv3 --pmc):
    extrema reads n^3*8 B with 8-byte-per-lane loads; eikonal_sign reads and writes n^3*8 B."""
    grid = lsm.CartesianGrid((-1, -1, -1), (1, 1, 1), (n, n, n))
    ic = lsm.LazyMeshField(lambda x: np.sqrt(x[0] ** 2 + x[1] ** 2 + x[2] ** 2) - 0.5, grid)
    eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.1),), ic=ic, bc=lsm.NeumannBC())
    b = eq.backend
    s0 = b.alloc()
    for _ in range(3):
        b.extrema(eq.state.buf)
        b.eikonal_sign(eq.state.buf, s0)
    torch.cuda.synchronize()
    return {"calib_bytes_read_each": n ** 3 * 8}


def run(which, steps=None, warmup=None):
    """`python bench.py --config {2,3,5,5r}`: one JSON object in bench.py's vocabulary for a non-headline BASELINE config."""
    kw = {} if steps is None else {"steps": steps}
    r = {"2": config2, "3": config3, "5": config5, "5r": config5r}[which](**kw)
    out = {"metric": "Mcells/s per RK3 step", "unit": "Mcells/s", "n_gpus": 1, "higher_is_better": True, "data": "synthetic",
           "dtype": "f32 storage / f64 arithmetic" if which in ("5", "5r") else "f64", "config": {"workload": r["config"]}, "detail": r}
    if which == "2":
        out["value"], out["ms_per_step"] = round(r["advect_Mcells_s"], 1), round(r["advect_ms_per_step"], 4)
    elif which == "3":
        out["value"], out["ms_per_step"] = round(r["Mcells_s"], 1), round(r["ms_per_step"], 4)
        a = r["GBs_algorithmic"]
        out["roofline"] = {"bound": "hbm", "achieved": round(a, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(a / 8000.0, 4), "traffic": None,
                           "kernel": "stage_kernel<3,NormalMotion,Curvature>", "avg_launch_ms": round(r["stage_ms"], 4)}
    elif which == "5r":
        out["value"], out["ms_per_step"] = r["Mcells_s_grid"], r["ms_per_step"]
    else:
        out["value"], out["ms_per_step"] = r["float32"]["Mcells_s_grid"], r["float32"]["ms_per_step"]
        # algorithmic bytes of a band step (SURVEY.md §8d with s = 4): 8 s = 32 B per ACTIVE node per RK3 step
        a = r["float32"]["active_nodes"] * 32.0 / (r["float32"]["ms_per_step"] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "bound_actual": "latency / instruction issue of sparse tiles (DESIGN.md §7.1)", "achieved": round(a, 1), "peak": 8000.0,
                           "unit": "GB/s", "frac": round(a / 8000.0, 4), "traffic": None,
                           "note": "32 B per active node per RK3 step; the whole step (three stages, three halo fills, update_band!, Δt) is timed"}
    return out


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    out = []
    if mode in ("all", "2"):
        out.append(config2())
    if mode in ("all", "3"):
        out.append(config3())
    if mode in ("all", "5"):
        out.append(config5())
    if mode == "5r":
        out.append(config5r())
    if mode in ("all", "upwind"):
        out.append(upwind())
    if mode == "terms":
        out.append(terms())
    if mode == "calib":
        out.append(calib())
    print(json.dumps(out, indent=1))

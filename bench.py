#!/usr/bin/env python
"""bench.py — Mcells/s per RK3 step of the headline equation (BASELINE.json):
WENO5 advection by the vortex-deformation field + Eikonal reinitialisation, fp64, RK3, NeumannBC.

  python bench.py --gpus N --steps K --warmup W

N = 1: 512³ on one MI355X (the configuration the metric is quoted on).
N > 1 (launched by torch.distributed.run, one rank per GPU): the grid is 1024 × 1024 × 128·N,
slab-decomposed along the last dimension (every rank owns 1024 × 1024 × 128 = 512³ cells: weak
scaling), ghost planes exchanged over RCCL/xGMI after every stage and Δt all-reduced INSIDE libhiplsm
(lsm_comm_attach_rccl; torch.distributed only carries the RCCL unique id to the ranks).

`--transport local --gpus N` rehearses the N-rank run on ONE device: N rank threads of this process, every rank a handle of an
in-process group (lsm_comm_attach_local: peer copies stand in for xGMI), through the same code below — slab construction, the
self-check of the two stage orders, the timed loop, the JSON line.  It measures the decomposition's own cost, not a transfer.

A "step" is one pass of the reference's step loop body (src/timestepping.jl:104-116):
compute_cfl + the 3 fused RK3 stage kernels + ghost fills (hooks = identity).  Inputs are resident
in HBM when the timed region starts.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this pool (see task notes)
os.environ.setdefault("LSM_COMM_TIMEOUT_MS", "30000")     # a rank that waits for a peer gives up after 30 s (LSM_ERR_COMM) instead of hanging the run
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (≈6.3 TB/s achievable)
FP64_PEAK_TFLOPS = 78.6


def build_equation(lsm, n, comm, device, mode):
    h = 1.0 / (n[0] - 1)
    grid = lsm.CartesianGrid((0.0, 0.0, 0.0), tuple((k - 1) * h for k in n), n)   # uniform spacing in every run
    ic = lsm.LazyMeshField(lambda x: np.sqrt((x[0] - 0.35) ** 2 + (x[1] - 0.35) ** 2 + (x[2] - 0.35) ** 2) - 0.15, grid)
    vel = lsm.vortex_deformation(grid)
    eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(vel, lsm.WENO5()), lsm.EikonalReinitializationTerm()), ic=ic,
                              bc=lsm.NeumannBC(), integrator=lsm.RK3(), comm=comm, device=device, mode=mode)
    return eq, grid, vel


def one_step(eq, tc):
    """Loop body of _integrate! (src/timestepping.jl:104-116) without hooks."""
    eq._update_terms(eq.state, tc)
    dt = eq.integrator.cfl * eq.compute_cfl(tc)
    eq._advance(tc, dt)
    return tc + dt


def cpu_baseline(n_sample, threads, steps=1):
    """The reference-faithful CPU oracle timed on a bounded sample of the same workload
    (kind 'port': the reference is Julia and cannot run here)."""
    from oracle import oracle as orc
    n = (n_sample,) * 3
    g = orc.Grid((0, 0, 0), (1, 1, 1), n)
    bc = orc.make_bc("neumann", 3)
    x, y, z = g.coords()
    s2 = lambda a: np.sin(np.pi * a) * np.sin(np.pi * a)
    s = lambda a: np.sin(2 * np.pi * a)
    terms = [orc.advection(orc.separable([[2 * s2(x), s(y), s(z)], [-s(x), s2(y), s(z)], [-s(x), s(y), s2(z)]], orc.TIME_COS, 3.0)),
             orc.eikonal()]
    phi = g.sample(lambda X, Y, Z: np.sqrt((X - 0.35) ** 2 + (Y - 0.35) ** 2 + (Z - 0.35) ** 2) - 0.15)
    orc.set_threads(threads)
    t0 = time.perf_counter()
    done, _, _ = orc.integrate(orc.RK3, g, bc, phi, terms, 1.0, max_steps=steps)
    el = time.perf_counter() - t0
    orc.set_threads(1)
    return n_sample ** 3 * done / el / 1e6, el


def committed_profile(n, world, mode):
    """The rocprofv3 PMC summary committed for THIS build of the kernels and THIS workload, or None.
    profiles/r4/pmc_per_dispatch.json (an earlier round's for a build that still has that round's kernel sources) records the sha256 of the kernel sources it was measured on (the GPU box has no
    .git, so the key is the source text, not a commit) and the workload; anything else gets no traffic figure."""
    import lsm_amd
    for rnd in ("r4", "r3", "r2"):
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", rnd, "pmc_per_dispatch.json")))
        except Exception:
            continue
        if pj.get("csrc_sha256") == lsm_amd._lib.source_hash() and pj.get("grid") == [n, n, n] and pj.get("n_gpus") == world and pj.get("mode") == mode:
            pj["_file"] = f"profiles/{rnd}/pmc_per_dispatch.json"
            return pj
    return None


class DistCtx:
    """The ranks of a torch.distributed.run launch (one process per GPU, RCCL)."""
    transport = "rccl"

    def __init__(self, torch, dist):
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.device = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(self.device)
        self.comm = None
        if self.world > 1:
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            self.comm = dist.group.WORLD

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


class LocalCtx:
    """Rank `r` of `world` rank threads on one device (api.LocalGroup: lsm_comm_attach_local)."""
    transport = "local"

    def __init__(self, torch, group, r, device=0):
        self.torch, self.group = torch, group
        self.world, self.rank, self.device = group.world, r, device
        self.comm = group.rank(r)

    def barrier(self):
        self.torch.cuda.synchronize()
        self.group.exchange(self.rank, None)
        self.torch.cuda.synchronize()

    def gather(self, obj):
        return self.group.exchange(self.rank, obj)

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--prewarm", type=int, default=10, help="untimed steps before the warm-up steps: the device's ramp out of idle (set-up)")
    ap.add_argument("--n", type=int, default=512, help="cells per side of the per-GPU 512³-equivalent workload")
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=256, help="side of the single-thread oracle sample")
    ap.add_argument("--cpu-full", action="store_true", help="SURVEY.md §8d's full CPU sample: 10 steps at 256^3 on one core, 2 at 512^3 on all (minutes)")
    ap.add_argument("--profile-every", type=int, default=4, help="HIP-event pairs around every N-th stage launch of the timed region (1 = all)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "local"],
                    help="rccl: one process per GPU (torch.distributed.run); local: --gpus N rank THREADS on device 0 over the in-process transport (rehearsal)")
    ap.add_argument("--config", default="headline", choices=["headline", "2", "3", "5", "5r"],
                    help="headline = BASELINE config 4's equation at 512^3 on one GPU (the metric); 2, 3, 5: the other single-GPU BASELINE configs; 5r: config 5 with reinitialize! every 10 steps (tools/configs.py)")
    args = ap.parse_args()
    if args.config != "headline":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import configs
        print(json.dumps(configs.run(args.config, steps=args.steps, warmup=args.warmup)))
        return

    import torch
    import torch.distributed as dist

    import lsm_amd as lsm

    if args.transport == "local" and args.gpus > 1:
        import threading
        torch.cuda.set_device(0)
        group = lsm.LocalGroup(args.gpus)
        errs = [None] * args.gpus

        def rank_main(r):
            try:
                torch.cuda.set_device(0)
                run(args, lsm, torch, LocalCtx(torch, group, r))
            except BaseException as e:   # noqa: BLE001 - a rank that dies must not leave the others at a barrier
                errs[r] = e
                group.abort()
        threads = [threading.Thread(target=rank_main, args=(r,), name=f"rank{r}") for r in range(args.gpus)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        first = next((e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or next((e for e in errs if e is not None), None)
        if first is not None:
            raise first
        return
    ctx = DistCtx(torch, dist)
    if args.gpus > 1 and ctx.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={ctx.world})")
    run(args, lsm, torch, ctx)
    ctx.close()


def run(args, lsm, torch, ctx):
    """One rank of the benchmark (the only rank at N = 1)."""
    world, rank, local_rank, comm = ctx.world, ctx.rank, ctx.device, ctx.comm
    barrier = ctx.barrier

    if world == 1:
        n = (args.n, args.n, args.n)
        workload = f"3D {args.n}^3 vortex-deformation WENO5 advection + EikonalReinitializationTerm, RK3, NeumannBC, fp64"
    else:
        side = 2 * args.n
        n = (side, side, (args.n // 4) * world)
        workload = (f"3D {n[0]}x{n[1]}x{n[2]} vortex-deformation WENO5 advection + EikonalReinitializationTerm, RK3, NeumannBC, "
                    f"fp64, slab-decomposed over {world} GPUs ({side}x{side}x{args.n // 4} per GPU)")
    cells = n[0] * n[1] * n[2]

    eq, grid, vel = build_equation(lsm, n, comm, local_rank, args.mode)

    # multi-GPU self-check (outside the timed region): one step with the exchange overlapped behind the
    # interior update must equal one step with the plain stage -> halo sequence BIT FOR BIT; otherwise
    # fall back to the plain sequence and say so.
    overlap_note = "n/a"
    overlap_fallback = None
    if world > 1:
        # the slab step runs inside the library (lsm_advance_rk3 on a handle with an RCCL communicator attached:
        # boundary planes first, exchange overlapped behind the interior update)
        # (if RCCL cannot be opened by the library the equation falls back, with a warning, to the same exchange driven
        # stage by stage over torch.distributed — reported as config.exchange)
        def set_overlap(on):
            if eq.lib_comm:
                eq.backend.comm_set_overlap(on)
            else:
                eq.overlap = on
        keep = eq.state.buf.clone()
        # the library's communicator has met librccl with more than one rank on no box of this pool yet: if its first step fails
        # on any rank (LSM_ERR_COMM after LSM_COMM_TIMEOUT_MS at the latest — nothing in it blocks for ever), every rank rebuilds
        # the equation with the stage-by-stage exchange over torch.distributed (same planes, same order, same results)
        err = None
        try:
            one_step(eq, 0.0)
            one_step(eq, 0.0)        # the second step's Δt all-reduce waits (with a timeout) behind the first step's exchanges
        except Exception as e:   # noqa: BLE001 - reported in config.exchange
            err = repr(e)
        errs = ctx.gather(err)
        if any(errs) and ctx.transport == "local":
            raise RuntimeError(f"the in-process slab step failed: {next(x for x in errs if x)}")      # no other exchange to fall back to
        if any(errs):
            os.environ["LSM_LIB_COMM"] = "0"
            del eq, keep
            torch.cuda.empty_cache()
            eq, grid, vel = build_equation(lsm, n, comm, local_rank, args.mode)
            overlap_fallback = next(x for x in errs if x)
        else:
            eq.state.buf.copy_(keep)
            eq.state.ghosts_dirty = True
        keep = eq.state.buf.clone()
        one_step(eq, 0.0)
        a_res = eq.state.buf.clone()
        eq.state.buf.copy_(keep)
        eq.state.ghosts_dirty = True
        set_overlap(False)
        one_step(eq, 0.0)
        # the nodes of the slab, not its padded buffer: in FAST mode the ghost rows of dimensions 1 and 2 are never read (the loads are
        # redirected) nor written inside a step, and the two orders fill different never-read corners of the ghost planes
        lay = eq.backend.lay
        nd = len(n)
        view = lambda t: torch.as_strided(t, tuple(int(lay.n[d]) for d in range(nd)), tuple(int(lay.stride[d]) for d in range(nd)), int(lay.origin))
        ok = all(ctx.gather(bool(torch.equal(view(a_res), view(eq.state.buf)))))
        eq.state.buf.copy_(keep)
        eq.state.ghosts_dirty = True
        set_overlap(ok)
        overlap_note = "on (self-check passed)" if ok else "off (self-check mismatch)"
        del keep, a_res

    tc = 0.0
    # The device leaves its idle power state over the first ≈50 ms of load (per-step times from a cold start at 512³:
    # 9.0, 4.4, 4.1, 3.9, 3.7, 3.6, 3.6, 3.6, 3.5 ... 3.45 ms from the 13th step on).  Part of set-up, reported in `config`:
    # a few untimed steps of the same equation BEFORE the W warm-up steps the command line asks for, so that a small W
    # measures the steady state and not the ramp.  --prewarm 0 switches it off.
    for _ in range(args.prewarm):
        tc = one_step(eq, tc)
    for _ in range(args.warmup):
        tc = one_step(eq, tc)
    # HIP events around every 4th stage launch, on the stream the kernels run on (an event costs the stream ≈3.7 µs: six per step would be
    # 0.6 % of it; the period is coprime to the 3 — or, slab-decomposed, 9 — stage launches of a step, so every kind is sampled alike)
    eq.backend.profile_enable(args.profile_every)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tc = one_step(eq, tc)
    barrier()
    el = time.perf_counter() - t0
    n_launch, stage_ms = eq.backend.profile_read()
    eq.backend.profile_enable(False)
    el = max(ctx.gather(el))

    ms_per_step = el / args.steps * 1e3
    mcells = cells * args.steps / el / 1e6
    # algorithmic bytes: RK3 step = 8·sizeof(T) = 64 B/node over 3 stage launches
    local_cells = cells // world
    bytes_per_launch = local_cells * 64.0 / 3.0
    avg_launch_s = stage_ms / max(1, n_launch) * 1e-3
    # = bytes_per_launch / avg launch duration at N = 1 (3 launches per step); in slab mode a stage is
    # split into boundary + interior launches, so use the step's bytes over the step's stage time
    achieved = local_cells * 64.0 * args.steps / (stage_ms * 1e-3) / 1e9 if n_launch else 0.0
    # fp64 VALU view of the same kernel (the binding resource): ≈1.0 kflop/node-stage
    out = {
        "metric": "Mcells/s per RK3 step (WENO5 advect+reinit), 512^3 fp64; HBM GB/s vs peak",
        "value": round(mcells, 2),
        "unit": "Mcells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "grid": list(n), "cells": cells, "integrator": "RK3", "cfl": 0.5,
                   "arithmetic_mode": args.mode, "parallelism": f"slab{world}" if world > 1 else "single",
                   "halo_overlap": overlap_note, "prewarm_steps": args.prewarm,
                   "exchange": "n/a" if world == 1 else ("libhiplsm in-process transport (lsm_advance_rk3 on a slab)" if eq.lib_comm and ctx.transport == "local" else
                                                             "libhiplsm RCCL (lsm_advance_rk3 on a slab)" if eq.lib_comm else
                                                             "torch.distributed fallback" + (f" (library communicator failed: {overlap_fallback})" if overlap_fallback else ""))},
        # `bound` names the roofline BASELINE.json asks this metric to be priced against (HBM); the resource that actually holds
        # this kernel is the fp64 vector pipe (bound_actual; the roofline_compute block below)
        "roofline": {"bound": "hbm", "bound_actual": "fp64_valu", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "stage_kernel<3,WENO5 adv,Eikonal> (fused RK3 stage)",
                     "stage_launches": int(n_launch), "launches_timed": f"every {args.profile_every}th (HIP events on the kernels' stream)" if args.profile_every > 1 else "all",
                     "avg_launch_ms": round(avg_launch_s * 1e3, 4),
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "stage_time_fraction_of_step": round(stage_ms / (el * 1e3), 4),
                     # since round 4 a step of this equation launches nothing but its three stage kernels (NeumannBC: no ghost fill) and the
                     # Δt candidates' 4 µs kernel on a side stream; a timed launch carries its event pair (≈3 µs), so 3 × avg_launch_ms
                     # may exceed ms_per_step by a few tenths of a percent — the step time is the untimed truth
                     "launch_timing_note": "every timed launch includes its HIP-event pair (≈3 µs): avg_launch_ms is an upper bound of the kernel's duration"},
    }
    # Counter-derived figures cannot be collected from inside this process: they come from the rocprofv3 passes of this
    # same command committed under profiles/r4/ (tools/profile_round.sh) — and only when that summary was measured on the
    # kernel sources this library was built from, on this grid, GPU count and mode.  Otherwise traffic stays null.
    #   traffic = 2·FETCH_SIZE + WRITE_SIZE (KiB -> B) per launch: FETCH_SIZE counts half of the bytes of the 8-byte-per-lane
    #   reads of this kernel, WRITE_SIZE is exact (calibrated on known-traffic kernels of the same access width).
    prof = committed_profile(args.n, world, args.mode) if world == 1 else None
    if prof:
        out["roofline"]["traffic"] = int(prof["hbm_traffic_bytes_per_launch"])
        out["roofline"]["traffic_source"] = f"profile-derived: {prof['_file']} (same kernel sources, grid, mode)"
    # The binding resource is the fp64 vector pipe, not HBM: every fp64 VALU instruction holds its SIMD for
    # 4 cycles (v_rcp/v_rsq_f64: 16; tools/ubench2.hip), and the kernel issues `valu_cycles_per_node_stage` of them per node.
    # frac_of_issue = that issue time at the clock the kernel holds ÷ the measured launch time.
    if prof and n_launch:
        wave_planes = local_cells / 64.0
        simds = 256 * 4
        cyc = prof["valu_busy_cycles_per_wave_plane"]
        clk = prof.get("in_kernel_clock_ghz") or prof["clock_ghz_grbm"]
        issue_s = wave_planes * cyc / simds / (clk * 1e9)
        out["roofline_compute"] = {"bound": "fp64_valu", "valu_per_node_stage": round(prof["valu_per_wave_plane"], 1),
                                   "valu_issue_cycles_per_node_stage": round(cyc, 1),
                                   "clock_ghz_measured": clk,
                                   "clock_source": "s_memtime/s_memrealtime stamps, diagnostic build (tools/clock_probe.py)" if prof.get("in_kernel_clock_ghz")
                                   else "GRBM_GUI_ACTIVE / 8 / duration (rocprofv3)",
                                   "clock_ghz_nominal": 2.4,
                                   "frac_of_issue": round(issue_s / avg_launch_s, 4),
                                   "frac_of_issue_at_nominal_clock": round(wave_planes * cyc / simds / 2.4e9 / avg_launch_s, 4),
                                   "source": f"{prof['_file']} (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU per dispatch) × this run's launch time"}
    # ≈290 flop per node-stage (72 FMAs among ≈220 fp64 instructions; per-block opcode counts of the wave-uniform path, tools/isa_blocks.py)
    flop_per_node_stage = 290.0
    out["fp64_vector"] = {"achieved_tflops": round(local_cells * 3 * args.steps * flop_per_node_stage / (stage_ms * 1e-3) / 1e12, 2)
                          if n_launch else 0.0, "peak_tflops": FP64_PEAK_TFLOPS,
                          "note": "algorithmic flop estimate from the ISA of the fused stage kernel (DESIGN.md §3.1)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the oracle on ONE core (the reference's hot path is single-threaded, src/timestepping.jl:101-202)
        # and OpenMP over all host cores.  Default: 2 RK3 steps at 256^3 on one core, 2 at 512^3 on all cores (≈25 s each);
        # --cpu-full: 10 steps at 256^3 on one core as SURVEY asks (≈2 min; the rate is the same: profiles/r3/cpu_full.json).
        from oracle import oracle as orc
        s1 = 10 if args.cpu_full else 2
        v1, t1 = cpu_baseline(args.cpu_sample, 1, s1)
        nthr = orc.max_threads()
        vN, tN = cpu_baseline(args.n, nthr, 2)
        out["cpu_baseline"] = {"value": round(v1, 4), "unit": "Mcells/s", "cores": 1, "kind": "port",
                               "sample": f"{s1} RK3 steps of the same equation on {args.cpu_sample}^3 (oracle, single thread as the "
                                         f"reference runs; {t1:.1f} s)",
                               "all_cores": {"value": round(vN, 4), "cores": nthr, "seconds": round(tN, 2),
                                             "sample": f"2 RK3 steps on {args.n}^3, OpenMP over the outer dimension"}}
    if ctx.transport == "local" and world > 1:
        # a rehearsal on ONE device: say so where a reader of the line looks first
        out["n_gpus"] = 1
        out["config"]["ranks"] = world
        out["config"]["transport"] = f"in-process ({world} rank threads sharing device 0; device-to-device copies stand in for xGMI)"
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

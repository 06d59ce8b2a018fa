"""Random 3-D band runs, brick stage (csrc/stage_brick.h) against the tiled band stage (LSM_BAND_BRICKS=0), bit for bit:
python tools/brick_stress.py [seed] [cases]   (LSM_BAND_MC from the environment)"""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
import lsm_amd as lsm
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(seed)
bad = 0
total = 0
menu = {
    "weno_rot": lambda: lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()),
    "weno_c": lambda: lsm.AdvectionTerm(tuple(rng.uniform(-1, 1, 3)), lsm.WENO5()),
    "up_c": lambda: lsm.AdvectionTerm(tuple(rng.uniform(-1, 1, 3)), lsm.Upwind()),
    "up_rot": lambda: lsm.AdvectionTerm(lsm.RigidRotation(), lsm.Upwind()),
    "nm": lambda: lsm.NormalMotionTerm(float(rng.uniform(-0.8, 0.8))),
    "curv": lambda: lsm.CurvatureTerm(-float(rng.uniform(0.005, 0.05))),
    "eik": lambda: lsm.EikonalReinitializationTerm(),
}
combos = [("weno_rot",), ("weno_c",), ("up_c",), ("up_rot",), ("nm",), ("curv",), ("eik",), ("weno_rot", "eik"), ("nm", "curv"),
          ("weno_c", "curv"), ("weno_rot", "nm"), ("weno_c", "nm", "curv"), ("up_c", "eik")]
for it in range(ncase):
    shape = tuple(int(rng.integers(9, 90)) for _ in range(3))
    while np.prod(shape) > 150000:
        shape = tuple(max(9, s * 3 // 4) for s in shape)
    nl = int(rng.integers(2, 6))
    ctr = rng.uniform(-0.6, 0.6, 3); r = rng.uniform(0.3, 0.9)
    dt = np.dtype(rng.choice(["float64", "float32"]))
    grid = lsm.CartesianGrid((-1.0,) * 3, (1.0,) * 3, shape)
    names = combos[int(rng.integers(0, len(combos)))]
    state = rng.bit_generator.state
    bcs = tuple(lsm.SymmetryBC() if rng.integers(0, 3) == 0 else lsm.ExtrapolationBC(int(rng.integers(0, 4))) for _ in range(3))
    integ = [lsm.RK3, lsm.RK2, lsm.ForwardEuler][int(rng.integers(0, 3))]
    phi = lsm.MeshField(lambda x: np.sqrt(sum((x[d] - ctr[d]) ** 2 for d in range(3))) - r, grid, dtype=dt)
    out = []
    try:
        for bricks in (1, 0):
            rng.bit_generator.state = state          # the same random coefficients for both runs
            _ = tuple(rng.integers(0, 3) for _ in range(3))
            terms = tuple(menu[n]() for n in names)
            eq = lsm.LevelSetEquation(terms=terms, ic=lsm.NarrowBandMeshField(phi, nlayers=nl), bc=bcs, integrator=integ(),
                                      tuning={"LSM_BAND_BRICKS": bricks})
            lsm.integrate_(eq, 4 * 0.5 * min(grid.meshsize()))
            st = eq.current_state()
            out.append((st.active_mask(), st.values()))
    except ValueError as e:
        print(it, shape, names, "raised:", str(e)[:70]); continue
    (ma, va), (mb, vb) = out
    ok = np.array_equal(ma, mb) and np.array_equal(va[ma], vb[mb])
    total += int(ma.sum())
    if not ok:
        bad += 1
        print("BAD", it, shape, nl, names, dt, integ.__name__, int(ma.sum()))
print("seed", seed, "cases", ncase, "band nodes compared", total, "BAD", bad)
sys.exit(1 if bad else 0)

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import numpy as np, lsm_amd as lsm
grid = lsm.CartesianGrid((-1.5, -1.5), (1.5, 1.5), (128, 128))
disk = lsm.MeshField(lambda x: np.hypot(x[0] + 0.75, x[1]) - 0.5, grid)
slot = lsm.MeshField(lambda x: np.maximum(np.abs(x[0] + 0.75) - 0.1, np.abs(x[1] + 0.25) - 0.5), grid)
eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()),), ic=disk.setdiff(slot),
                          bc=lsm.NeumannBC(), integrator=lsm.RK3())
vols = []
lsm.integrate_(eq, 0.5, posthook=lambda e: vols.append(lsm.volume(e)))
phi = eq.current_state()
lsm.reinitialize_(phi)
print("steps", len(vols), "volume", vols[0], vols[-1], "t", eq.current_time(), phi.values().shape)

import sys, os, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lsm_amd as lsm
n=int(sys.argv[1]) if len(sys.argv)>1 else 768
grid = lsm.CartesianGrid((-1,-1,-1),(1,1,1),(n,n,n))
f = lambda x: np.sqrt(x[0]**2+x[1]**2+x[2]**2)-0.5
ic = lsm.NarrowBandMeshField(lsm.MeshField(lsm.LazyMeshField(f, grid).local_values(None), grid), nlayers=3)
eq = lsm.LevelSetEquation(terms=(lsm.AdvectionTerm(lsm.RigidRotation(), lsm.WENO5()), lsm.CurvatureTerm(-0.01)), ic=ic, bc=lsm.NeumannBC(), integrator=lsm.RK3())
t = eq.state.tiles.cpu().numpy().astype(bool)
nbx, nby, nbm = (n+31)//32, (n+7)//8, (n+7)//8
T = t.reshape((nbm, nby, nbx))
W = np.zeros_like(T)
for dz in (-1,0,1):
    for dy in (-1,0,1):
        for dx in (-1,0,1):
            W |= np.roll(T, (dz,dy,dx), axis=(0,1,2))
print(json.dumps({"tiles": int(T.size), "active": int(T.sum()), "work": int(W.sum())}))

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import lsm_amd as lsm
import oracle as orc
from _reinit_ref import ReinitRef
from scipy.spatial import cKDTree
n, order, upsample = (33, 29), 3, 2
nd = len(n)
lc, hc = (-1.0,) * nd, (1.0,) * nd
og = orc.Grid(lc, hc, n)
X = np.meshgrid(*og.coords(), indexing="ij")
vals = np.asfortranarray(np.sqrt(sum((x - 0.04 * (d + 1)) ** 2 for d, x in enumerate(X))) - 0.55 + 0.06 * np.sin(4 * X[0]) * np.cos(3 * X[-1]))
eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.MeshField(vals, lsm.CartesianGrid(lc, hc, n)), bc=lsm.ExtrapolationBC(2))
sdf = lsm.NewtonSDF(eq.current_state(), order=order, upsample=upsample, maxiters=12)
got = np.asarray(sdf.get_sample_points())
obc = orc.make_bc(("extrapolation", 2), nd)
ref = ReinitRef(lambda J: orc.get(og, obc, vals, J), n, lc, hc, order=order, upsample=upsample, maxiters=12)
want = np.asarray(ref.pts)
h = ref.h
d, i = cKDTree(want).query(got)
for k in np.nonzero(d > 1e-11)[0]:
    print("device", got[k], "cell coords", (got[k] - ref.lc) / h, "nearest ref", want[i[k]], d[k])
d2, i2 = cKDTree(got).query(want)
for k in np.nonzero(d2 > 1e-11)[0]:
    print("ref   ", want[k], "cell coords", (want[k] - ref.lc) / h, "nearest dev", got[i2[k]], d2[k])

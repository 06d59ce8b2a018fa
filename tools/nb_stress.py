import sys, itertools
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import lsm_amd as lsm
from _nb_ref import NBRef
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(40):
    nd = int(rng.choice([2, 3, 3, 3]))
    shape = tuple(int(rng.integers(9, 45 if nd == 3 else 300)) for _ in range(nd))
    if nd == 3 and np.prod(shape) > 40000:
        shape = tuple(min(s, 34) for s in shape)
    nl = int(rng.integers(1, 6))
    ctr = rng.uniform(-0.9, 0.9, nd); r = rng.uniform(0.25, 0.8)
    grid = lsm.CartesianGrid((-1.0,) * nd, (1.0,) * nd, shape)
    ax = [np.linspace(-1, 1, n) for n in shape]
    X = np.meshgrid(*ax, indexing="ij")
    phi = np.asfortranarray(np.sqrt(sum((X[d] - ctr[d]) ** 2 for d in range(nd))) - r)
    bcs = []
    for d in range(nd):
        k = int(rng.integers(0, 3))
        bcs.append(lsm.SymmetryBC() if k == 0 else lsm.ExtrapolationBC(int(rng.integers(0, 4))))
    try:
        ref = NBRef(phi, nl)
    except Exception as e:
        print("ref failed", e); continue
    if len(ref.d) == 0:
        continue
    try:
        eq = lsm.LevelSetEquation(terms=(lsm.NormalMotionTerm(0.0),), ic=lsm.NarrowBandMeshField(lsm.MeshField(phi, grid), nlayers=nl), bc=tuple(bcs))
    except ValueError as e:
        print(it, shape, nl, "GPU raised:", str(e)[:60]); continue
    st = eq.current_state()
    ok = np.array_equal(st.active_mask(), ref.mask())
    m = ref.mask()
    ok2 = np.array_equal(st.values()[m], ref.dense()[m])
    # move and rebuild twice
    h = min(grid.meshsize())
    ok3 = True
    for shift in (0.8 * h, -1.3 * h):
        st.buf += shift
        ref.d = {I: v + shift for I, v in ref.d.items()}
        try:
            st.rebuild(from_dense=False)
            ref.update_band()
        except ValueError as e:
            print(it, "rebuild raised", str(e)[:50]); break
        m = ref.mask()
        if not (np.array_equal(st.active_mask(), m) and np.array_equal(st.values()[m], ref.dense()[m])):
            ok3 = False
    # halo values
    st.prepare(st.buf)
    dense = st.backend.download(st.buf)
    halo = st.backend.mask_to_host(st.halo)
    band = ref.mask()
    ok4 = True
    for I in np.argwhere(halo & ~band)[::7]:
        I = tuple(int(i) for i in I)
        try:
            if dense[I] != ref.extrapolate(I): ok4 = False
        except ValueError:
            pass
    print(it, shape, nl, len(ref.d), ok, ok2, ok3, ok4)
    bad += not (ok and ok2 and ok3 and ok4)
print("BAD", bad)

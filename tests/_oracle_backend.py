"""OracleBackend — TEST-ONLY stand-in for HipBackend that evaluates the backend interface with the
CPU oracle on torch CPU tensors.  It exists so the slab decomposition, ghost-plane exchange and Δt
all-reduce of lsm_amd.api can be exercised with world_size-2 gloo process groups on a machine
without a GPU.  The product never constructs it (lsm_amd has no reference to oracle/)."""
import ctypes as C

import numpy as np
import torch

from oracle import oracle as orc


class _Lay:
    pass


class OracleBackend:
    name = "oracle-test"

    def __init__(self, grid_c, bc_c, slab):
        self.ndim = int(grid_c.ndim)
        self.grid = orc.Grid([grid_c.lc[d] for d in range(self.ndim)], [grid_c.hc[d] for d in range(self.ndim)],
                             [grid_c.n[d] for d in range(self.ndim)])
        self.bc = orc.BcArray.from_buffer_copy(bytes(bc_c))
        self.slab = slab
        self.lay = orc.layout(self.grid, slab)
        self.device = torch.device("cpu")

    # memory
    def alloc(self):
        return torch.zeros(int(self.lay.total), dtype=torch.float64)

    def clone(self, t):
        return t.clone()

    def copy_(self, dst, src):
        dst.copy_(src)

    def flat(self, t):
        return t

    def _pad(self, t):
        shape = orc.padded_shape(self.lay, self.ndim)
        return t.numpy().reshape(shape, order="F")

    def local_shape(self):
        return tuple(int(self.lay.n[d]) for d in range(self.ndim))

    def upload(self, t, dense):
        g = orc.GHOST
        sl = tuple(slice(g, g + n) for n in self.local_shape())
        self._pad(t)[sl] = np.asarray(dense, dtype=np.float64)

    def download(self, t):
        return orc.from_padded(self.lay, self.ndim, self._pad(t))

    def table(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).copy())

    # kernels
    def _terms(self, terms_c, n):
        return orc.LsmTerm.__mul__(n).from_buffer_copy(bytes(terms_c)[:C.sizeof(orc.LsmTerm) * n]) if False else \
            (orc.LsmTerm * n).from_buffer_copy(bytes(terms_c)[:C.sizeof(orc.LsmTerm) * n])

    def _dp(self, t):
        return C.cast(C.c_void_p(t.data_ptr()), C.POINTER(C.c_double)) if t is not None else None

    def fill_ghosts(self, t, mask=7):
        orc.fill_ghosts_padded(self.grid, self.bc, self.lay, self._pad(t), slab=self.slab, dim_mask=mask)

    def stage(self, terms_c, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t):
        slab = C.byref(orc.LsmSlab(*self.slab)) if self.slab is not None else None
        orc.lib().orc_stage_padded(C.byref(self.grid.c), self.bc, slab, C.byref(self.lay), self._terms(terms_c, nterms), nterms,
                                   self._dp(psi), self._dp(phin), self._dp(out), self._dp(out2), base_mode, cdt, cdt2, t)

    def stage_planes(self, terms_c, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t, m0, m1):
        """Plane-restricted stage: evaluate the whole stage into scratch, keep planes [m0, m1)."""
        tmp = out.clone()
        tmp2 = out2.clone() if out2 is not None else None
        self.stage(terms_c, nterms, psi, phin if phin is not out else tmp, tmp, tmp2, base_mode, cdt, cdt2, t)
        g = orc.GHOST
        sl = int(self.lay.stride[self.ndim - 1])
        out[(m0 + g) * sl:(m1 + g) * sl] = tmp[(m0 + g) * sl:(m1 + g) * sl]
        if out2 is not None:
            out2[(m0 + g) * sl:(m1 + g) * sl] = tmp2[(m0 + g) * sl:(m1 + g) * sl]

    def fill_ghosts_planes(self, t, m0, m1, fill_last=False):
        # lower-dimension ghosts of every plane (idempotent; planes updated later are refilled later)
        self.fill_ghosts(t, mask=((1 << (self.ndim - 1)) - 1) | ((1 << (self.ndim - 1)) if fill_last else 0))

    def compute_cfl_local(self, terms_c, nterms, phi, t):
        slab = C.byref(orc.LsmSlab(*self.slab)) if self.slab is not None else None
        return orc.lib().orc_cfl_padded(C.byref(self.grid.c), self.bc, slab, C.byref(self.lay), self._terms(terms_c, nterms),
                                        nterms, self._dp(phi), t)

    def eikonal_sign(self, phi0, s0):
        dx = min(self.grid.meshsize())
        v = phi0.numpy()
        s0.numpy()[...] = v / np.sqrt(v * v + dx * dx)

    def extrema(self, t):
        d = self.download(t)
        return float(d.min()), float(d.max())

    def sync(self):
        pass

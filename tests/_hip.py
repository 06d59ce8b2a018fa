"""Test helper: builds the SAME problem for the CPU oracle (host pointers) and for libhiplsm
(device pointers) from one spec, and moves padded arrays between the two."""
import ctypes as C

import numpy as np

import lsm_amd
from lsm_amd import _lib as L
from lsm_amd.backend import HipBackend
from oracle import oracle as orc


class Case:
    def __init__(self, shape, bcspec, lc=None, hc=None, mode="strict", dtype=np.float64):
        import torch
        self.torch = torch
        nd = len(shape)
        self.nd = nd
        lc = lc or (-1.0,) * nd
        hc = hc or (1.0,) * nd
        self.grid = orc.Grid(lc, hc, shape)
        self.bc = orc.make_bc(bcspec, nd)
        self.olay = orc.layout(self.grid)
        gc = L.LsmGrid.from_buffer_copy(bytes(self.grid.c))
        bcc = L.BcArray.from_buffer_copy(bytes(self.bc))
        self.dtype = np.dtype(dtype)     # storage of the level-set fields; side arrays (coefficients, S0) stay float64
        self.be = HipBackend(gc, bcc, mode=mode, dtype=dtype)
        self.lay = self.be.lay
        self.keep = []

    # ---- padded arrays host <-> device (honours the library's strides)
    def _view(self, flat):
        nd, lay = self.nd, self.lay
        shape = tuple(int(lay.n[d] + 2 * lay.g[d]) for d in range(nd))
        strides = tuple(int(lay.stride[d]) * 8 for d in range(nd))
        off = int(lay.origin) - sum(int(lay.g[d]) * int(lay.stride[d]) for d in range(nd))
        return np.lib.stride_tricks.as_strided(flat[off:], shape=shape, strides=strides)

    def to_dev(self, padded, side=False):
        flat = np.zeros(int(self.lay.total), dtype=np.float64)
        self._view(flat)[...] = padded
        if not side and self.dtype == np.float32:
            flat = flat.astype(np.float32)
        return self.torch.from_numpy(flat).to(self.be.device)

    def to_host(self, t):
        flat = t.cpu().numpy().astype(np.float64)
        return np.asfortranarray(self._view(flat).copy())

    def pad(self, dense, fill=True):
        p = orc.to_padded(self.olay, self.nd, np.asfortranarray(dense))
        if fill:
            orc.fill_ghosts_padded(self.grid, self.bc, self.olay, p)
        return p

    def zero_pad(self, dense):
        """Padded copy with zero ghosts (coefficient fields are only ever read in-grid)."""
        p = orc.to_padded(self.olay, self.nd, np.zeros_like(np.asfortranarray(dense)))
        p[np.isnan(p)] = 0.0
        sl = tuple(slice(int(self.olay.g[d]), int(self.olay.g[d] + self.olay.n[d])) for d in range(self.nd))
        p[sl] = dense
        return p

    def interior(self, p):
        return orc.from_padded(self.olay, self.nd, p)

    # ---- terms for both sides
    def _coeff(self, spec, ncomp):
        kind = spec[0]
        hc = L.LsmCoeff()
        if kind == "const":
            oc = orc.const(*spec[1])
            hc.kind = L.COEFF_CONST
            for i, v in enumerate(spec[1]):
                hc.value[i] = v
        elif kind == "rot":
            oc = orc.rotation(*spec[1:])
            hc.kind = L.COEFF_ROTATION
            for i, v in enumerate(spec[1:]):
                hc.value[i] = v
        elif kind == "sep":
            tables, time = spec[1], spec[2]
            tk, tp = (orc.TIME_COS, time[1]) if time else (orc.TIME_ONE, 1.0)
            oc = orc.separable(tables, tk, tp)
            hc.kind, hc.time_kind, hc.time_param = L.COEFF_SEPARABLE, tk, tp
            for i, comp in enumerate(tables):
                t = self.be.table(np.concatenate([np.asarray(a, dtype=np.float64) for a in comp]))
                self.keep.append(t)
                hc.sep[i] = t.data_ptr()
        else:  # field: dense arrays
            padded = [self.zero_pad(a) for a in spec[1]]
            oc = orc.field(*padded)
            hc.kind = L.COEFF_FIELD
            for i, p in enumerate(padded):
                t = self.to_dev(p, side=True)
                self.keep.append(t)
                hc.field[i] = t.data_ptr()
        self.keep.append(oc)
        return oc, hc

    def terms(self, specs):
        """specs: ('adv', coeff, 'weno5'|'upwind') | ('nm', coeff) | ('curv', coeff) | ('eik', None|dense ϕ₀)."""
        ot, arr = [], (L.LsmTerm * len(specs))()
        for i, s in enumerate(specs):
            if s[0] == "adv":
                oc, hc = self._coeff(s[1], self.nd)
                sch = orc.SCHEME_WENO5 if s[2] == "weno5" else orc.SCHEME_UPWIND
                ot.append(orc.advection(oc, sch))
                arr[i].kind, arr[i].scheme = L.TERM_ADVECTION, sch
                C.memmove(C.byref(arr[i].coeff), C.byref(hc), C.sizeof(L.LsmCoeff))
            elif s[0] in ("nm", "curv"):
                oc, hc = self._coeff(s[1], 1)
                ot.append(orc.normal_motion(oc) if s[0] == "nm" else orc.curvature(oc))
                arr[i].kind = L.TERM_NORMAL_MOTION if s[0] == "nm" else L.TERM_CURVATURE
                C.memmove(C.byref(arr[i].coeff), C.byref(hc), C.sizeof(L.LsmCoeff))
            else:
                arr[i].kind = L.TERM_EIKONAL
                if s[1] is None:
                    ot.append(orc.eikonal())
                else:
                    s0 = orc.eikonal_sign(self.grid, np.asfortranarray(s[1]))
                    p = self.pad(s0, fill=False)
                    ot.append(orc.eikonal(p))
                    t = self.to_dev(np.nan_to_num(p, nan=0.0), side=True)
                    self.keep.append(t)
                    arr[i].s0 = t.data_ptr()
        self.keep.append(ot)
        return ot, arr

    def dense_terms(self, specs):
        """Oracle terms for the dense (reference-layout) API."""
        out = []
        for s in specs:
            if s[0] == "adv":
                out.append(orc.advection(self._dense_coeff(s[1]), orc.SCHEME_WENO5 if s[2] == "weno5" else orc.SCHEME_UPWIND))
            elif s[0] == "nm":
                out.append(orc.normal_motion(self._dense_coeff(s[1])))
            elif s[0] == "curv":
                out.append(orc.curvature(self._dense_coeff(s[1])))
            else:
                out.append(orc.eikonal(None if s[1] is None else orc.eikonal_sign(self.grid, np.asfortranarray(s[1]))))
        return out

    def _dense_coeff(self, spec):
        if spec[0] == "const":
            return orc.const(*spec[1])
        if spec[0] == "rot":
            return orc.rotation(*spec[1:])
        if spec[0] == "sep":
            tk, tp = (orc.TIME_COS, spec[2][1]) if spec[2] else (orc.TIME_ONE, 1.0)
            return orc.separable(spec[1], tk, tp)
        return orc.field(*[np.asfortranarray(a) for a in spec[1]])

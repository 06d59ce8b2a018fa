"""HipBackend — device memory (torch tensors as plain HBM buffers), streams and the calls into
libhiplsm.so.  torch is plumbing here: allocation, the current HIP stream, and — in slab mode — the bootstrap
of the library's own RCCL communicator (the ghost-plane exchange, the Δt all-reduce and the overlap planes of a
slab-decomposed narrow band all run inside the library).

The interface (layout / alloc / upload / download / fill_ghosts / stage / compute_cfl_local /
advance_single / eikonal_sign / extrema / table) is the seam the host logic in api.py talks to.
The only implementation in this package is the HIP one; it raises if the library or the GPU is
missing — there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib as L


class HipBackend:
    name = "hip"

    def __init__(self, grid_c, bc_c, slab=None, mode="fast", device=0, dtype=np.float64, tuning=None):
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("levelsetmethods.jl_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self.torch = torch
        self.lib = L.lib()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.ndim = int(grid_c.ndim)
        self._grid = grid_c
        self._bc = bc_c
        self.slab = slab
        self.dtype = np.dtype(dtype)          # storage of the level-set fields (float64 | float32); side arrays are float64
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError(f"unsupported field dtype {self.dtype}: float64 or float32")
        h = C.c_void_p()
        slab_c = L.LsmSlab(slab[0], slab[1]) if slab is not None else None
        code = self.lib.lsm_create(C.byref(grid_c), bc_c, C.byref(slab_c) if slab_c is not None else None,
                                   L.DTYPE_F32 if self.dtype == np.float32 else L.DTYPE_F64,
                                   L.MODE_STRICT if mode == "strict" else L.MODE_FAST, device, C.byref(h))
        if code != L.OK:
            raise L.LsmError(f"lsm_create failed ({code}): {self.lib.lsm_last_error(None).decode()}")
        self.h = h
        # kernels, torch copies and RCCL all order on torch's current stream
        L.check(self.h, self.lib.lsm_set_stream(self.h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                "lsm_set_stream")
        self.lay = L.LsmLayout()
        L.check(self.h, self.lib.lsm_layout(self.h, C.byref(self.lay)), "lsm_layout")
        for name, value in (tuning or {}).items():     # include/lsm.h, "tuning switches": before anything is built on the handle
            self.set_tuning(name, value)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lsm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- memory
    def alloc(self):
        """A level-set field (ϕ, stage buffer, extension target) in the handle's storage type."""
        tdt = self.torch.float32 if self.dtype == np.float32 else self.torch.float64
        return self.torch.zeros(int(self.lay.total), dtype=tdt, device=self.device)

    def alloc_side(self):
        """A side array (coefficient field, frozen sign / mask): always float64."""
        return self.torch.zeros(int(self.lay.total), dtype=self.torch.float64, device=self.device)

    def clone(self, t):
        return t.clone()

    def copy_(self, dst, src):
        dst.copy_(src)

    def ptr(self, t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def local_shape(self):
        return tuple(int(self.lay.n[d]) for d in range(self.ndim))

    def upload(self, t, dense):
        a = np.asfortranarray(dense, dtype=self.dtype)
        assert a.shape == self.local_shape(), (a.shape, self.local_shape())
        L.check(self.h, self.lib.lsm_upload(self.h, self.ptr(t), a.ctypes.data_as(C.c_void_p)), "lsm_upload")

    def download(self, t):
        out = np.empty(self.local_shape(), dtype=self.dtype, order="F")
        L.check(self.h, self.lib.lsm_download(self.h, self.ptr(t), out.ctypes.data_as(C.c_void_p)), "lsm_download")
        return out

    def upload_side(self, t, dense):
        a = np.asfortranarray(dense, dtype=np.float64)
        assert a.shape == self.local_shape(), (a.shape, self.local_shape())
        L.check(self.h, self.lib.lsm_upload_f64(self.h, self.ptr(t), a.ctypes.data_as(C.c_void_p)), "lsm_upload_f64")

    def download_side(self, t):
        out = np.empty(self.local_shape(), dtype=np.float64, order="F")
        L.check(self.h, self.lib.lsm_download_f64(self.h, self.ptr(t), out.ctypes.data_as(C.c_void_p)), "lsm_download_f64")
        return out

    def table(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.device)

    def flat(self, t):
        return t

    # ---- kernels
    def fill_ghosts(self, t, mask=7):
        L.check(self.h, self.lib.lsm_fill_ghosts(self.h, self.ptr(t), mask, None), "lsm_fill_ghosts")

    def stage(self, terms_c, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t):
        L.check(self.h, self.lib.lsm_stage(self.h, terms_c, nterms, self.ptr(psi), self.ptr(phin), self.ptr(out),
                                           self.ptr(out2), base_mode, cdt, cdt2, t, None), "lsm_stage")

    def stage_planes(self, terms_c, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t, m0, m1):
        L.check(self.h, self.lib.lsm_stage_planes(self.h, terms_c, nterms, self.ptr(psi), self.ptr(phin), self.ptr(out),
                                                  self.ptr(out2), base_mode, cdt, cdt2, t, m0, m1, None), "lsm_stage_planes")

    def fill_ghosts_planes(self, t, m0, m1, fill_last=False):
        L.check(self.h, self.lib.lsm_fill_ghosts_planes(self.h, self.ptr(t), m0, m1, 1 if fill_last else 0, None),
                "lsm_fill_ghosts_planes")

    def compute_cfl_local(self, terms_c, nterms, phi, t):
        dt = C.c_double(0.0)
        L.check(self.h, self.lib.lsm_compute_cfl(self.h, terms_c, nterms, self.ptr(phi), t, C.byref(dt)), "lsm_compute_cfl")
        return dt.value

    def cfl_cache(self, enable):
        L.check(self.h, self.lib.lsm_cfl_cache(self.h, 1 if enable else 0), "lsm_cfl_cache")

    def advance_single(self, which, terms_c, nterms, phi, b1, b2, tc, dt, hook):
        cb = hook if hook is not None else C.cast(None, L.StageHook)
        if which == "fe":
            code = self.lib.lsm_advance_fe(self.h, terms_c, nterms, self.ptr(phi), self.ptr(b1), tc, dt, cb, None)
        elif which == "rk2":
            code = self.lib.lsm_advance_rk2(self.h, terms_c, nterms, self.ptr(phi), self.ptr(b1), self.ptr(b2), tc, dt, cb, None)
        else:
            code = self.lib.lsm_advance_rk3(self.h, terms_c, nterms, self.ptr(phi), self.ptr(b1), self.ptr(b2), tc, dt, cb, None)
        L.check(self.h, code, f"lsm_advance_{which}")

    def check_range(self, t):
        """(ok, max|ϕ|): is the field inside the domain of the handle's arithmetic mode (include/lsm.h, LSM_FAST_MAX_ABS)?"""
        ok, m = C.c_int(), C.c_double()
        L.check(self.h, self.lib.lsm_check_range(self.h, self.ptr(t), C.byref(ok), C.byref(m)), "lsm_check_range")
        return bool(ok.value), m.value

    # ---- multi-GPU: slab communicator inside the library (include/lsm.h, "multi-GPU")
    def comm_unique_id(self):
        buf = C.create_string_buffer(L.COMM_ID_BYTES)
        L.check(None, self.lib.lsm_comm_unique_id(buf), "lsm_comm_unique_id")
        return buf.raw

    def comm_attach_rccl(self, unique_id, rank, world):
        """Collective over the ranks: ncclCommInitRank on this handle's device."""
        L.check(self.h, self.lib.lsm_comm_attach_rccl(self.h, C.c_char_p(unique_id), rank, world), "lsm_comm_attach_rccl")

    @staticmethod
    def comm_attach_local(backends):
        """All ranks in this process: backends[r] = rank r (LSM_COMM_LOCAL)."""
        arr = (C.c_void_p * len(backends))(*[b.h for b in backends])
        code = backends[0].lib.lsm_comm_attach_local(arr, len(backends))
        if code != L.OK:
            msgs = [b.lib.lsm_last_error(b.h) for b in backends]
            raise L.LsmError(f"lsm_comm_attach_local failed ({code}): " + "; ".join(m.decode() for m in msgs if m))

    def comm_detach(self):
        L.check(self.h, self.lib.lsm_comm_detach(self.h), "lsm_comm_detach")

    def comm_info(self):
        r, w, t = C.c_int(), C.c_int(), C.c_int()
        L.check(self.h, self.lib.lsm_comm_info(self.h, C.byref(r), C.byref(w), C.byref(t)), "lsm_comm_info")
        return r.value, w.value, t.value

    def comm_set_overlap(self, enable):
        L.check(self.h, self.lib.lsm_comm_set_overlap(self.h, 1 if enable else 0), "lsm_comm_set_overlap")

    def halo_start(self, t):
        L.check(self.h, self.lib.lsm_halo_start(self.h, self.ptr(t)), "lsm_halo_start")

    def halo_wait(self):
        L.check(self.h, self.lib.lsm_halo_wait(self.h), "lsm_halo_wait")

    def halo_exchange(self, t):
        L.check(self.h, self.lib.lsm_halo_exchange(self.h, self.ptr(t)), "lsm_halo_exchange")

    def allreduce_dt(self, dt):
        v = C.c_double(dt)
        L.check(self.h, self.lib.lsm_allreduce_dt(self.h, C.byref(v)), "lsm_allreduce_dt")
        return v.value

    def comm_abort(self):
        """Fail this rank's communicator (LOCAL: the whole group): peers blocked in an exchange get LSM_ERR_COMM."""
        if getattr(self, "h", None):
            self.lib.lsm_comm_abort(self.h)

    def band_overlap_config(self, overlap):
        L.check(self.h, self.lib.lsm_band_overlap_config(self.h, int(overlap)), "lsm_band_overlap_config")

    def band_overlap_mask(self, mask):
        L.check(self.h, self.lib.lsm_band_overlap_mask(self.h, self.ptr(mask)), "lsm_band_overlap_mask")

    def band_overlap_values(self, t):
        L.check(self.h, self.lib.lsm_band_overlap_values(self.h, self.ptr(t)), "lsm_band_overlap_values")

    def eikonal_sign(self, phi0, s0):
        L.check(self.h, self.lib.lsm_eikonal_sign(self.h, self.ptr(phi0), self.ptr(s0), None), "lsm_eikonal_sign")

    def extrema(self, t):
        lo, hi = C.c_double(), C.c_double()
        L.check(self.h, self.lib.lsm_extrema(self.h, self.ptr(t), C.byref(lo), C.byref(hi)), "lsm_extrema")
        return lo.value, hi.value

    def geometry(self, what, phi, outs, scale=1.0, band_width=-1.0, fill=0.0, frozen_out=None, mask=None):
        """curvature / gradient / normal of phi at every node (mask: at the band nodes of a prepared band field) into
        fp64 side arrays (lsm_geometry / lsm_band_geometry)."""
        o = [self.ptr(t) for t in outs] + [None] * (3 - len(outs))
        if mask is None:
            L.check(self.h, self.lib.lsm_geometry(self.h, int(what), self.ptr(phi), float(scale), float(band_width), float(fill), o[0], o[1], o[2],
                                                  self.ptr(frozen_out), None), "lsm_geometry")
        else:
            L.check(self.h, self.lib.lsm_band_geometry(self.h, int(what), self.ptr(phi), self.ptr(mask), float(scale), float(band_width), float(fill),
                                                       o[0], o[1], o[2], self.ptr(frozen_out), None), "lsm_band_geometry")

    def interpolate(self, phi, order, pts, want_grad, want_hess):
        """InterpolatedField evaluation at host points (npts x ndim): returns (values, gradients or None, hessians or None)."""
        t = self.torch
        p = t.from_numpy(np.ascontiguousarray(pts, dtype=np.float64)).to(self.device)
        n, N = int(p.shape[0]), self.ndim
        val = t.empty(n, dtype=t.float64, device=self.device)
        grad = t.empty((n, N), dtype=t.float64, device=self.device) if want_grad else None
        hess = t.empty((n, N, N), dtype=t.float64, device=self.device) if want_hess else None
        L.check(self.h, self.lib.lsm_interpolate(self.h, self.ptr(phi), int(order), n, self.ptr(p), self.ptr(val), self.ptr(grad), self.ptr(hess), None),
                "lsm_interpolate")
        self.sync()
        return val.cpu().numpy(), (grad.cpu().numpy() if want_grad else None), (hess.cpu().numpy() if want_hess else None)

    # ---- NewtonSDF objects (lsm_sdf_*)
    def sdf_create(self, phi, mask, order, upsample, maxiters, xtol, ftol):
        out, ns = C.c_void_p(), C.c_int64()
        L.check(self.h, self.lib.lsm_sdf_create(self.h, self.ptr(phi), self.ptr(mask), int(order), int(upsample), int(maxiters), float(xtol), float(ftol),
                                                C.byref(out), C.byref(ns)), "lsm_sdf_create")
        return out, int(ns.value)

    def sdf_eval(self, sdf, pts, want_cp):
        t = self.torch
        p = t.from_numpy(np.ascontiguousarray(pts, dtype=np.float64)).to(self.device)
        n, N = int(p.shape[0]), self.ndim
        dist = t.empty(n, dtype=t.float64, device=self.device)
        cp = t.empty((n, N), dtype=t.float64, device=self.device) if want_cp else None
        nf = C.c_int64()
        L.check(self.h, self.lib.lsm_sdf_eval(sdf, n, self.ptr(p), self.ptr(dist), self.ptr(cp), C.byref(nf)), "lsm_sdf_eval")
        return dist.cpu().numpy(), (cp.cpu().numpy() if want_cp else None), int(nf.value)

    def sdf_samples(self, sdf, nsamples):
        out = self.torch.empty((max(nsamples, 1), self.ndim), dtype=self.torch.float64, device=self.device)
        L.check(self.h, self.lib.lsm_sdf_samples(sdf, self.ptr(out)), "lsm_sdf_samples")
        return out[:nsamples].cpu().numpy()

    def sdf_destroy(self, sdf):
        self.lib.lsm_sdf_destroy(sdf)

    def extend_along_normals(self, F, phi, frozen, nb_iters, cfl, interface_band, min_norm):
        work = [self.alloc()] + [self.alloc_side() for _ in range(self.ndim)]   # F staging + the normal components
        w = [self.ptr(x) for x in work] + [None] * (4 - len(work))
        L.check(self.h, self.lib.lsm_extend_along_normals(self.h, self.ptr(F), self.ptr(phi), self.ptr(frozen), w[0], w[1], w[2], w[3],
                                                          nb_iters, cfl, interface_band, min_norm), "lsm_extend_along_normals")
        self.sync()   # the work buffers are released on return

    # ---- narrow band (byte masks over the padded index space)
    def alloc_mask(self):
        return self.torch.zeros(int(self.lay.total), dtype=self.torch.uint8, device=self.device)

    def band_tile_count(self, mc):
        n = C.c_int64()
        L.check(self.h, self.lib.lsm_band_tile_count(self.h, int(mc), C.byref(n)), "lsm_band_tile_count")
        return int(n.value)

    def alloc_tiles(self, mc):
        return self.torch.zeros(self.band_tile_count(mc), dtype=self.torch.uint8, device=self.device)

    def alloc_halo_list(self, cap):
        """(entries, counter): 16-byte (node, nearest band node) records and the device uint32 count."""
        return (self.torch.empty(2 * int(cap), dtype=self.torch.int64, device=self.device),
                self.torch.zeros(1, dtype=self.torch.int32, device=self.device))

    def band_update(self, vals, mask, from_dense, nlayers, scratch_a, scratch_b, halo, tiles, mc, hlist, hcount):
        L.check(self.h, self.lib.lsm_band_update(self.h, self.ptr(vals), self.ptr(mask), 1 if from_dense else 0, int(nlayers),
                                                 self.ptr(scratch_a), self.ptr(scratch_b), self.ptr(halo), self.ptr(tiles), int(mc),
                                                 self.ptr(hlist), hlist.numel() // 2, self.ptr(hcount)), "lsm_band_update")

    def band_halo(self, vals, mask, halo, tiles, mc, hlist, hcount):
        L.check(self.h, self.lib.lsm_band_halo(self.h, self.ptr(vals), self.ptr(mask), self.ptr(halo), self.ptr(tiles), int(mc),
                                               self.ptr(hlist), hlist.numel() // 2, self.ptr(hcount)), "lsm_band_halo")

    def set_tuning(self, name, value):
        """lsm_set_tuning: one of the switches of include/lsm.h ("LSM_BAND_BRICKS", ...) on this handle."""
        L.check(self.h, self.lib.lsm_set_tuning(self.h, name.encode(), int(value)), "lsm_set_tuning")

    def get_tuning(self, name):
        v = C.c_int()
        L.check(self.h, self.lib.lsm_get_tuning(self.h, name.encode(), C.byref(v)), "lsm_get_tuning")
        return int(v.value)

    def band_retile(self, mask, tiles, mc):
        L.check(self.h, self.lib.lsm_band_retile(self.h, self.ptr(mask), self.ptr(tiles), int(mc)), "lsm_band_retile")

    def band_fill_list(self, vals, mask, hlist, hcount):
        L.check(self.h, self.lib.lsm_band_fill_list(self.h, self.ptr(vals), self.ptr(mask), self.ptr(hlist), hlist.numel() // 2,
                                                    self.ptr(hcount)), "lsm_band_fill_list")

    def band_prepare(self, vals, mask, hlist, hcount, tiles, mc):
        L.check(self.h, self.lib.lsm_band_prepare(self.h, self.ptr(vals), self.ptr(mask), self.ptr(hlist), hlist.numel() // 2,
                                                  self.ptr(hcount), self.ptr(tiles), int(mc)), "lsm_band_prepare")

    def band_status(self, hcount):
        n, m = C.c_int64(), C.c_int()
        L.check(self.h, self.lib.lsm_band_status(self.h, self.ptr(hcount), C.byref(n), C.byref(m)), "lsm_band_status")
        return int(n.value), bool(m.value)

    def band_fill(self, vals, mask, targets, tiles, mc):
        L.check(self.h, self.lib.lsm_band_fill(self.h, self.ptr(vals), self.ptr(mask), self.ptr(targets), self.ptr(tiles), int(mc)),
                "lsm_band_fill")

    def band_count(self, mask):
        n = C.c_int64()
        L.check(self.h, self.lib.lsm_band_count(self.h, self.ptr(mask), C.byref(n)), "lsm_band_count")
        return int(n.value)

    def band_missed(self):
        m = C.c_int()
        L.check(self.h, self.lib.lsm_band_missed(self.h, C.byref(m)), "lsm_band_missed")
        return bool(m.value)

    def stage_band(self, terms_c, nterms, psi, phin, out, out2, base_mode, cdt, cdt2, t, mask, tiles, mc):
        L.check(self.h, self.lib.lsm_stage_band(self.h, terms_c, nterms, self.ptr(psi), self.ptr(phin), self.ptr(out), self.ptr(out2),
                                                base_mode, cdt, cdt2, t, self.ptr(mask), self.ptr(tiles), int(mc), None), "lsm_stage_band")

    def band_c(self, mask, tiles, mc, hlist, hcount):
        """LsmBand: what lsm_band_update maintains for a field, as lsm_advance_band_* takes it."""
        return L.LsmBand(self.ptr(mask), self.ptr(tiles), int(mc), 0, self.ptr(hlist), hlist.numel() // 2, self.ptr(hcount))

    def advance_band(self, which, terms_c, nterms, band_c, phi, b1, b2, tc, dt, hook):
        cb = hook if hook is not None else C.cast(None, L.StageHook)
        if which == "fe":
            code = self.lib.lsm_advance_band_fe(self.h, terms_c, nterms, C.byref(band_c), self.ptr(phi), self.ptr(b1), tc, dt, cb, None)
        elif which == "rk2":
            code = self.lib.lsm_advance_band_rk2(self.h, terms_c, nterms, C.byref(band_c), self.ptr(phi), self.ptr(b1), self.ptr(b2), tc, dt, cb, None)
        else:
            code = self.lib.lsm_advance_band_rk3(self.h, terms_c, nterms, C.byref(band_c), self.ptr(phi), self.ptr(b1), self.ptr(b2), tc, dt, cb, None)
        L.check(self.h, code, f"lsm_advance_band_{which}")

    def compute_cfl_band(self, terms_c, nterms, phi, mask, t, tiles=None, mc=0):
        dt = C.c_double(0.0)
        L.check(self.h, self.lib.lsm_compute_cfl_band(self.h, terms_c, nterms, self.ptr(phi), self.ptr(mask), self.ptr(tiles), int(mc),
                                                      t, C.byref(dt)), "lsm_compute_cfl_band")
        return dt.value

    def mask_to_host(self, mask):
        """Boolean (local) interior of a mask, Fortran order."""
        flat = mask.cpu().numpy()
        n = tuple(int(self.lay.n[d]) for d in range(self.ndim))
        strides = tuple(int(self.lay.stride[d]) * flat.itemsize for d in range(self.ndim))   # the library's layout: pitch and origin are its own
        return np.lib.stride_tricks.as_strided(flat[int(self.lay.origin):], shape=n, strides=strides).astype(bool)

    def volume_local(self, t):
        out = C.c_double()
        L.check(self.h, self.lib.lsm_volume(self.h, self.ptr(t), C.byref(out)), "lsm_volume")
        return out.value

    def perimeter_local(self, t):
        out = C.c_double()
        L.check(self.h, self.lib.lsm_perimeter(self.h, self.ptr(t), C.byref(out)), "lsm_perimeter")
        return out.value

    def band_volume(self, t, mask):
        out = C.c_double()
        L.check(self.h, self.lib.lsm_band_volume(self.h, self.ptr(t), self.ptr(mask), C.byref(out)), "lsm_band_volume")
        return out.value

    def band_perimeter(self, t, mask):
        out = C.c_double()
        L.check(self.h, self.lib.lsm_band_perimeter(self.h, self.ptr(t), self.ptr(mask), C.byref(out)), "lsm_band_perimeter")
        return out.value

    def reinitialize(self, phi, mask, order, upsample, maxiters, xtol, ftol):
        work = self.alloc()
        nc, nfail, nfar = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self.h, self.lib.lsm_reinitialize(self.h, self.ptr(phi), self.ptr(mask), self.ptr(work), int(order), int(upsample),
                                                  int(maxiters), float(xtol), float(ftol), C.byref(nc), C.byref(nfail), C.byref(nfar)),
                "lsm_reinitialize")
        return int(nc.value), int(nfail.value), int(nfar.value)

    def sync(self):
        L.check(self.h, self.lib.lsm_sync(self.h), "lsm_sync")

    def profile_enable(self, on=True):
        """on = True / 1: time every stage launch; N > 1: every N-th (lsm_profile_read scales); False / 0: off."""
        L.check(self.h, self.lib.lsm_profile_enable(self.h, int(on)), "lsm_profile_enable")

    def profile_read(self):
        n, ms = C.c_int64(), C.c_double()
        L.check(self.h, self.lib.lsm_profile_read(self.h, C.byref(n), C.byref(ms)), "lsm_profile_read")
        return n.value, ms.value

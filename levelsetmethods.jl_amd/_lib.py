"""ctypes binding of libhiplsm.so — the C ABI declared in include/lsm.h.

This is the Python twin of the `ccall` layer a Julia maintainer would write (see INTEGRATION.md
and julia/ROCMeshField.jl).  There is NO CPU fallback here: if the shared library is missing or
no MI355X is visible, loading/creating fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSM_AMD_LIB") or os.path.join(_HERE, "libhiplsm.so")   # env override: kernel-variant experiments
CSRC = os.path.join(_HERE, "csrc")

GHOST = 3
MAX_TERMS = 8
OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_COMM = 0, -1, -2, -3, -4
BC_PERIODIC, BC_EXTRAPOLATION, BC_SYMMETRY, BC_NONE = 0, 1, 2, 3
TERM_ADVECTION, TERM_NORMAL_MOTION, TERM_CURVATURE, TERM_EIKONAL = 0, 1, 2, 3
SCHEME_UPWIND, SCHEME_WENO5 = 0, 1
COEFF_CONST, COEFF_ROTATION, COEFF_SEPARABLE, COEFF_FIELD = 0, 1, 2, 3
TIME_ONE, TIME_COS = 0, 1
BASE_PSI, BASE_RK3_S2, BASE_RK3_S3, BASE_OTHER = 0, 1, 2, 3
MODE_FAST, MODE_STRICT = 0, 1
DTYPE_F64 = 0
DTYPE_F32 = 1
GEOM_CURVATURE, GEOM_GRADIENT, GEOM_NORMAL = 0, 1, 2
FAST_MAX_ABS = 2.5e34
COMM_ID_BYTES = 128
COMM_NONE, COMM_RCCL, COMM_LOCAL = 0, 1, 2


class LsmGrid(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("_pad", C.c_int32), ("n", C.c_int64 * 3), ("lc", C.c_double * 3),
                ("hc", C.c_double * 3)]


class LsmBc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("degree", C.c_int32)]


class LsmSlab(C.Structure):
    _fields_ = [("lo", C.c_int64), ("n", C.c_int64)]


class LsmLayout(C.Structure):
    _fields_ = [("n", C.c_int64 * 3), ("g", C.c_int64 * 3), ("stride", C.c_int64 * 3), ("origin", C.c_int64),
                ("total", C.c_int64)]


class LsmCoeff(C.Structure):
    _fields_ = [("kind", C.c_int32), ("time_kind", C.c_int32), ("time_param", C.c_double), ("value", C.c_double * 4),
                ("field", C.c_void_p * 3), ("sep", C.c_void_p * 3)]


class LsmTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("scheme", C.c_int32), ("coeff", LsmCoeff), ("s0", C.c_void_p)]


class LsmBand(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("tiles", C.c_void_p), ("mc", C.c_int32), ("_pad", C.c_int32), ("halo_list", C.c_void_p),
                ("halo_cap", C.c_int64), ("halo_count", C.c_void_p)]


BcArray = (LsmBc * 2) * 3
StageHook = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double)

# every symbol include/lsm.h declares: (name, restype, argtypes)
_H = C.c_void_p
_SIGS = [
    ("lsm_create", C.c_int, [C.POINTER(LsmGrid), BcArray, C.POINTER(LsmSlab), C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    ("lsm_destroy", None, [_H]),
    ("lsm_last_error", C.c_char_p, [_H]),
    ("lsm_version", C.c_char_p, []),
    ("lsm_sync", C.c_int, [_H]),
    ("lsm_set_stream", C.c_int, [_H, C.c_void_p]),
    ("lsm_layout", C.c_int, [_H, C.POINTER(LsmLayout)]),
    ("lsm_upload", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("lsm_download", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("lsm_upload_f64", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("lsm_download_f64", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("lsm_fill_ghosts", C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p]),
    ("lsm_stage", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                            C.c_double, C.c_double, C.c_double, C.c_void_p]),
    ("lsm_stage_planes", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_double, C.c_double, C.c_double, C.c_int64, C.c_int64, C.c_void_p]),
    ("lsm_fill_ghosts_planes", C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    ("lsm_compute_cfl", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_double, C.POINTER(C.c_double)]),
    ("lsm_cfl_cache", C.c_int, [_H, C.c_int]),
    ("lsm_advance_fe", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                 StageHook, C.c_void_p]),
    ("lsm_advance_rk2", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                  C.c_double, StageHook, C.c_void_p]),
    ("lsm_advance_rk3", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                  C.c_double, StageHook, C.c_void_p]),
    ("lsm_comm_unique_id", C.c_int, [C.c_void_p]),
    ("lsm_comm_attach_rccl", C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    ("lsm_comm_attach_local", C.c_int, [C.POINTER(_H), C.c_int]),
    ("lsm_comm_detach", C.c_int, [_H]),
    ("lsm_comm_info", C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("lsm_comm_set_overlap", C.c_int, [_H, C.c_int]),
    ("lsm_halo_start", C.c_int, [_H, C.c_void_p]),
    ("lsm_halo_wait", C.c_int, [_H]),
    ("lsm_halo_exchange", C.c_int, [_H, C.c_void_p]),
    ("lsm_allreduce_dt", C.c_int, [_H, C.POINTER(C.c_double)]),
    ("lsm_comm_abort", C.c_int, [_H]),
    ("lsm_band_overlap_config", C.c_int, [_H, C.c_int64]),
    ("lsm_band_overlap_mask", C.c_int, [_H, C.c_void_p]),
    ("lsm_band_overlap_values", C.c_int, [_H, C.c_void_p]),
    ("lsm_advance_band_fe", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.POINTER(LsmBand), C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                      StageHook, C.c_void_p]),
    ("lsm_advance_band_rk2", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.POINTER(LsmBand), C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                       C.c_double, StageHook, C.c_void_p]),
    ("lsm_advance_band_rk3", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.POINTER(LsmBand), C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                       C.c_double, StageHook, C.c_void_p]),
    ("lsm_eikonal_sign", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("lsm_extrema", C.c_int, [_H, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("lsm_check_range", C.c_int, [_H, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("lsm_geometry", C.c_int, [_H, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_void_p]),
    ("lsm_band_geometry", C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    ("lsm_interpolate", C.c_int, [_H, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("lsm_sdf_create", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_void_p),
                                 C.POINTER(C.c_int64)]),
    ("lsm_sdf_eval", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    ("lsm_sdf_samples", C.c_int, [C.c_void_p, C.c_void_p]),
    ("lsm_sdf_destroy", None, [C.c_void_p]),
    ("lsm_extend_along_normals", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_int, C.c_double, C.c_double, C.c_double]),
    ("lsm_band_tile_count", C.c_int, [_H, C.c_int, C.POINTER(C.c_int64)]),
    ("lsm_band_update", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_int64, C.c_void_p]),
    ("lsm_band_halo", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    ("lsm_band_retile", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_int]),
    ("lsm_band_invalidate", C.c_int, [_H]),
    ("lsm_set_tuning", C.c_int, [_H, C.c_char_p, C.c_int]),
    ("lsm_get_tuning", C.c_int, [_H, C.c_char_p, C.POINTER(C.c_int)]),
    ("lsm_band_fill_list", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ("lsm_reinitialize", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("lsm_band_prepare", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]),
    ("lsm_band_status", C.c_int, [_H, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    ("lsm_band_fill", C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    ("lsm_band_count", C.c_int, [_H, C.c_void_p, C.POINTER(C.c_int64)]),
    ("lsm_band_missed", C.c_int, [_H, C.POINTER(C.c_int)]),
    ("lsm_stage_band", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("lsm_compute_cfl_band", C.c_int, [_H, C.POINTER(LsmTerm), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                       C.POINTER(C.c_double)]),
    ("lsm_volume", C.c_int, [_H, C.c_void_p, C.POINTER(C.c_double)]),
    ("lsm_perimeter", C.c_int, [_H, C.c_void_p, C.POINTER(C.c_double)]),
    ("lsm_band_volume", C.c_int, [_H, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    ("lsm_band_perimeter", C.c_int, [_H, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    ("lsm_profile_enable", C.c_int, [_H, C.c_int]),
    ("lsm_profile_read", C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
]
EXPORTS = [s[0] for s in _SIGS]


STAGE_KERNEL_SOURCES = ("stage_kernel.h", "stage_math.h", "stage_tu.hip", "lsm_internal.h", "Makefile")


def source_hash():
    """sha256 over the sources the fused stage kernels are compiled from (csrc/stage_kernel.h, stage_math.h,
    stage_tu.hip, lsm_internal.h, Makefile): ties a committed profile of that kernel to the build it was measured on
    (the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    for f in STAGE_KERNEL_SOURCES:
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()


def build(jobs=8):
    """Compile libhiplsm.so for gfx950 with hipcc (csrc/Makefile); works without a GPU."""
    subprocess.check_call(["make", "-s", "-j", str(jobs), "-C", CSRC])


_lib = None


def lib():
    """Load libhiplsm.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). This package has no CPU fallback.")
        # torch ships its own HIP/HSA runtime libraries and must bring them in first: if libhiplsm.so pulls in
        # /opt/rocm's copies before torch is imported, the process ends up with two HSA runtimes and the second
        # one finds no device ("no ROCm-capable device is detected")
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in _SIGS:
            fn = getattr(L, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class LsmError(RuntimeError):
    pass


class LsmCommError(LsmError):
    """LSM_ERR_COMM: a peer rank left, aborted or did not answer in time; the communicator stays failed."""


def check(handle, code, what=""):
    if code != OK:
        msg = lib().lsm_last_error(handle)
        raise (LsmCommError if code == ERR_COMM else LsmError)(f"{what} failed ({code}): {msg.decode() if msg else ''}")

"""Diagnose library loading on the GPU box: HIP runtime status before/after dlopen of libhiplsm.so."""
import ctypes as C, os, sys
hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
n = C.c_int(0)
r = hip.hipGetDeviceCount(C.byref(n)); print("before dlopen: rc", r, hip.hipGetErrorString(r), "count", n.value)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "levelsetmethods.jl_amd", "libhiplsm.so")
lib = C.CDLL(os.path.abspath(path))
r = hip.hipGetDeviceCount(C.byref(n)); print("after dlopen: rc", r, hip.hipGetErrorString(r), "count", n.value)
r = hip.hipGetLastError(); print("last error", r, hip.hipGetErrorString(r))

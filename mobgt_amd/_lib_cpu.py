"""ctypes binding of libmobgt_cpu.so (include/mobgt_cpu.h): the fork-safe host half of the boundary.  Plain C++ --
no HIP, no threads -- so it may be loaded and called inside forked DataLoader workers."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmobgt_cpu.so")
CSRC = os.path.join(_HERE, "csrc_cpu")

_c = ctypes
_vp, _i = _c.c_void_p, _c.c_int

SIGNATURES = {
    "mobgt_cpu_abi_version": (_i, []),
    "mobgt_floyd_warshall_cpu": (_i, [_vp, _i, _vp, _vp]),
    "mobgt_gen_edge_input_cpu": (_i, [_i, _vp, _vp, _i, _i, _vp]),
    "mobgt_get_all_edges_cpu": (_i, [_vp, _i, _i, _i, _vp, _i, _vp]),
}
EINDEX, ERECURSION, ENOMEM = 1, 3, 4

_lib = None


def build(force=False):
    srcs = [os.path.join(CSRC, "algos_cpu.cpp"), os.path.join(os.path.dirname(_HERE), "include", "mobgt_cpu.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", CSRC])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib

"""ORACLE (test infrastructure): ctypes front end of oracle/algos_ref.c, same call signatures as the
reference's `algos` module (algos.pyx:9,57,65)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    so = os.path.join(_HERE, "liboracle_algos.so")
    src = os.path.join(_HERE, "algos_ref.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle_algos.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        i64p = ctypes.POINTER(ctypes.c_int64)
        lib.oracle_floyd_warshall.argtypes = [i64p, ctypes.c_int, i64p, i64p]
        lib.oracle_gen_edge_input.argtypes = [ctypes.c_int, i64p, i64p, ctypes.c_int, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_float)]
        lib.oracle_get_all_edges.argtypes = [i64p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        _LIB = lib
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def floyd_warshall(adjacency_matrix):
    """algos.pyx:9-54 -> (M, path) int64 [n,n]."""
    nrows, ncols = adjacency_matrix.shape
    assert nrows == ncols
    adj = np.ascontiguousarray(adjacency_matrix.astype(np.int64))
    M = np.empty((nrows, nrows), np.int64)
    path = np.empty((nrows, nrows), np.int64)
    _lib().oracle_floyd_warshall(_p(adj, ctypes.c_int64), nrows, _p(M, ctypes.c_int64), _p(path, ctypes.c_int64))
    return M, path


def get_all_edges(path, i, j):
    """algos.pyx:57-62."""
    path = np.ascontiguousarray(np.asarray(path).astype(np.int64))
    n = path.shape[0]
    buf = np.empty(n + 4, np.int32)
    ln = _lib().oracle_get_all_edges(_p(path, ctypes.c_int64), n, int(i), int(j), _p(buf, ctypes.c_int), n + 2)
    if ln < 0:
        raise RecursionError("path matrix does not terminate")
    return [int(v) for v in buf[:ln]]


def gen_edge_input(max_dist, path, edge_feat):
    """algos.pyx:65-96 -> float32 [n,n,max_dist,F], fill -1."""
    nrows, ncols = path.shape
    assert nrows == ncols
    path = np.ascontiguousarray(np.asarray(path).astype(np.int64))
    feat = np.ascontiguousarray(np.asarray(edge_feat).astype(np.int64))
    F = feat.shape[-1]
    out = np.empty((nrows, nrows, int(max_dist), F), np.float32)
    rc = _lib().oracle_gen_edge_input(int(max_dist), _p(path, ctypes.c_int64), _p(feat, ctypes.c_int64), nrows, F,
                                      _p(out, ctypes.c_float))
    if rc == 1:
        raise IndexError("a path has more hops than max_dist")
    if rc:
        raise MemoryError
    return out

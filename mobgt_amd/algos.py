"""Drop-in for the reference's only native module, `graphormer/algos.pyx`.

Same call signatures and return types (numpy in, numpy out) as `algos.floyd_warshall` (:9), `algos.get_all_edges`
(:57) and `algos.gen_edge_input` (:65).  Two native back ends sit behind them:

* **host** -- `libmobgt_cpu.so` (include/mobgt_cpu.h, plain C++).  The reference calls these functions per sample
  from forked DataLoader workers (`wrapper.py:55-60` under `data.py:282-295`, `--num_workers 8`), where HIP must not
  be touched; this is what runs there, and in any process that has not initialised the GPU.
* **device** -- the HIP kernels of `csrc/spd.hip` through `include/mobgt_hip.h`, used when the calling process
  already owns an initialised GPU context (and is not a fork of one).

`MOBGT_ALGOS_BACKEND=host|device` forces one.  Neither is a fallback of the other -- both are product code with the
same bit-exact results (tests/test_host_logic.py, tests/test_gpu_model.py against golden G1).  The batched path the
trainer uses for whole padded batches is `mobgt_amd.ops.spd_batched`.
"""
import ctypes
import os

import numpy as np

from . import _lib_cpu


def backend():
    """'device' when this process may use its GPU context for per-item calls, else 'host'."""
    forced = os.environ.get("MOBGT_ALGOS_BACKEND")
    if forced in ("host", "device"):
        return forced
    import torch
    if torch.cuda.is_initialized() and not torch.cuda._is_in_bad_fork():
        return "device"
    return "host"


def _i64(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.int64))


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


# ----------------------------------------------------------------------------------------------- host back end
def _fw_host(adj):
    n = adj.shape[0]
    M = np.empty((n, n), dtype=np.int64)
    path = np.empty((n, n), dtype=np.int64)
    rc = _lib_cpu.lib().mobgt_floyd_warshall_cpu(_ptr(adj), n, _ptr(M), _ptr(path))
    if rc == _lib_cpu.ENOMEM:
        raise MemoryError("mobgt_floyd_warshall_cpu")
    if rc:
        raise RuntimeError(f"mobgt_floyd_warshall_cpu failed ({rc})")
    return M, path


def _edge_input_host(max_dist, p, f):
    n, F = p.shape[0], f.shape[-1]
    out = np.empty((n, n, max_dist, F), dtype=np.float32)
    rc = _lib_cpu.lib().mobgt_gen_edge_input_cpu(max_dist, _ptr(p), _ptr(f), n, F, _ptr(out))
    if rc == _lib_cpu.EINDEX:
        raise IndexError("gen_edge_input: a shortest path has more hops than max_dist")
    if rc == _lib_cpu.ERECURSION:
        raise RecursionError("gen_edge_input: path matrix does not terminate")
    if rc:
        raise RuntimeError(f"mobgt_gen_edge_input_cpu failed ({rc})")
    return out


def _all_edges_host(p, i, j):
    n = p.shape[0]
    out = np.empty(n + 2, dtype=np.int32)
    ln = ctypes.c_int32(0)
    rc = _lib_cpu.lib().mobgt_get_all_edges_cpu(_ptr(p), n, int(i), int(j), _ptr(out), n + 2, ctypes.byref(ln))
    if rc == _lib_cpu.ERECURSION or rc == _lib_cpu.EINDEX:
        raise RecursionError("get_all_edges: path matrix does not terminate")
    if rc:
        raise RuntimeError(f"mobgt_get_all_edges_cpu failed ({rc})")
    return [int(v) for v in out[: ln.value]]


# --------------------------------------------------------------------------------------------- device back end
def _dev():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("mobgt_amd.algos: the device back end needs a GPU (MOBGT_ALGOS_BACKEND=host runs on the CPU)")
    return torch.device("cuda", torch.cuda.current_device())


def _fw_device(adj):
    import torch
    from . import _lib
    from .ops import _p, _stream
    n = adj.shape[0]
    dev = _dev()
    a = torch.from_numpy(adj).to(dev)
    M = torch.empty(n, n, dtype=torch.int64, device=dev)
    path = torch.empty(n, n, dtype=torch.int64, device=dev)
    work = torch.empty(int(_lib.lib().mobgt_floyd_warshall_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().mobgt_floyd_warshall(_p(a), n, _p(M), _p(path), _p(work), _stream()), "mobgt_floyd_warshall")
    return M.cpu().numpy(), path.cpu().numpy()


def _edge_input_device(max_dist, p, f):
    import torch
    from . import _lib
    from .ops import _p, _stream
    n, F = p.shape[0], f.shape[-1]
    dev = _dev()
    pd, fd = torch.from_numpy(p).to(dev), torch.from_numpy(f).to(dev)
    out = torch.empty(n, n, max_dist, F, dtype=torch.float32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mobgt_gen_edge_input(max_dist, _p(pd), _p(fd), n, F, _p(out), _p(err), _stream()),
               "mobgt_gen_edge_input")
    code = int(err.item())
    if code in (1, 2):
        raise IndexError("gen_edge_input: a shortest path has more hops than max_dist")
    if code:
        raise RecursionError("gen_edge_input: path matrix does not terminate")
    return out.cpu().numpy()


def _all_edges_device(p, i, j):
    import torch
    from . import _lib
    from .ops import _p, _stream
    n = p.shape[0]
    dev = _dev()
    pd = torch.from_numpy(p).to(dev)
    out = torch.empty(n + 2, dtype=torch.int32, device=dev)
    ln = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(2 * n + 4, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mobgt_get_all_edges(_p(pd), n, int(i), int(j), _p(out), _p(ln), _p(work), _stream()),
               "mobgt_get_all_edges")
    k = int(ln.item())
    if k < 0:
        raise RecursionError("get_all_edges: path matrix does not terminate")
    return [int(v) for v in out[:k].cpu().numpy()]


# ------------------------------------------------------------------------------------------------ reference API
def floyd_warshall(adjacency_matrix):
    """algos.pyx:9-54 -> (M, path), int64 [n,n]; unreachable = 510."""
    (nrows, ncols) = adjacency_matrix.shape
    assert nrows == ncols
    adj = _i64(adjacency_matrix)
    return _fw_device(adj) if backend() == "device" else _fw_host(adj)


def gen_edge_input(max_dist, path, edge_feat):
    """algos.pyx:65-96 -> float32 [n,n,max_dist,F], -1 where there is no hop."""
    (nrows, ncols) = path.shape
    assert nrows == ncols
    p, f = _i64(path), _i64(edge_feat)
    max_dist = int(max_dist)
    return _edge_input_device(max_dist, p, f) if backend() == "device" else _edge_input_host(max_dist, p, f)


def get_all_edges(path, i, j):
    """algos.pyx:57-62: intermediate nodes of the path i -> j (node 0 reads as 'no intermediate')."""
    p = _i64(path)
    return _all_edges_device(p, i, j) if backend() == "device" else _all_edges_host(p, i, j)

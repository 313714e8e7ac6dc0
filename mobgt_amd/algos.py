"""Drop-in for the reference's only native module, `graphormer/algos.pyx`, running on the GPU.

Same call signatures and return types (numpy in, numpy out) as `algos.floyd_warshall` (:9),
`algos.get_all_edges` (:57) and `algos.gen_edge_input` (:65); the work is done by the HIP kernels of
`csrc/spd.hip` through the C ABI.  The batched path used by the data pipeline is
`mobgt_amd.ops.spd_batched` (one launch for a whole padded batch); these per-graph wrappers exist so
that `wrapper.preprocess_item` keeps working unchanged.  Must be called from a process that may
touch the GPU (i.e. not from forked DataLoader workers).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check
from .ops import _p, _stream


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("mobgt_amd.algos runs on the GPU only (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def floyd_warshall(adjacency_matrix):
    """algos.pyx:9-54 -> (M, path), int64 [n,n]; unreachable = 510."""
    (nrows, ncols) = adjacency_matrix.shape
    assert nrows == ncols
    n = nrows
    dev = _dev()
    adj = torch.from_numpy(np.ascontiguousarray(np.asarray(adjacency_matrix).astype(np.int64))).to(dev)
    M = torch.empty(n, n, dtype=torch.int64, device=dev)
    path = torch.empty(n, n, dtype=torch.int64, device=dev)
    work = torch.empty(int(_lib.lib().mobgt_floyd_warshall_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    check(_lib.lib().mobgt_floyd_warshall(_p(adj), n, _p(M), _p(path), _p(work), _stream()), "mobgt_floyd_warshall")
    return M.cpu().numpy(), path.cpu().numpy()


def gen_edge_input(max_dist, path, edge_feat):
    """algos.pyx:65-96 -> float32 [n,n,max_dist,F], -1 where there is no hop."""
    (nrows, ncols) = path.shape
    assert nrows == ncols
    n, max_dist = nrows, int(max_dist)
    dev = _dev()
    p = torch.from_numpy(np.ascontiguousarray(np.asarray(path).astype(np.int64))).to(dev)
    f = torch.from_numpy(np.ascontiguousarray(np.asarray(edge_feat).astype(np.int64))).to(dev)
    F = f.shape[-1]
    out = torch.empty(n, n, max_dist, F, dtype=torch.float32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    check(_lib.lib().mobgt_gen_edge_input(max_dist, _p(p), _p(f), n, F, _p(out), _p(err), _stream()),
          "mobgt_gen_edge_input")
    code = int(err.item())
    if code in (1, 2):
        raise IndexError("gen_edge_input: a shortest path has more hops than max_dist")
    if code:
        raise RecursionError("gen_edge_input: path matrix does not terminate")
    return out.cpu().numpy()


def get_all_edges(path, i, j):
    """algos.pyx:57-62: intermediate nodes of the path i -> j (node 0 reads as 'no intermediate')."""
    p = np.asarray(path)
    n = p.shape[0]
    dev = _dev()
    pd = torch.from_numpy(np.ascontiguousarray(p.astype(np.int64))).to(dev)
    out = torch.empty(n + 2, dtype=torch.int32, device=dev)
    ln = torch.zeros(1, dtype=torch.int32, device=dev)
    work = torch.empty(2 * n + 4, dtype=torch.int32, device=dev)
    check(_lib.lib().mobgt_get_all_edges(_p(pd), n, int(i), int(j), _p(out), _p(ln), _p(work), _stream()),
          "mobgt_get_all_edges")
    k = int(ln.item())
    if k < 0:
        raise RecursionError("get_all_edges: path matrix does not terminate")
    return [int(v) for v in out[:k].cpu().numpy()]

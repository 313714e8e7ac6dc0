"""`graphormer/modelGNN.py:21-74`: GraphConvolution / GCN with a dense normalised adjacency.  These are
library GEMMs (hipBLASLt through torch), not hand kernels; they produce the POI / category tables the
node-feature gathers read (model_fqandtoyo.py:1236-1237) and run inside every train step.  Parameter
names and init match the reference.  Two MI355X-minded changes in HOW the same maths is evaluated:

* the first layer's input X is a constant, so `A (X W)` is evaluated as `(A X) W` with `A X` computed once
  (fp32) — that removes one P x P product from the forward and one from the backward of every step;
* weight gradients `X^T dS` have a tiny output and K = P (7856): a plain GEMM call runs them on one or two
  workgroups (measured 153 us each), so they are evaluated split-K as a batched GEMM + a sum.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter


def mm_tn_splitk(x, g, max_chunks=32, bf16_operands=False, dw=None):
    """x^T @ g for tall-skinny x [P,a], g [P,b] (P >> a,b): split the long K axis over a batch dimension.
    `bf16_operands` (the bf16 configuration): the split-K MFMA kernel of csrc/wgrad.hip instead, f32 operands rounded
    to bf16 while loading, f32 accumulate (21 us + a reduce -> ~8 us for the [64 x 7856] x [7856 x 192] case)."""
    if (bf16_operands and x.is_cuda and x.dtype == torch.float32 and g.dtype == torch.float32 and x.shape[1] % 2 == 0
            and g.shape[1] % 2 == 0 and x.is_contiguous() and g.is_contiguous()
            and x.data_ptr() % 8 == 0 and g.data_ptr() % 8 == 0):
        from . import ops
        return ops.linear_wgrad(x, g, leaf=True, dw=dw)[0]
    P = x.shape[0]
    s = max((c for c in range(1, max_chunks + 1) if P % c == 0), default=1)
    if s == 1 or P < 2048:
        return x.t() @ g
    return torch.bmm(x.view(s, P // s, -1).transpose(1, 2), g.view(s, P // s, -1)).sum(0)


_OUT_DTYPE = [None]


def _mm_f32(a, b, bias=None):
    """a @ b (+ bias) with an fp32 result for fp32 or bf16 operands: one GEMM launch, no cast / add kernels."""
    if a.dtype == torch.float32:
        return torch.addmm(bias, a, b) if bias is not None else a @ b
    if _OUT_DTYPE[0] is None:
        try:
            torch.mm(a[:8, :8].contiguous(), b[:8, :8].contiguous(), out_dtype=torch.float32)
            _OUT_DTYPE[0] = True
        except Exception:
            _OUT_DTYPE[0] = False
    if _OUT_DTYPE[0]:
        if bias is not None:
            return torch.addmm(bias, a, b, out_dtype=torch.float32)
        return torch.mm(a, b, out_dtype=torch.float32)
    out = (a @ b).float()
    return out + bias if bias is not None else out


import os as _os
_SMALL_GEMM = True


def _mm_small(a, b, b_is_nk=False, out_dtype=torch.float32):
    """a @ b (or a @ b.T) in f32.  Short contractions (K <= 64: the GCN's `input @ weight` and its data gradient) go to the
    one-wave-per-tile MFMA kernel of csrc/sgemm.hip; the library's kernels for such shapes are all ramp-up."""
    if _SMALL_GEMM and a.is_cuda and a.shape[1] <= 64 and a.dtype == torch.float32 and b.dtype == torch.float32 \
            and a.stride(1) == 1 and b.stride(1) == 1:
        from . import ops
        return ops.small_gemm(a, b, None, b_is_nk, out_dtype=out_dtype)
    return (a @ (b.t() if b_is_nk else b)).to(out_dtype)


def _colsum(g):
    if g.is_cuda:
        from . import ops
        return ops.colsum(g)
    return g.sum(0)


class _GraphConvFn(torch.autograd.Function):
    """out = adj @ (x @ W) + b   (modelGNN.py:38-44), with the big product in adj's dtype."""

    @staticmethod
    def forward(ctx, x, weight, bias, adj, adj_t=None):
        support = _mm_small(x, weight, out_dtype=adj.dtype)         # (bf16 adjacency: rounded on the way out, no cast launch)
        out = _mm_f32(adj, support, bias)                           # bias in the GEMM epilogue, fp32 result
        ctx.save_for_backward(x, weight, adj, adj_t)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, adj, adj_t = ctx.saved_tensors
        # adj is a constant: with a stored transpose the backward product streams rows like the forward one
        # (measured 60 us for adj.t() @ g through the library's transposed-operand path vs 42 us)
        d_support = _mm_f32(adj_t if adj_t is not None else adj.t(), g.to(adj.dtype))
        dW = mm_tn_splitk(x, d_support, bf16_operands=adj.dtype == torch.bfloat16)
        dx = _mm_small(d_support, weight, True) if ctx.needs_input_grad[0] else None
        db = _colsum(g) if ctx.has_bias else None
        return dx, dW, db, None, None


class _RowsConvFn(torch.autograd.Function):
    """out = adj[rows] @ (x @ W) + b for R << P rows of a bf16 adjacency, both skinny products on the split-K MFMA
    kernel of csrc/wgrad.hip instead of the library:
      forward   adj[rows] @ s  is  (adj[rows]^T)^T @ s  -- "g^T x" with g = adj[rows]^T [P,R], x = s [P,C]
                (K = P rows, a 608 x 192 output: the library ran it on ~30 workgroups, 51 us; here ~8 us + a
                9 MB transpose);
      backward  d_s = adj[rows]^T @ g  is "g^T x" with g = adj[rows] [R,P], x = dout [R,C]  (59 us -> ~18 us)."""

    @staticmethod
    def forward(ctx, x, weight, bias, adj, rows):
        from . import ops
        support = _mm_small(x, weight, out_dtype=torch.bfloat16)       # [P,C], rounded to bf16 on the way out
        a_rows, a_rows_t = ops.gather_rows_t(adj, rows)                # [R,P] and its transpose [P,R], one pass
        out = ops.linear_wgrad(a_rows_t, support, out_bias=bias)[0]    # [R,C] f32 (+ bias, inside the kernel)
        ctx.save_for_backward(x, weight, a_rows)
        ctx.has_bias = bias is not None
        ctx.sinks = (ops.grad_sink(weight), ops.grad_sink(bias) if bias is not None else None)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import ops
        from . import _lib
        from .ops import _p, _stream
        x, weight, a_rows = ctx.saved_tensors
        g = g.contiguous()
        R, P = a_rows.shape
        C = g.shape[1]
        if g.dtype == torch.float32 and C % 2 == 0 and g.data_ptr() % 8 == 0:
            # the f32 gradient as it arrives: rounded to bf16 while loading, bias gradient = its column sums, one launch
            d_support = ops.zeros_f32((P, C), g.device)
            db = (ctx.sinks[1][:] if ctx.sinks[1] is not None else ops.zeros_f32((C,), g.device)) if ctx.has_bias else None
            _lib.check(_lib.lib().mobgt_linear_wgrad_mixed(_p(a_rows), a_rows.stride(0), _p(g), g.stride(0), _p(d_support), C,
                                                           _p(db), R, P, C, _stream()), "mobgt_linear_wgrad_mixed")
        else:
            d_support = ops.linear_wgrad(a_rows, g.to(torch.bfloat16).contiguous())[0]      # [P,C] f32
            db = _colsum(g) if ctx.has_bias else None
        dW = mm_tn_splitk(x, d_support, bf16_operands=True, dw=ctx.sinks[0][:] if ctx.sinks[0] is not None else None)
        dx = _mm_small(d_support, weight, True) if ctx.needs_input_grad[0] else None
        return dx, dW, db, None, None


def _rows_conv_ok(x, adj, rows):
    return (x.is_cuda and adj.dtype == torch.bfloat16 and adj.shape[1] % 8 == 0 and adj.stride(0) % 8 == 0
            and rows.numel() % 8 == 0 and rows.numel() * 2 <= adj.shape[0] and rows.numel() <= 4096)


class MaskAdj:
    """The reference's normalised adjacency (D+I)^-1 (A+I) for a 0/1 matrix A as what it IS: a bit per entry plus one
    scale per row (csrc/maskgemm.hip).  `mask` / `mask_t`: uint32 words [P, ceil(P/32)] of A+I and of its transpose,
    `scale` f32 [P] = 1/(deg+1)."""

    def __init__(self, mask, mask_t, scale):
        self.mask, self.mask_t, self.scale = mask, mask_t, scale
        self.shape = (scale.numel(), scale.numel())
        self.dtype = torch.float32

    @staticmethod
    def from_dense01(adj01):
        """0/1 numpy matrix A (zero diagonal) -> (mask, mask_t, scale) CPU tensors, or None if A is not 0/1."""
        import numpy as np
        a = np.asarray(adj01)
        if not (np.all((a == 0) | (a == 1)) and not np.any(np.diagonal(a))):
            return None
        P = a.shape[0]
        m = (a != 0) | np.eye(P, dtype=bool)
        words = (P + 127) // 128 * 4                                   # whole groups of four 32-bit words per row

        def pack(b):
            by = np.packbits(b, axis=1, bitorder="little")
            pad = np.zeros((P, words * 4), dtype=np.uint8)
            pad[:, :by.shape[1]] = by
            return torch.from_numpy(pad.view(np.uint32).astype(np.int64).astype(np.int32).reshape(P, words))
        scale = torch.from_numpy((1.0 / (a.sum(axis=1).astype(np.float64) + 1.0)).astype(np.float32))
        return pack(m), pack(m.T.copy()), scale


_XT_WORK = {}


def xt_workspace(device, P, N, slot=0):
    """The bitmask product's operand buffer, bf16 [N, roundup(P, 128)], kept per (device, P, N, slot) and ZERO beyond column P
    (the kernels that fill it write columns < P only): a producer that writes its result there transposed
    (ops.bias_act(yt=), ops.small_gemm(ct=)) saves the transpose launch in front of `mask_gemm`."""
    key = (str(device), int(P), int(N), int(slot))
    t = _XT_WORK.get(key)
    if t is None:
        t = _XT_WORK[key] = torch.zeros(N, (P + 127) // 128 * 128, dtype=torch.bfloat16, device=device)
    return t


def mask_gemm(adj, x, transposed=False, bias=None, xt=None):
    """adj @ x (or adj^T @ x) -> f32 [P, N] through csrc/maskgemm.hip; x f32 [P, N], N in {16, 32, 48, 64}.
    `xt` (an `xt_workspace` that already holds x transposed -- for the transposed product: times adj.scale per row of x):
    x itself is not read."""
    from . import _lib
    from .ops import _p, _stream
    mask = adj.mask_t if transposed else adj.mask
    if xt is not None:
        N, P = xt.shape[0], adj.scale.numel()
        out = torch.empty(P, N, dtype=torch.float32, device=xt.device)
        _lib.check(_lib.lib().mobgt_mask_gemm(_p(mask), mask.shape[1], None, 0, None, _p(None if transposed else adj.scale),
                                              _p(bias), _p(out), N, _p(xt), P, P, N, _stream()), "mobgt_mask_gemm")
        return out
    x = x.contiguous()
    P, N = x.shape
    out = torch.empty(P, N, dtype=torch.float32, device=x.device)
    work = torch.empty(int(_lib.lib().mobgt_mask_gemm_workspace_bytes(P, N)), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.lib().mobgt_mask_gemm(_p(mask), mask.shape[1], _p(x), x.stride(0), _p(adj.scale if transposed else None),
                                          _p(None if transposed else adj.scale), _p(bias), _p(out), N, _p(work), P, P, N,
                                          _stream()), "mobgt_mask_gemm")
    return out


class _MaskConvFn(torch.autograd.Function):
    """out = adj @ (x @ W) + b evaluated as (adj @ x) @ W + b: the P x P product then has the layer's INPUT width (16 for
    the second GraphConvolution) instead of its output width (64), and so has its transposed twin in the backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, adj):
        ax = mask_gemm(adj, x)                                          # [P, in] f32
        out = _mm_small(ax, weight)                                     # [P, out]  (K = in <= 64)
        if bias is not None:
            out = out + bias
        ctx.save_for_backward(ax, weight)
        ctx.adj, ctx.has_bias = adj, bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        ax, weight = ctx.saved_tensors
        g = g.contiguous()
        dW = mm_tn_splitk(ax, g, bf16_operands=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dax = _mm_small(g, weight, True)                            # g @ W^T  [P, in]
            dx = mask_gemm(ctx.adj, dax, transposed=True)               # adj^T @ dax
        db = _colsum(g) if ctx.has_bias else None
        return dx, dW, db, None


class _ConvActFn(torch.autograd.Function):
    """A hidden GraphConvolution with its activation, y = dropout(leaky_relu((adj @ x) W + b)) (modelGNN.py:38-44, 66-72),
    without a launch for the activation either way: forward, bias + LeakyReLU + dropout are the small GEMM's epilogue;
    backward, the derivative m(y) is applied to the incoming gradient while the weight-gradient kernel and the data-gradient
    GEMM load it (bias gradient = column sums of the masked gradient, from the weight-gradient kernel).
    `ax`: the precomputed adj @ x of the first layer (x constant), else `adj` is a MaskAdj and adj @ x is a mask GEMM."""

    @staticmethod
    def forward(ctx, x, ax, weight, bias, adj, slope, p_drop, seed, seed_dev, salt, xt=None):
        from . import ops
        # (`xt`: x already transposed to the product's operand layout by the launch that produced it)
        # `ax` may be zero-padded to a whole number of 16-deep k-steps (a 303-wide input as 304 columns: 16-byte operand loads,
        # an even width for the weight-gradient kernel); then `xt` is where the NEXT bitmask product wants y transposed
        if ax is not None:
            t = ax
            y = ops.small_gemm(t, weight, bias, leaky=slope, drop=(p_drop, seed, seed_dev, salt) if p_drop > 0 else None,
                               k_b=weight.shape[0], ct=(xt.t, None, False) if xt is not None else None)
        else:
            t = mask_gemm(adj, x, xt=xt.t if xt is not None else None)  # [P, in] f32
            y = ops.small_gemm(t, weight, bias, leaky=slope, drop=(p_drop, seed, seed_dev, salt) if p_drop > 0 else None)
        ctx.save_for_backward(t, weight, y)
        ctx.adj = adj if ax is None else None
        ctx.mv = ops.act_mask_values(slope, p_drop)
        ctx.sinks = (ops.grad_sink(weight), ops.grad_sink(bias) if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import ops
        t, weight, y = ctx.saved_tensors
        g = g.contiguous()
        k_w, k_b = ctx.sinks
        db = k_b[:] if k_b is not None else ops.zeros_f32((weight.shape[1],), g.device)
        dw_dst = None
        extra = (t.shape[1] - weight.shape[0]) * weight.shape[1]
        if k_w is not None and k_w.is_contiguous() and 0 <= extra <= 16:
            # the parameter's gradient sink, seen with the rows of the zero padding (their zero products land in the slack that
            # train.flat_offsets leaves behind every slot)
            dw_dst = torch.as_strided(k_w, (t.shape[1], weight.shape[1]), (weight.shape[1], 1))
        dW = ops.linear_wgrad_masked(t, g, x_mask=y, mask_vals=ctx.mv, db=db, db_of_x=True, leaf=True, dw=dw_dst)   # t^T (g * m(y)), db = colsum
        dW = dW[:weight.shape[0]]                                                                # (rows of the zero padding)
        db = db[:]                                                                               # (a fresh view: see ops.wgrad_deferral)
        dx = None
        if ctx.adj is not None and ctx.needs_input_grad[0]:
            # dt = (g * m(y)) W^T [P, in] leaves the GEMM transposed, scaled by 1/(deg+1) and in bf16 -- the operand of
            # adj^T @ dt -- and nowhere else
            P, K = t.shape
            buf = xt_workspace(g.device, P, K, slot=1)
            ops.small_gemm(g, weight, b_is_nk=True, a_mask=(y, *ctx.mv), ct=(buf, ctx.adj.scale, True))
            dx = mask_gemm(ctx.adj, None, transposed=True, xt=buf)                               # adj^T @ dt
        return dx, None, dW, db, None, None, None, None, None, None, None


def _conv_act_ok(t_cols, weight, bias):
    return (os.environ.get("MOBGT_NO_CONV_ACT") != "1" and weight.is_cuda and weight.dtype == torch.float32 and bias is not None
            and weight.shape[0] == t_cols and t_cols <= 64 and t_cols % 2 == 0 and weight.shape[1] % 2 == 0
            and weight.is_contiguous())


def _mask_conv_ok(x, weight):
    return x.is_cuda and x.shape[1] in (16, 32, 48, 64) and weight.shape[0] == x.shape[1] and weight.shape[0] <= 64


class CsrAdj:
    """The normalised adjacency (D+I)^-1 (A+I) as CSR on the device, with the CSR of its transpose (the backward's
    `adj^T @ g` is then the same gather kernel).  rowptr int64 [P+1], col int32 [nnz], val f32 [nnz]."""

    def __init__(self, rowptr, col, val, t_rowptr, t_col, t_val):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.t_rowptr, self.t_col, self.t_val = t_rowptr, t_col, t_val
        self.shape = (rowptr.numel() - 1, t_rowptr.numel() - 1)
        self.dtype = torch.float32

    @staticmethod
    def from_scipy(a):
        """scipy.sparse matrix -> (six CPU tensors) in the order of the constructor."""
        a = a.tocsr()
        a.sort_indices()
        t = a.T.tocsr()
        t.sort_indices()
        f = lambda m: (torch.from_numpy(m.indptr.astype("int64")), torch.from_numpy(m.indices.astype("int32")),
                       torch.from_numpy(m.data.astype("float32")))
        return f(a) + f(t)


def spmm(adj, b, bias=None, rows=None, transposed=False):
    """adj[rows] @ b (+ bias) through csrc/spmm.hip; `transposed`: adj^T @ b (all rows)."""
    from . import _lib
    from .ops import _p, _stream
    rp, col, val = (adj.t_rowptr, adj.t_col, adj.t_val) if transposed else (adj.rowptr, adj.col, adj.val)
    b = b.contiguous()
    R = rows.numel() if rows is not None else rp.numel() - 1
    C = b.shape[1]
    out = torch.empty(R, C, dtype=torch.float32, device=b.device)
    _lib.check(_lib.lib().mobgt_spmm_csr(_p(rp), _p(col), _p(val), _p(rows), _p(b), b.stride(0), _p(bias), _p(out), C, R, C,
                                         _stream()), "mobgt_spmm_csr")
    return out


import os as _os_sp
_SP_GATHER = [True]                                                # False: the atomic scatter (tests)


class _SpConvFn(torch.autograd.Function):
    """out = adj[rows] @ (x @ W) + b with a CSR adjacency (modelGNN.py:38-44); rows = None: every row."""

    @staticmethod
    def forward(ctx, x, weight, bias, adj, rows):
        support = _mm_small(x, weight)                                  # [P, C] f32
        out = spmm(adj, support, bias, rows)
        ctx.save_for_backward(x, weight)
        ctx.adj, ctx.rows, ctx.has_bias = adj, rows, bias is not None
        from . import ops
        ctx.sinks = (ops.grad_sink(weight), None)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib, ops
        from .ops import _p, _stream
        x, weight = ctx.saved_tensors
        adj, rows = ctx.adj, ctx.rows
        g = g.contiguous()
        if rows is None:
            d_support = spmm(adj, g, transposed=True)                   # adj^T @ g: a gather over the stored transpose
        elif _SP_GATHER[0] and g.shape[1] % 4 == 0 and g.shape[1] <= 512:
            # adj[rows]^T @ g as a gather over the stored transpose (no atomics, every row written: no zero fill)
            P = adj.shape[1]
            head = getattr(adj, "_rows_head", None)
            if head is None or head.device != g.device:
                head = adj._rows_head = torch.full((adj.shape[0],), -1, dtype=torch.int32, device=g.device)
            nxt = torch.empty(rows.numel(), dtype=torch.int32, device=g.device)
            d_support = torch.empty(P, g.shape[1], dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().mobgt_spmm_csr_t_rows_gather(_p(adj.t_rowptr), _p(adj.t_col), _p(adj.t_val), _p(rows), _p(head),
                                                               _p(nxt), _p(g), g.stride(0), _p(d_support), d_support.stride(0),
                                                               P, rows.numel(), g.shape[1], _stream()),
                       "mobgt_spmm_csr_t_rows_gather")
        else:                                                           # adj[rows]^T @ g: scatter of R rows
            d_support = torch.zeros(adj.shape[1], g.shape[1], dtype=torch.float32, device=g.device)
            _lib.check(_lib.lib().mobgt_spmm_csr_t_rows(_p(adj.rowptr), _p(adj.col), _p(adj.val), _p(rows), _p(g), g.stride(0),
                                                        _p(d_support), d_support.stride(0), rows.numel(), g.shape[1],
                                                        _stream()), "mobgt_spmm_csr_t_rows")
        dW = mm_tn_splitk(x, d_support, bf16_operands=True, dw=ctx.sinks[0][:] if ctx.sinks[0] is not None else None)
        dx = _mm_small(d_support, weight, True) if ctx.needs_input_grad[0] else None
        db = _colsum(g) if ctx.has_bias else None
        return dx, dW, db, None, None


class _PreAggConvFn(torch.autograd.Function):
    """out = (adj @ x) @ W + b with the constant product ax = adj @ x supplied by the caller."""

    @staticmethod
    def forward(ctx, ax, weight, bias):
        out = torch.addmm(bias, ax, weight) if bias is not None else ax @ weight
        ctx.save_for_backward(ax)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        (ax,) = ctx.saved_tensors
        return None, mm_tn_splitk(ax, g), (_colsum(g) if ctx.has_bias else None)


class GraphConvolution(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, input, adj, adj_input=None, adj_t=None, bias=True, mask_adj=None):
        # X.W stays fp32 (raw features such as lat/lon need the mantissa); only the big dense adjacency
        # product runs in `adj`'s dtype (fp32, or bf16 in the bf16 configuration).  bias=False: the caller adds it
        # (GCN.forward fuses bias + LeakyReLU + dropout into one launch).
        b = self.bias if bias else None
        with torch.autocast(device_type=input.device.type, enabled=False):
            if adj_input is not None:
                return _PreAggConvFn.apply(adj_input, self.weight, b)
            if isinstance(adj, CsrAdj):
                return _SpConvFn.apply(input.float(), self.weight, b, adj, None)
            if mask_adj is not None and _mask_conv_ok(input, self.weight):
                return _MaskConvFn.apply(input.float(), self.weight, b, mask_adj)
            return _GraphConvFn.apply(input.float(), self.weight, b, adj, adj_t)


def _small_gcn_ok(gcn, x, adj, adj_x, adj_t):
    """The whole network as one launch each way (csrc/smallgcn.hip): a 3-layer GCN on a small dense f32 graph whose
    constant first product adj @ x and transpose are supplied."""
    from . import ops as _ops
    if os.environ.get("MOBGT_NO_SMALL_GCN") == "1" or _ops.SAFE_FORMS[0] or len(gcn.gcn) != 3:
        return False
    if not (torch.is_tensor(adj) and adj.is_cuda and adj.dtype == torch.float32 and adj.dim() == 2 and adj.is_contiguous()):
        return False
    if adj_x is None or adj_t is None or adj_x.dtype != torch.float32 or adj_t.dtype != torch.float32:
        return False
    n = adj.shape[0]
    if n > 1024 or adj.shape[1] != n or tuple(adj_t.shape) != (n, n) or adj_x.shape[0] != n:
        return False
    if adj_x.shape[1] != gcn.gcn[0].in_features or not adj_x.is_contiguous() or not adj_t.is_contiguous():
        return False
    h1, h2, h3 = (g.out_features for g in gcn.gcn)
    return all(g.bias is not None and g.weight.dtype == torch.float32 and g.weight.data_ptr() % 16 == 0 for g in gcn.gcn) and \
        (h1, h2, h3) == (16, 64, 32) and gcn.gcn[1].in_features == h1 and gcn.gcn[2].in_features == h2


def _small_gcn_launch(ax, a, ws, slope, p_drop, seed, seed_dev, salt):
    """The forward launch of the one-launch GCN on the current stream -> (keep = [h1 | t | h2 | t2], out)."""
    from . import _lib, ops
    from .ops import _p, _stream
    n, K0 = ax.shape
    H1, H2, H3 = ws[0].shape[1], ws[2].shape[1], ws[4].shape[1]
    keep = torch.empty(n * (2 * H1 + 2 * H2), dtype=torch.float32, device=ax.device)
    h1, t, h2, t2 = keep.split([n * H1, n * H1, n * H2, n * H2])
    out = torch.empty(n, H3, dtype=torch.float32, device=ax.device)
    counter = ops.zeros_f32((4,), ax.device)                 # zero bits = zero ints
    # a weight pack the model deferred (model.refresh_shadows(defer_pack=True)) rides along as passenger workgroups
    from .model import take_pending_pack
    import ctypes
    jobs = take_pending_pack()
    nj = len(jobs)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    pack = ((vp * nj)(*[j[0].data_ptr() for j in jobs]), (vp * nj)(*[j[1].data_ptr() for j in jobs]), (ci * nj)(*[j[2] for j in jobs]),
            (ci * nj)(*[j[3] for j in jobs]), (ci * nj)(*[j[4] for j in jobs])) if nj else (None, None, None, None, None)
    # ... and the hop table's forward / the node features' index derivation (ops.front_deferral)
    hop, ni = ops.take_front_jobs()
    front = ([1] + ni[0] if ni is not None else [0, None, 0, 0, 0, None, 0, 0, None, None, None, 0, None, None, 0, 0, 0]) + \
            ([1] + hop[0] if hop is not None else [0, None, None, None, 0, 0, 0, 0])
    _lib.check(_lib.lib().mobgt_small_gcn_fwd_pack(_p(ax), _p(a), *[_p(w) for w in ws], _p(h1), _p(t), _p(h2), _p(t2), _p(out),
                                                   _p(counter), n, K0, H1, H2, H3, slope, p_drop, seed, _p(seed_dev), salt,
                                                   nj, *pack, *front, _stream()), "mobgt_small_gcn_fwd_pack")
    return keep, out


def prelaunch_small_gcn(gcn, x, adj, adj_x, adj_t, same_stream=False):
    """same_stream: launch the one-launch GCN's forward NOW on the calling stream (no autograd: GCN.forward, called later where
    the reference calls it, finds the result and builds the autograd node there -- so the backward keeps its place behind the
    bias tables' node).  The model uses this to run the network FIRST in the step, with the front-of-step launches that precede
    its other consumers as passengers (the weight pack, ops.front_deferral's jobs).  -> True when launched.
    (Rounds 2-3 also had a side-stream form: inside the captured step its fork / join made the replay SLOWER -- S-FSQ 0.710 ms
    against 0.664 -- a cross-stream edge of a hipGraph costs more than the launch it hides; removed in round 5.)"""
    from . import ops
    if not same_stream or not (torch.is_tensor(x) and x.is_cuda):
        return False
    if adj_x is not None and adj_x.shape[1] != gcn.gcn[0].in_features:
        adj_x = adj_x[:, :gcn.gcn[0].in_features]
    if not _small_gcn_ok(gcn, x, adj, adj_x, adj_t):
        return False
    p_drop = gcn.dropout if gcn.training else 0.0
    seed, seed_dev = ops.dropout_seed(p_drop)
    g0, g1, g2 = gcn.gcn
    ws = [w.detach().contiguous() for w in (g0.weight, g0.bias, g1.weight, g1.bias, g2.weight, g2.bias)]
    salt = (0x2000 + g2.out_features) & 0xFFFFFFFF
    with torch.autocast(device_type="cuda", enabled=False):
        keep, out = _small_gcn_launch(adj_x, adj, ws, float(gcn.leaky_relu.negative_slope), float(p_drop), seed, seed_dev, salt)
    gcn._prelaunched = ((float(p_drop), seed, id(seed_dev), salt), (keep, out, None))
    return True


class _SmallGcnFn(torch.autograd.Function):
    """graphormer/modelGNN.py:53-74 for a small graph: forward and backward are one persistent launch each."""

    @staticmethod
    def forward(ctx, ax, a, a_t, w0, b0, w1, b1, w2, b2, slope, p_drop, seed, seed_dev, salt, pre=None):
        from . import ops
        n, K0 = ax.shape
        H1, H2, H3 = w0.shape[1], w1.shape[1], w2.shape[1]
        ws = [w.contiguous() for w in (w0, b0, w1, b1, w2, b2)]
        if pre is not None:
            keep, out, _ = pre                                   # launched earlier on this stream (prelaunch_small_gcn)
        else:
            keep, out = _small_gcn_launch(ax, a, ws, slope, p_drop, seed, seed_dev, salt)
        ctx.save_for_backward(ax, a_t, ws[2], ws[4], keep)
        ctx.misc = (slope, p_drop, seed, seed_dev, salt, n, K0, H1, H2, H3)
        ctx.sinks = [ops.grad_sink(w) for w in (w0, b0, w1, b1, w2, b2)]
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib, ops
        from .ops import _p, _stream
        ax, a_t, w1, w2, keep = ctx.saved_tensors
        slope, p_drop, seed, seed_dev, salt, n, K0, H1, H2, H3 = ctx.misc
        h1, t, h2, t2 = keep.split([n * H1, n * H1, n * H2, n * H2])
        shapes = [(K0, H1), (H1,), (H1, H2), (H2,), (H2, H3), (H3,)]
        # the kernel ACCUMULATES its workgroups' partial sums: into the trainer's (zeroed) flat gradient when it
        # registered one, else into fresh zeros
        grads = [s[:] if s is not None else ops.zeros_f32(shape, ax.device) for s, shape in zip(ctx.sinks, shapes)]
        scratch = torch.empty(n * (H1 + H2), dtype=torch.float32, device=ax.device)
        counter = ops.zeros_f32((4,), ax.device)
        # the bias tables' backward, if its inputs are complete, rides in this launch as passenger workgroups (ops.take_bias_bwd_job)
        job = ops.take_bias_bwd_job()
        if job is not None:
            outs, bias_args = ops.bias_bwd_job_args(job)
            extra = [1] + bias_args
        else:
            extra = [0, None, 0, 1, 0] + [None] * 8 + [0] * 9 + [0, 0, 0]
        _lib.check(_lib.lib().mobgt_small_gcn_bwd_bias(_p(g.contiguous()), _p(ax), _p(a_t), _p(w1), _p(w2), _p(h1), _p(t), _p(h2),
                                                       _p(t2), *[_p(x) for x in grads], _p(scratch[n * H1:]), _p(scratch[:n * H1]),
                                                       _p(counter), n, K0, H1, H2, H3, slope, p_drop, seed, _p(seed_dev), salt,
                                                       *extra, _stream()), "mobgt_small_gcn_bwd_bias")
        if job is not None:
            job["done"] = outs
        return (None, None, None, *grads, None, None, None, None, None, None)


class _DistGcnFn(torch.autograd.Function):
    """The distance GCN of the fq model (three GraphConvolutions 303 -> 16 -> 64 -> hidden over all P POIs, modelGNN.py:38-44,
    :66-72; model_fqandtoyo.py:1236) evaluated for the batch's POI rows only (:1264), as THREE launches forward and four
    backward (round 4; csrc/maskgemm.hip "Round 4"):
      forward   y0 = leaky((A X) W0 + b0)                         small GEMM on the precomputed, zero-padded A X (+ y0^T bf16)
                t1 = A y0,  y1 = dropout(leaky(t1 W1 + b1))       bitmask product with the layer in its epilogue (+ y1^T bf16)
                u = A[rows] y1,  out = u W2 + b2                  bitmask product on the batch's rows; out leaves as four
                                                                  partial tables (one per 16 columns of u) that the consumer adds
      backward  dW2 = u^T g, db2;  gu = g W2^T (scaled, transposed, bf16);  dy1 = A[rows]^T gu with dt1 = (dy1 m(y1)) W1^T
                in its epilogue;  dy0 = A^T dt1;  dW1 = t1^T (dy1 m(y1)), dW0 = (A X)^T (dy0 m(y0)) in the step's grouped
                weight-gradient launch.
    The last layer is reassociated -- (A[rows] y1) W2 instead of A[rows] (y1 W2): the P-row product has width 64 instead of
    `hidden`, and its operand is the bf16 rounding of y1 (was: of y1 W2)."""

    @staticmethod
    def forward(ctx, ax_pad, mask, mask_t, scale, rows, w0, b0, w1, b1, w2, b2, slope, p_drop, seed, seed_dev, salt):
        from . import ops, _lib
        from .ops import _p, _stream
        P, R, NO = ax_pad.shape[0], rows.numel(), w2.shape[1]
        dev = ax_pad.device
        y0t = xt_workspace(dev, P, 16, slot=0)
        y0 = ops.small_gemm(ax_pad, w0, b0, leaky=slope, k_b=w0.shape[0], ct=(y0t, None, False))
        y1t = xt_workspace(dev, P, 64, slot=0)
        t1 = torch.empty(P, 16, dtype=torch.float32, device=dev)
        y1 = torch.empty(P, 64, dtype=torch.float32, device=dev)
        u = torch.empty(R, 64, dtype=torch.float32, device=dev)
        parts = torch.empty(4, R, NO, dtype=torch.float32, device=dev)                # out = parts[0] + parts[1] + parts[2] + parts[3]
        rs_rows = torch.empty(R, dtype=torch.float32, device=dev)
        L = _lib.lib()
        _lib.check(L.mobgt_mask_gemm_l1_fwd(_p(mask), mask.shape[1], _p(scale), _p(y0t), y0t.stride(0), _p(t1), _p(w1), _p(b1),
                                            float(slope), float(p_drop), int(seed), _p(seed_dev), int(salt) & 0xFFFFFFFF, _p(y1),
                                            _p(y1t), y1t.stride(0), None, 0, P, P, _stream()), "mobgt_mask_gemm_l1_fwd")
        _lib.check(L.mobgt_mask_rows_fwd(_p(mask), mask.shape[1], _p(rows), _p(scale), _p(y1t), y1t.stride(0), _p(w2), _p(b2),
                                         _p(u), _p(parts), _p(rs_rows), R, P, NO, _stream()), "mobgt_mask_rows_fwd")
        ctx.save_for_backward(ax_pad, y0, t1, y1, u, rs_rows, rows, w0, w1, w2, mask_t, scale)
        ctx.mv0, ctx.mv1 = ops.act_mask_values(slope, 0.0), ops.act_mask_values(slope, p_drop)
        ctx.sinks = tuple(ops.grad_sink(t) for t in (w0, b0, w1, b1, w2, b2))
        # the last layer's output as the four partial tables of its column blocks: the first carries the gradient (d out /
        # d parts[q] = 1 for every q), the other three are constants to autograd
        p0, rest = parts[0], parts[1:]
        ctx.mark_non_differentiable(rest)
        ctx.set_materialize_grads(False)             # (no zero tensor for `rest`'s absent gradient)
        return p0, rest

    @staticmethod
    def backward(ctx, g, _g_rest=None):
        from . import ops, _lib
        from .ops import _p, _stream
        ax_pad, y0, t1, y1, u, rs_rows, rows, w0, w1, w2, mask_t, scale = ctx.saved_tensors
        k_w0, k_b0, k_w1, k_b1, k_w2, k_b2 = ctx.sinks
        if g is None:
            return (None,) * 16
        g = g.contiguous()
        dev = g.device
        P, R = ax_pad.shape[0], rows.numel()

        def dst(k, n):
            return k[:] if k is not None else ops.zeros_f32((n,), dev)
        db2 = dst(k_b2, w2.shape[1])
        dW2 = ops.linear_wgrad_masked(u, g, db=db2, db_of_x=True, leaf=True, dw=k_w2[:] if k_w2 is not None else None)
        gut = xt_workspace(dev, R, 64, slot=2)
        ops.small_gemm(g, w2, b_is_nk=True, ct=(gut, rs_rows, True))                 # (rs[rows] * (g W2^T))^T, bf16
        dy1 = torch.empty(P, 64, dtype=torch.float32, device=dev)
        dtt = xt_workspace(dev, P, 16, slot=1)
        _lib.check(_lib.lib().mobgt_mask_rows_bwd(_p(mask_t), mask_t.shape[1], _p(rows), _p(gut), gut.stride(0), _p(y1),
                                                  float(ctx.mv1[0]), float(ctx.mv1[1]), float(ctx.mv1[2]), _p(w1), _p(scale), _p(dy1),
                                                  _p(dtt), dtt.stride(0), R, P, _stream()), "mobgt_mask_rows_bwd")
        db1 = dst(k_b1, 64)
        dW1 = ops.linear_wgrad_masked(t1, dy1, x_mask=y1, mask_vals=ctx.mv1, db=db1, db_of_x=True, leaf=True,
                                      dw=k_w1[:] if k_w1 is not None else None)
        dy0 = mask_gemm(MaskAdj(None, mask_t, scale), None, transposed=True, xt=dtt)  # A^T dt1  [P,16]
        db0 = dst(k_b0, 16)
        dw_dst = None
        extra = (ax_pad.shape[1] - w0.shape[0]) * w0.shape[1]
        if k_w0 is not None and k_w0.is_contiguous() and 0 <= extra <= 16:
            # the parameter's gradient sink, seen with the rows of the zero padding (see _ConvActFn.backward)
            dw_dst = torch.as_strided(k_w0, (ax_pad.shape[1], w0.shape[1]), (w0.shape[1], 1))
        dW0 = ops.linear_wgrad_masked(ax_pad, dy0, x_mask=y0, mask_vals=ctx.mv0, db=db0, db_of_x=True, leaf=True, dw=dw_dst)
        dW0 = dW0[:w0.shape[0]]
        return (None, None, None, None, None, dW0, db0[:], dW1, db1[:], dW2, db2[:], None, None, None, None, None)


def _dist_gcn_ok(gcn, adj_x_pad, rows, mask_adj):
    """The shapes / layouts _DistGcnFn's kernels take (anything else keeps the launch-per-product path)."""
    if os.environ.get("MOBGT_NO_DIST_GCN_FUSED") == "1" or adj_x_pad is None or rows is None or mask_adj is None or len(gcn.gcn) != 3:
        return False
    g0, g1, g2 = gcn.gcn
    if not (adj_x_pad.is_cuda and adj_x_pad.dtype == torch.float32 and adj_x_pad.is_contiguous() and adj_x_pad.shape[1] % 16 == 0
            and 0 <= adj_x_pad.shape[1] - g0.in_features < 16):
        return False
    if (g0.out_features, g1.in_features, g1.out_features, g2.in_features) != (16, 16, 64, 64) or g2.out_features % 4 or g2.out_features > 192:
        return False
    for g in (g0, g1, g2):
        if g.bias is None or g.weight.dtype != torch.float32 or not g.weight.is_contiguous() or not g.bias.is_contiguous():
            return False
    P = adj_x_pad.shape[0]
    if rows.dtype != torch.int64 or not rows.is_contiguous() or rows.numel() < 1 or mask_adj.scale.numel() != P:
        return False
    from . import _lib
    return int(_lib.lib().mobgt_mask_rows_bwd_lds_bytes(mask_adj.mask_t.shape[1], rows.numel())) <= 120 * 1024


class GCN(nn.Module):
    def __init__(self, ninput, nhid, noutput, dropout):
        super().__init__()
        self.gcn = nn.ModuleList()
        self.dropout = dropout
        self.leaky_relu = nn.LeakyReLU(0.2)
        channels = [ninput] + nhid + [noutput]
        for i in range(len(channels) - 1):
            self.gcn.append(GraphConvolution(channels[i], channels[i + 1]))

    def forward(self, x, adj, adj_x=None, rows=None, adj_t=None, mask_adj=None, parts_ok=False):
        """`adj_x` = adj @ x precomputed (x is a constant feature matrix in MobGT): skips the first P x P product.
        `rows` (int64 [R]): return only these rows of the output table, i.e. evaluate the LAST layer as
        adj[rows] @ (h W) + b.  The model reads the table only at the batch's POI ids
        (model_fqandtoyo.py:1264), so for R << P this replaces a P x P product (and its transpose in the
        backward) by an R x P one without changing any value that is used.
        `parts_ok`: the caller accepts the result as `out` + the tensors in `out._mobgt_parts` (to be ADDED to it, in order)."""
        n_hidden = len(self.gcn) - 1
        adj_x_pad = None
        if adj_x is not None and adj_x.shape[1] != self.gcn[0].in_features:      # zero-padded columns (see _ConvActFn)
            adj_x_pad, adj_x = adj_x, adj_x[:, :self.gcn[0].in_features]
        if rows is None and mask_adj is None and x.is_cuda and _small_gcn_ok(self, x, adj, adj_x, adj_t):
            from . import ops
            p_drop = self.dropout if self.training else 0.0
            g0, g1, g2 = self.gcn
            salt = (0x2000 + g2.out_features) & 0xFFFFFFFF
            pre = self.__dict__.pop("_prelaunched", None)
            seed_dev = ops._DROPOUT_STATE["seed_dev"]
            if pre is not None and (pre[0][0], pre[0][2], pre[0][3]) == (float(p_drop), id(seed_dev), salt):
                seed = pre[0][1]                                        # (the masks the prelaunched pass drew)
            else:
                seed, seed_dev = ops.dropout_seed(p_drop)
                if pre is not None:
                    if pre[1][2] is not None:
                        torch.cuda.current_stream().wait_event(pre[1][2])   # (not what was prelaunched: drop it, but do not race with it)
                    pre = None
            with torch.autocast(device_type="cuda", enabled=False):
                return _SmallGcnFn.apply(adj_x, adj, adj_t, g0.weight, g0.bias, g1.weight, g1.bias, g2.weight, g2.bias,
                                         float(self.leaky_relu.negative_slope), float(p_drop), seed, seed_dev, salt,
                                         pre[1] if pre is not None else None)
        ax_in = adj_x_pad if adj_x_pad is not None else adj_x      # (a width that is already a whole number of k-steps: no padding)
        if x.is_cuda and not isinstance(adj, CsrAdj) and _dist_gcn_ok(self, ax_in, rows, mask_adj):
            from . import ops
            p_drop = self.dropout if self.training else 0.0
            seed, seed_dev = ops.dropout_seed(p_drop)
            g0, g1, g2 = self.gcn
            with torch.autocast(device_type="cuda", enabled=False):
                p0, rest = _DistGcnFn.apply(ax_in, mask_adj.mask, mask_adj.mask_t, mask_adj.scale, rows, g0.weight, g0.bias,
                                            g1.weight, g1.bias, g2.weight, g2.bias, float(self.leaky_relu.negative_slope),
                                            float(p_drop), seed, seed_dev, 0x2000 + g2.out_features)
            if parts_ok:                 # the caller adds the partial tables itself (model_fqandtoyo.node_features: in its gather)
                p0._mobgt_parts = tuple(rest.unbind(0))
                return p0
            return ((p0 + rest[0]) + rest[1]) + rest[2]
        if x.is_cuda and all(g.out_features % 4 == 0 and g.bias is not None for g in self.gcn[:-1]):
            from . import ops
            for i in range(n_hidden):       # bias + LeakyReLU (+ the dropout in front of the last layer) in one launch
                p_drop = self.dropout if (i == n_hidden - 1 and self.training) else 0.0
                salt = 0x2000 + self.gcn[-1].out_features
                gc = self.gcn[i]
                pre = adj_x if i == 0 else None
                pre_pad = adj_x_pad if i == 0 else None
                # (bitmask-adjacency layers only: that is the bf16 configuration, whose weight gradients already round
                # their operands to bf16; the first layer's f32 product and the f32 configuration keep their exact path)
                # first layer with a zero-padded A X (model_fqandtoyo: 303 -> 304 columns) in front of a bitmask layer: the same
                # node with `ax`, which also writes its result transposed for that next product
                if (pre_pad is not None and mask_adj is not None and not isinstance(adj, CsrAdj) and i + 1 < n_hidden
                        and pre_pad.dtype == torch.float32 and pre_pad.is_contiguous() and pre_pad.shape[1] % 16 == 0
                        and 0 <= pre_pad.shape[1] - gc.in_features < 16 and gc.bias is not None and gc.out_features in (16, 32, 48, 64)
                        and gc.weight.is_contiguous() and os.environ.get("MOBGT_NO_CONV_ACT") != "1"
                        and _conv_act_ok(gc.out_features, self.gcn[i + 1].weight, self.gcn[i + 1].bias)):
                    seed, seed_dev = ops.dropout_seed(p_drop)
                    yt = xt_workspace(pre_pad.device, pre_pad.shape[0], gc.out_features, slot=0)
                    with torch.autocast(device_type="cuda", enabled=False):
                        x = _ConvActFn.apply(None, pre_pad, gc.weight, gc.bias, None, float(self.leaky_relu.negative_slope),
                                             float(p_drop), seed, seed_dev, salt, ops._OutRef(yt))
                    x._mobgt_xt = yt
                    continue
                if (pre is None and mask_adj is not None and not isinstance(adj, CsrAdj) and x.dtype == torch.float32
                        and _mask_conv_ok(x, gc.weight) and _conv_act_ok(x.shape[1], gc.weight, gc.bias)):
                    # ... and no launch for the activation at all: GEMM epilogue forward, masked operand loads backward
                    seed, seed_dev = ops.dropout_seed(p_drop)
                    xt = getattr(x, "_mobgt_xt", None)
                    with torch.autocast(device_type="cuda", enabled=False):
                        x = _ConvActFn.apply(None if pre is not None else x, pre, gc.weight, gc.bias, mask_adj,
                                             float(self.leaky_relu.negative_slope), float(p_drop), seed, seed_dev, salt,
                                             ops._OutRef(xt) if xt is not None else None)
                    continue
                h = gc(x, adj, pre, adj_t, bias=False, mask_adj=mask_adj)
                # when the next layer is a bitmask product, the activation also leaves its result in that product's operand
                # layout (no transpose launch)
                nxt = self.gcn[i + 1] if i + 1 < n_hidden else None
                yt = None
                if (nxt is not None and mask_adj is not None and not isinstance(adj, CsrAdj) and h.dim() == 2
                        and h.shape[1] in (16, 32, 48, 64) and _conv_act_ok(h.shape[1], nxt.weight, nxt.bias)):
                    yt = xt_workspace(h.device, h.shape[0], h.shape[1], slot=0)
                x = ops.bias_act(h, gc.bias, self.leaky_relu.negative_slope, p_drop, self.training, salt, yt=yt)
                if yt is not None:
                    x._mobgt_xt = yt
        else:
            for i in range(n_hidden):
                x = self.leaky_relu(self.gcn[i](x, adj, adj_x if i == 0 else None, adj_t, mask_adj=mask_adj))
            if x.is_cuda:
                from . import ops
                x = ops.dropout(x, self.dropout, self.training, 0x2000 + self.gcn[-1].out_features)
            else:
                x = F.dropout(x, self.dropout, training=self.training)
        if isinstance(adj, CsrAdj):
            last = self.gcn[-1]
            with torch.autocast(device_type=x.device.type, enabled=False):
                return _SpConvFn.apply(x.float(), last.weight, last.bias, adj, rows)
        if rows is not None:
            last = self.gcn[-1]
            if _rows_conv_ok(x, adj, rows):
                with torch.autocast(device_type=x.device.type, enabled=False):
                    return _RowsConvFn.apply(x.float(), last.weight, last.bias, adj, rows)
            return last(x, adj.index_select(0, rows))
        return self.gcn[-1](x, adj, None, adj_t, mask_adj=mask_adj)

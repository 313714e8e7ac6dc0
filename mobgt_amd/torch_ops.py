"""`torch.library` registration of the hot path's central operator (SURVEY §8b "Threading / Autograd").

    torch.ops.mobgt.attention(q, k, v, attn_bias, num_heads, scale, dropout_p, seed) -> Tensor[G,T,C]

is `softmax((q * scale) k^T + attn_bias) v` per head with attention dropout -- `graphormer/model.py:436-455`
(= `model_fqandtoyo.py:1687-1706`) between the `linear_q/k/v` projections and `output_layer` -- on the HIP kernels of
`csrc/attn.hip`, as a dispatcher-visible custom op:

* autograd is registered with `torch.library.register_autograd` (backward = `mobgt::attention_backward`, itself an op),
  so the op works under `torch.no_grad`, `torch.inference_mode`, `torch.func` transforms that accept custom ops, and shows
  up by name in profiler traces;
* AMP: `torch.library.register_autocast("cuda", bfloat16)` -- under `torch.autocast` (the reference trains with
  `--precision 16`, README.md:62) q / k / v / bias are cast to bf16 and the kernels run their bf16-I/O instantiation
  (softmax statistics, accumulation and the log-sum-exp stay fp32 inside the kernel, as always);
* a fake (meta) implementation gives shapes / dtypes for tracing.

`mobgt_amd.model.MultiHeadAttention` and the fused encoder layer call the same C entry points through `mobgt_amd.ops`
with a pre-packed bias (`ops.PackedBias`, shared by all layers of a step); this op is the stand-alone form for callers
that hold a plain `[G,H,T,T]` bias tensor, e.g. the reference's own `MultiHeadAttention.forward`.
"""
from typing import Tuple

import torch

from . import ops

_LIB_NS = "mobgt"


@torch.library.custom_op(f"{_LIB_NS}::attention_forward", mutates_args=(), device_types="cuda")
def attention_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_bias: torch.Tensor, num_heads: int, scale: float,
                      dropout_p: float, seed: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (out [G,T,C], row statistics: 1-D f32 -- [G,H,T] log-sum-exp values and, for bf16 I/O, the output's bf16 rounding
    residual behind them (ops._lse_alloc; `ops.lse_rows` gives the [G,H,T] view))."""
    G, T, C = q.shape
    io = q.dtype if q.dtype in (torch.float32, torch.bfloat16) else torch.float32
    q, k, v = (t.to(io).contiguous() for t in (q, k, v))
    pack = ops.pack_bias(attn_bias.detach(), G, num_heads, T, dtype=torch.bfloat16 if attn_bias.dtype == torch.bfloat16 else torch.float32)
    out, lse = ops._attn_fwd(q, k, v, pack, float(scale), float(dropout_p), int(seed), None)
    return out, lse


@attention_forward.register_fake
def _(q, k, v, attn_bias, num_heads, scale, dropout_p, seed):
    G, T, C = q.shape
    io = q.dtype if q.dtype in (torch.float32, torch.bfloat16) else torch.float32
    return q.new_empty((G, T, C), dtype=io), q.new_empty((ops.lse_numel(G, num_heads, T, C, io),), dtype=torch.float32)


@torch.library.custom_op(f"{_LIB_NS}::attention_backward", mutates_args=(), device_types="cuda")
def attention_backward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_bias: torch.Tensor, out: torch.Tensor,
                       lse: torch.Tensor, dout: torch.Tensor, num_heads: int, scale: float, dropout_p: float,
                       seed: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (dq, dk, dv [G,T,C] in the I/O dtype, dbias [G,H,T,T] f32)."""
    G, T, C = q.shape
    io = out.dtype
    q, k, v, dout = (t.to(io).contiguous() for t in (q, k, v, dout))
    pack = ops.pack_bias(attn_bias.detach(), G, num_heads, T, dtype=torch.bfloat16 if attn_bias.dtype == torch.bfloat16 else torch.float32)
    pack.needs_grad, pack.n_use = True, 1
    dq, dk, dv = torch.empty_like(out), torch.empty_like(out), torch.empty_like(out)
    ops._attn_bwd(q, k, v, out, lse, dout, dq, dk, dv, pack, float(scale), float(dropout_p), int(seed), None)
    return dq, dk, dv, pack.grad_total().float().contiguous()


@attention_backward.register_fake
def _(q, k, v, attn_bias, out, lse, dout, num_heads, scale, dropout_p, seed):
    G, T, C = q.shape
    e = lambda: out.new_empty(out.shape)
    return e(), e(), e(), out.new_empty((G, num_heads, T, T), dtype=torch.float32)


def _setup_context(ctx, inputs, output):
    q, k, v, attn_bias, num_heads, scale, dropout_p, seed = inputs
    out, lse = output
    ctx.save_for_backward(q, k, v, attn_bias, out, lse)
    ctx.misc = (num_heads, scale, dropout_p, seed, q.dtype, k.dtype, v.dtype, attn_bias.dtype, tuple(attn_bias.shape))


def _backward(ctx, dout, _dlse):
    q, k, v, attn_bias, out, lse = ctx.saved_tensors
    num_heads, scale, dropout_p, seed, qd, kd, vd, bd, bshape = ctx.misc
    dq, dk, dv, dbias = torch.ops.mobgt.attention_backward(q, k, v, attn_bias, out, lse, dout, num_heads, scale, dropout_p, seed)
    if tuple(dbias.shape) != bshape:                       # the bias was broadcast over graphs / heads
        dbias = dbias.sum_to_size(bshape)
    return dq.to(qd), dk.to(kd), dv.to(vd), dbias.to(bd), None, None, None, None


torch.library.register_autograd(f"{_LIB_NS}::attention_forward", _backward, setup_context=_setup_context)
# AMP: run the bf16-I/O instantiation (fp16 autocast maps to bf16 as well: the kernels have no fp16 form, and bf16 is the
# MFMA operand type of the f32 instantiation anyway)
torch.library.register_autocast(f"{_LIB_NS}::attention_forward", "cuda", torch.bfloat16)


def attention(q, k, v, attn_bias, num_heads, scale=None, dropout_p=0.0, seed=0):
    """Functional form: `torch.ops.mobgt.attention_forward(...)[0]` with the reference's default scale d^-0.5
    (`model.py:415`); `attn_bias` broadcastable to [G,H,T,T]."""
    G, T, C = q.shape
    if scale is None:
        scale = (C // num_heads) ** -0.5
    if attn_bias.dim() == 3:
        attn_bias = attn_bias.unsqueeze(1)
    attn_bias = attn_bias.expand(G, num_heads, T, T)
    return torch.ops.mobgt.attention_forward(q, k, v, attn_bias, num_heads, float(scale), float(dropout_p), int(seed))[0]

"""One autograd node per EncoderLayer (graphormer/model.py:479-489, model_fqandtoyo.py:1731-1743).

The reference's layer is ~25 eager kernel launches forward and ~70 backward, almost all
of them tiny at MobGT's sizes (G*T = a few hundred rows).  Here the forward is
    [LN] -> QKV GEMM -> attention -> out GEMM -> dropout+residual+LN -> FFN GEMM -> GELU -> FFN GEMM ->
    dropout+residual[+LN]
i.e. 4 library GEMMs (bias in the epilogue), the HIP attention kernel and 3-4 fused HIP elementwise kernels
(`csrc/layer.hip`); the backward mirrors it with 8 GEMMs, 2 attention kernels and 4-5 fused kernels that also
produce every bias / LayerNorm gradient (column sums accumulated in registers), so no separate reduce,
dropout-mask, cast or accumulate kernels remain.  The residual stream and all statistics are fp32; the
GEMM-facing activations are fp32 or bf16 (`act_dtype`).
"""
import ctypes

import torch

from . import _lib, ops
from ._lib import check
from .ops import _p, _stream, _DT


_MM_OUT_DTYPE = [None]


def _split_factor(R, tiles):
    """Divisor s of R so that s * tiles fills the chip (>= ~512 workgroups) while each slice keeps K >= 32."""
    want = max(1, (512 + tiles - 1) // tiles)
    best = 1
    for s in range(1, 257):
        if R % s == 0 and R // s >= 32:
            best = s
            if s >= want:
                break
    return best


def _wgrad_hip(dtype, M, N, R):
    """True when the weight gradient of an [M,N] Linear over R rows goes to csrc/wgrad.hip.  Its 32x32 output
    tiles re-read g / x once per tile column / row: with many rows AND more than ~128 k outputs (c5's 768x256
    and 256x1024 at R = 12 560: 48 / 50 us vs 29 us) the library's split-K GEMM with its 64x64+ tiles is the
    faster one; at R <= 4096 the kernel is on par or ahead for the 1024x192 FFN shapes as well (9 vs 10.6 us at
    R = 800, 16.6 vs 17.9 at R = 2432) and saves the reduce launch."""
    return dtype == torch.bfloat16 and M % 2 == 0 and N % 2 == 0 and (M * N <= _WGRAD_HIP_MN[0] or R <= 4096)


import os as _os_wg
_WGRAD_HIP_MN = [131072]      # outputs up to which long batches stay on csrc/wgrad.hip


class _WgradBatch:
    """Collects the weight-gradient problems of one layer's backward and issues those that go to csrc/wgrad.hip as
    ONE grouped launch at the end of that backward (nothing downstream depends on a weight gradient).  Deferring
    further -- one launch for all layers when the autograd engine finishes the pass -- was tried and dropped: the
    engine's AccumulateGrad clones a gradient that anything else still references (here: the pending list), i.e.
    it would copy the buffers before they are filled."""

    def __init__(self):
        self.items = []

    def add(self, g, x, db=None, sink=None):
        """-> the dW tensor (filled when `flush` runs), or an immediately computed one for the library path.
        `sink`: a zero-initialised [M,N] f32 destination (the trainer's flat gradient slice) to accumulate into."""
        if not _wgrad_hip(g.dtype, g.shape[1], x.shape[1], g.shape[0]):
            return _wgrad(g, x, db, sink=sink)
        dw = sink[:] if sink is not None else ops.zeros_f32((g.shape[1], x.shape[1]), g.device)
        self.items.append((g, x, dw, db))
        return dw

    def flush(self, tail=None):
        """One launch for all queued weight gradients; `tail` = (a [M,K] bf16, w [K,N] bf16, c [M,N] f32): c += a @ w rides
        in the same launch (mobgt_layer_backward_tail).  Returns True if the tail was taken along."""
        it = self.items
        if not it:
            return False
        n = len(it)
        R = it[0][0].shape[0]
        vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        garr = (vp * n)(*[t[0].data_ptr() for t in it])
        xarr = (vp * n)(*[t[1].data_ptr() for t in it])
        warr = (vp * n)(*[t[2].data_ptr() for t in it])
        barr = (vp * n)(*[(t[3].data_ptr() if t[3] is not None else None) for t in it])
        ldg = (i64 * n)(*[t[0].stride(0) for t in it])
        ldx = (i64 * n)(*[t[1].stride(0) for t in it])
        ldw = (i64 * n)(*[t[2].shape[1] for t in it])
        M = (ci * n)(*[t[0].shape[1] for t in it])
        N = (ci * n)(*[t[1].shape[1] for t in it])
        self.items = []
        if tail is not None and R <= 1024:           # (the tail GEMM of mobgt_layer_backward_tail is sized for <= 1024 rows)
            a, w, c = tail
            if (a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and c.dtype == torch.float32 and a.shape[1] % 32 == 0
                    and w.shape[1] % 8 == 0 and a.is_contiguous() and w.is_contiguous() and c.is_contiguous()):
                check(_lib.lib().mobgt_layer_backward_tail(n, garr, ldg, xarr, ldx, warr, ldw, barr, R, M, N, _DT[it[0][0].dtype],
                                                           _p(a), a.stride(0), _p(w), w.stride(0), _p(c), c.stride(0),
                                                           a.shape[0], w.shape[1], a.shape[1], _stream()),
                      "mobgt_layer_backward_tail")
                return True
        check(_lib.lib().mobgt_linear_wgrad_group(n, garr, ldg, xarr, ldx, warr, ldw, barr, R, M, N, _DT[it[0][0].dtype],
                                                  _stream()), "mobgt_linear_wgrad_group")
        return False


def _wgrad(g, x, db=None, sink=None):
    """Weight gradient g^T @ x (fp32) of a Linear layer; bf16 operands go to the split-K MFMA kernel
    (csrc/wgrad.hip), which also accumulates the bias gradient g.sum(0) into `db` when given.  `sink` (library path): the
    gradient's destination in the trainer's flat buffer -- the sum over the split-K partial products may be parked for it."""
    if _wgrad_hip(g.dtype, g.shape[1], x.shape[1], g.shape[0]):
        return ops.linear_wgrad(g, x, db=db)[0]
    if db is not None:
        check(_lib.lib().mobgt_colsum(_p(g), _p(db), g.shape[0], g.shape[1], _DT[g.dtype], _stream()), "mobgt_colsum")
    return _mm_tn_f32(g, x, sink=sink)


_BMM_OUT_DTYPE = [None]


def _mm_tn_f32(g, x, sink=None):
    """g^T @ x for row-major g [R,M], x [R,N] with an fp32 result (library path: fp32 operands).
    The output is small and K = R is long, so K is split over a batch axis (strided views, no copies):
    a plain GEMM call maps a 192x192 output onto ONE workgroup and leaves the other 255 CUs idle."""
    R, M = g.shape
    N = x.shape[1]
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    # split only where a plain GEMM call is starved: <= 12 output tiles, or a very long K
    s = _split_factor(R, tiles) if (tiles <= 12 or (R >= 4096 and tiles < 256)) else 1
    if s > 1:
        ga, xa = g.view(s, R // s, M).transpose(1, 2), x.view(s, R // s, N)
        if g.dtype != torch.float32 and _BMM_OUT_DTYPE[0] is None:
            try:
                torch.bmm(ga[:1], xa[:1], out_dtype=torch.float32)
                _BMM_OUT_DTYPE[0] = True
            except Exception:
                _BMM_OUT_DTYPE[0] = False
        if g.dtype != torch.float32 and _BMM_OUT_DTYPE[0]:
            part = torch.bmm(ga, xa, out_dtype=torch.float32)           # f32 partial sums: no cast launch, no bf16 rounding
            # round 4: inside a train step the sum waits for the end of the backward pass, where all of the step's run as ONE
            # launch into the gradients' sinks (36 `.sum(0)` launches of 5.5 us per S-BIG step)
            parked = ops.defer_partial_sum(part, sink) if sink is not None else None
            return parked if parked is not None else part.sum(0)
        part = torch.bmm(ga, xa)
        return part.float().sum(0) if part.dtype != torch.float32 else part.sum(0)
    if g.dtype == torch.float32:
        return g.t() @ x
    if _MM_OUT_DTYPE[0] is None:
        try:
            torch.mm(g.t(), x, out_dtype=torch.float32)
            _MM_OUT_DTYPE[0] = True
        except Exception:
            _MM_OUT_DTYPE[0] = False
    if _MM_OUT_DTYPE[0]:
        return torch.mm(g.t(), x, out_dtype=torch.float32)      # bf16 operands, fp32 accumulate AND fp32 result
    return (g.t() @ x).float()


_ADDMM_OUT_DTYPE = [None]
import os as _os
_os_ln = _os
_TAIL = [True]
# (round 4 measured and round 5 removed: the 16-row chain kernels also past 4 096 rows -- S-BIG 9.53 -> 10.93 ms: a 16-row
#  workgroup re-streams the layer's weights from L2 785 times and loses against the library's tiles there)
_OWN_GEMM = [True]            # (False: the layer's GEMMs through torch -- tests flip it)


def _addmm_f32(c, a, b, inplace=False):
    """c (fp32) + a @ b (bf16 operands) with an fp32 result.  `inplace`: c is a scratch buffer of the caller and
    receives the result (beta = 1 GEMM straight onto it; out-of-place, addmm first copies c into a new tensor)."""
    if _ADDMM_OUT_DTYPE[0] is None:
        try:
            torch.addmm(c, a, b, out_dtype=torch.float32)
            _ADDMM_OUT_DTYPE[0] = True
        except Exception:
            _ADDMM_OUT_DTYPE[0] = False
    if _ADDMM_OUT_DTYPE[0]:
        if inplace:
            return torch.ops.aten.addmm.dtype_out(c, a, b, torch.float32, out=c)
        return torch.addmm(c, a, b, out_dtype=torch.float32)
    return c + (a @ b).float()


def _k1_fwd(x, y, x1, w, b, z, z32, mean, rstd, R, C, p, seed, seed_dev, salt, act):
    check(_lib.lib().mobgt_dropout_add_ln_fwd(_p(x), _p(y), _p(x1), _p(w), _p(b), _p(z), _p(z32), _p(mean), _p(rstd), R, C,
                                              p, seed, _p(seed_dev), salt, act, _stream()), "mobgt_dropout_add_ln_fwd")


_LN_GEMM = [True]                                               # False: the two-launch form (tests)
# Measured on the S-FSQ step (608 rows, C = 192), kernel durations inside the replayed graph:
#   forward   dropout_add_ln 4.8 us + FFN-1 GEMM 5.2 us  ->  fused 9.1 us
#   backward  dropout_add_ln' 5.7 us + GEMM 5.7 us       ->  fused 12.9 - 14.4 us (the first column tile of every row block
#             also reduces dgamma / dbeta / dbias and is the last workgroup to finish)
# i.e. the gaps between dependent launches of a replayed graph are ~0.1 us; what a stage costs is its own ramp-up + one
# memory round trip, and fusing two stages saves only the intermediate's round trip.  The forward fusion is on by
# default, the two backward ones only with MOBGT_LN_GEMM_BWD=1 (kept: parity-tested, and the better trade at other sizes).
_LN_GEMM_BWD = [_os_ln.environ.get("MOBGT_LN_GEMM_BWD") == "1"]
# csrc/chain.hip: out-proj -> LN -> FFN -> LN -> the next layer's QKV in one launch (MOBGT_NO_CHAIN=1: the separate launches)
_CHAIN = [_os_ln.environ.get("MOBGT_NO_CHAIN") != "1"]
_CHAIN_BIG = [_os_ln.environ.get("MOBGT_NO_CHAIN_BIG") != "1"]      # ... past 4 096 rows: the 64-row forward chain (else the library's GEMMs)
_CHAIN_BWD = [True]
_WGRAD_BIG = [True]      # past 4 096 rows: the layer's weight gradients as one launch (csrc/wgradbig.hip)      # ... and the same chain backwards (d(out) -> d(attention out))


# ---- what a layer's backward may leave to the backward of the layer BELOW it --------------------------------------------------
# After its attention backward a layer still owes (a) its input gradient dx = dx1 + dqkv Wqkv and (b) four weight gradients: one
# launch of ~9 us that keeps ~530 workgroups busy for a moment.  The next thing on the device is the lower layer's chain_bwd
# launch: 38 workgroups on 38 of 256 compute units for 16 us.  So, when the layer below is a fused layer that will run
# chain_bwd (its forward produced this layer's qkv), this layer returns dx1 as its input gradient, parks (dqkv, Wqkv^T, the
# four problems) under that tensor's address, and the lower layer's chain launch finishes dx per row block in front of its
# first norm and runs the weight gradients as extra workgroups (csrc/chain.hip).  A parked entry that no layer picks up is
# an error, raised when the backward pass ends -- never a silently incomplete gradient.
_DEFER = [_os_ln.environ.get("MOBGT_NO_DEFER_TAIL") != "1"]
_DEFER_MAX_R = [4096]      # S-GOW: 1024 / 2048 / 4096 / 16384 -> 18.96 / 19.33 / 19.65 / 19.5 k check-ins/s
_PENDING_TAIL = {}          # (graph task id, device index, address of dx1) -> parked work; see _pending_key
_PENDING_CB = [None]        # graph task id for which the end-of-backward check is queued


def _task_id():
    """Id of the autograd graph task (one per top-level backward / autograd.grad call) this code runs under, -1 outside."""
    fn = getattr(torch._C, "_current_graph_task_id", None)
    return int(fn()) if fn is not None else -1


def _pending_key(t):
    return (_task_id(), t.device.index, t.data_ptr())


def _drop_stale_pending():
    """Entries parked by a backward pass that is not the running one: that pass died before its end-of-backward callback
    ran (an exception in some backward function -- the engine then skips the callbacks).  Their buffers belong to a graph
    that no longer exists: drop them, never feed them to a kernel.  (An entry holds views of its buffers, so while it is
    parked no other tensor can be allocated at its address: a key of the RUNNING task always names the tensor it was
    parked under.)"""
    tid = _task_id()
    if _PENDING_CB[0] is not None and _PENDING_CB[0] != tid:
        _PENDING_CB[0] = None
    for k in [k for k in _PENDING_TAIL if k[0] != tid]:
        del _PENDING_TAIL[k]


def _complete_pending(pend):
    """The parked work as its own launch (what the layer would have issued itself): dx1 += dqkv Wqkv and the four dW."""
    if pend.get("preln"):
        # (a pre-LN layer's tail goes back through a norm whose parameters and statistics only its host holds)
        raise RuntimeError("mobgt fused layer (pre-LN): the parked input gradient of a chained layer found no host "
                           "(was the layer's input used by something else as well?); set MOBGT_NO_DEFER_TAIL=1")
    if pend.get("big"):                                   # (past 4 096 rows: the library's GEMM, onto dx1 in place)
        _addmm_f32(pend["dx1"], pend["dqkv"], pend["wqkv"], inplace=True)
        return
    wb = _WgradBatch()
    wb.items = pend["items"]
    if not wb.flush(tail=(pend["dqkv"], pend["wqkv"], pend["dx1"])):
        ops.layer_gemm(pend["dqkv"], pend["wqkv"], None, True, ops.GEMM_ADD, aux_in=pend["dx1"])


def _pending_check():
    """End of a backward pass (autograd engine callback): nothing may still be parked.  If the lower layer's output had a
    second consumer whose gradient reached the engine's input buffer FIRST, the buffer's sum is a new tensor, no layer
    recognises it and the entry is still here: the weight gradients are completed, the input gradient cannot be repaired
    (the sum was taken without dqkv Wqkv) -- hence the error, never a silent result.  (Had dx1 arrived first, the engine
    accumulates in place into it and the hosted product lands on top: correct.)"""
    tid = _task_id()
    _PENDING_CB[0] = None
    mine = [k for k in _PENDING_TAIL if k[0] == tid or tid == -1]
    if mine:
        left = [_PENDING_TAIL.pop(k) for k in mine]
        try:
            for pend in left:                    # finish the arithmetic, then say that the protocol was broken
                _complete_pending(pend)
        finally:
            _PENDING_TAIL.clear()
        raise RuntimeError("mobgt fused layer: a deferred input gradient was not consumed by the layer below "
                           "(was the layer's input used by something else as well?); set MOBGT_NO_DEFER_TAIL=1")


_CHAIN_WS = {}              # device index -> workspace of the chain kernels' cluster form


def chain_workspace(dev, C=192, R=None):
    """Hand-over counters + exchange buffers of the cluster form of the chain kernels (include/mobgt_hip.h:
    mobgt_chain_ws_bytes): zeroed once, for ONE stream at a time (the step's compute stream).  One per device AND per launch
    geometry (model width C, cluster size 4 / 2 -- csrc/chain.hip: pick_ncl): a row block's exchange area is addressed as
    block x members x 16 x C, so launches of different geometry on one workspace would poll words another block wrote, and
    a stale packet whose generation happens to match would be accepted.  Must exist before a graph capture starts
    (an eager warm-up step creates it).  ops.SAFE_FORMS -> None (one-workgroup form)."""
    if ops.SAFE_FORMS[0]:
        return None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    nblk = (int(R) + 15) // 16 if R is not None else 1
    ncl = 4 if nblk * 4 <= 256 else 2                     # (as pick_ncl; beyond 2 members the launch takes the one-workgroup form)
    key = (idx, int(C), ncl)
    ws = _CHAIN_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the chain kernels' workspace must be allocated before graph capture: run one eager step first")
        ws = _CHAIN_WS[key] = torch.zeros(int(_lib.lib().mobgt_chain_ws_bytes()), dtype=torch.uint8, device=dev)
    return ws


def _chain_ok(C, F, *ts):
    return (_CHAIN[0] and (C, F) in ((128, 1024), (192, 1024), (256, 1024))
            and all(t is None or (t.is_contiguous() and t.data_ptr() % 16 == 0) for t in ts))


def _ln_gemm_ok(C, *ts):
    """csrc/lngemm.hip: bf16 activations, model width a multiple of 32 up to 256, contiguous operands."""
    return (_LN_GEMM[0] and C % 32 == 0 and C <= 256
            and all(t is None or (t.is_contiguous() and t.data_ptr() % 16 == 0) for t in ts))


def _ln_gemm_fwd(x, y, x1, w, b, z, mean, rstd, R, C, p, seed, seed_dev, salt, weight, bias, epilogue):
    """x1 = x + dropout(y); z = LayerNorm(x1); out = z @ weight.T + bias [-> GELU] in ONE launch (mobgt_ln_gemm_fwd)."""
    N = weight.shape[0]
    out = torch.empty(R, N, dtype=torch.bfloat16, device=x.device)
    aux = torch.empty(R, N, dtype=torch.bfloat16, device=x.device) if epilogue == ops.GEMM_GELU else None
    check(_lib.lib().mobgt_ln_gemm_fwd(_p(x), _p(y), _p(x1), _p(w), _p(b), _p(z), _p(mean), _p(rstd), R, C, p, seed,
                                       _p(seed_dev), salt, _p(weight), weight.stride(0), _p(bias), _p(out), N, epilogue,
                                       _p(aux), N, _stream()), "mobgt_ln_gemm_fwd")
    return (out, aux) if aux is not None else out


def _ln_gemm_bwd(dz, dz32, dres, x1, mean, rstd, w, dx1, dy, dgamma, dbeta, dbias, R, C, p, seed, seed_dev, salt, weight_kn,
                 epilogue, aux_in):
    """dx1 = dres + LayerNorm'(dz + dz32); dy = dropout'(dx1); out = dy @ weight_kn [* gelu'(aux_in)] in ONE launch."""
    N = weight_kn.shape[1]
    out = torch.empty(R, N, dtype=torch.bfloat16, device=x1.device)
    check(_lib.lib().mobgt_ln_gemm_bwd(_p(dz), _p(dz32), _p(dres), _p(x1), _p(mean), _p(rstd), _p(w), _p(dx1), _p(dy),
                                       _p(dgamma), _p(dbeta), _p(dbias), R, C, p, seed, _p(seed_dev), salt, _p(weight_kn),
                                       weight_kn.stride(0), _p(out), N, epilogue, _p(aux_in), N, _stream()),
          "mobgt_ln_gemm_bwd")
    return out


def _k1_bwd(dz, dz32, dres, x1, mean, rstd, w, dx1, dy, dgamma, dbeta, dbias, R, C, p, seed, seed_dev, salt, act):
    check(_lib.lib().mobgt_dropout_add_ln_bwd(_p(dz), _p(dz32), _p(dres), _p(x1), _p(mean), _p(rstd), _p(w), _p(dx1), _p(dy),
                                              _p(dgamma), _p(dbeta), _p(dbias), R, C, p, seed, _p(seed_dev), salt, act,
                                              _stream()), "mobgt_dropout_add_ln_bwd")


class LayerConfig:
    """Per-call constants of one fused layer (not tensors that need gradients)."""

    def __init__(self, variant, num_heads, scale, p_drop, p_att, seed, seed_dev, salt, pack, act_dtype):
        self.variant, self.H, self.scale = variant, num_heads, scale
        self.p, self.p_att = float(p_drop), float(p_att)
        self.seed, self.seed_dev, self.salt = int(seed), seed_dev, int(salt) & 0xFFFFFFFF
        self.pack = pack
        self.act_dtype = act_dtype
        self.from_layer = False   # the layer's input is the output of another fused layer (model.fused_layer_forward)
        self.next_qkv = None      # (packed wqkv, bqkv) of the NEXT fused layer: its QKV projection rides in this layer's chain
        self.packed = None        # (wo, w1, w2) of this layer in MFMA operand order (model.pack_layer_weights)
        self.packed_t = None      # (w2^T, w1^T, wo^T, wqkv^T) likewise, for the backward chain
        self.out_act = self.out_qkv = None
        self.next_norm = None     # pre-LN: (weight, bias) of the NEXT layer's self_attention_norm, applied by this layer's chain
        self.out_preln = False    # ... and this layer's output then carries that layer's normed input + qkv


class _FusedLayerFn(torch.autograd.Function):
    """params (fp32 masters, reference names):
       fq   : wq, bq, wk, bk, wv, bv, wo, bo, n1 = ffn_norm1, nx = ffn_norm2, w1, b1, w2, b2
       stock: wq, bq, wk, bk, wv, bv, wo, bo, n1 = ffn_norm,  nx = self_attention_norm, w1, b1, w2, b2
    `shadows`: (wqkv [3C,C], bqkv [3C], wo, bo, w1, b1, w2, b2) in act_dtype (the fused masters when fp32)."""

    @staticmethod
    def forward(ctx, x, token, cfg, shadows, xa_pre, qkv_pre, wq, bq, wk, bk, wv, bv, wo, bo, n1w, n1b, nxw, nxb, w1, b1, w2, b2,
                nnw=None, nnb=None):
        # wq/wk/wv (+ biases) are views of one fused [3C, C] storage (MultiHeadAttention.fuse_qkv_storage); they
        # are separate arguments only so that autograd has an edge to each reference-named parameter.
        G, T, C = x.shape
        R = G * T
        A = cfg.act_dtype
        act = _DT[A]
        dev = x.device
        s_wqkv, s_bqkv, s_wo, s_bo, s_w1, s_b1, s_w2, s_b2 = shadows
        x = x.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        stats = torch.empty(6, R, **f32)
        seed, sd, salt = cfg.seed, cfg.seed_dev, cfg.salt
        stock = cfg.variant == "stock"
        # pre-LN, input normed and projected by the chain launch of the layer below (model.fused_layer_forward: no own norm)
        chained_in = stock and nxw is None
        if chained_in:
            if not (xa_pre is not None and qkv_pre is not None and xa_pre.dtype == A and xa_pre.numel() == R * C
                    and qkv_pre.dtype == A and qkv_pre.numel() == 3 * R * C):
                raise RuntimeError("mobgt fused layer (pre-LN): a chained input without its normed copy / qkv")
            xa = xa_pre.view(R, C)
        elif stock:                                                   # y = self_attention_norm(x)  (model.py:480)
            xa = torch.empty(R, C, dtype=A, device=dev)
            # (the first layer of a stack: norm + QKV projection as one launch when the own-GEMM path takes the layer -- below)
            norm_in_gemm = (A == torch.bfloat16 and qkv_pre is None and _OWN_GEMM[0] and ops.layer_gemm_ok(xa, s_wqkv)
                            and _ln_gemm_ok(C, x, s_wqkv, s_bqkv, nxw, nxb) and _os_ln.environ.get("MOBGT_NO_STOCK_LN_QKV") != "1")
            if not norm_in_gemm:
                _k1_fwd(x, None, None, nxw, nxb, xa, None, stats[0], stats[1], R, C, 0.0, seed, sd, salt, act)
        else:
            if A == torch.float32:
                xa = x.view(R, C)
            elif xa_pre is not None and xa_pre.dtype == A and xa_pre.numel() == R * C:
                xa = xa_pre.view(R, C)                             # written by the previous layer's LayerNorm kernel
            else:
                xa = x.view(R, C).to(A)
        # bf16 configuration at MobGT's sizes: the split-K MFMA kernel of csrc/gemm.hip (GELU fused after FFN layer 1);
        # otherwise the library
        own = _OWN_GEMM[0] and ops.layer_gemm_ok(xa, s_wqkv) and ops.layer_gemm_ok(xa, s_w1) and C % 32 == 0 \
            and s_w1.shape[0] % 32 == 0
        ctx.own_gemm = own
        if qkv_pre is not None and qkv_pre.dtype == A and qkv_pre.numel() == 3 * R * C and (not stock or chained_in):
            qkv = qkv_pre.view(G, T, 3 * C)                            # written by the previous layer's chain kernel
        elif stock and not chained_in and norm_in_gemm:
            # round 4: xa = self_attention_norm(x) and qkv = xa Wqkv^T + b in ONE launch (csrc/lngemm.hip, no residual input)
            qkv = _ln_gemm_fwd(x, None, None, nxw, nxb, xa, stats[0], stats[1], R, C, 0.0, seed, sd, salt, s_wqkv, s_bqkv,
                               ops.GEMM_BIAS).view(G, T, 3 * C)
        elif own:
            qkv = ops.layer_gemm(xa, s_wqkv, s_bqkv).view(G, T, 3 * C)
        else:
            qkv = torch.addmm(s_bqkv, xa, s_wqkv.t()).view(G, T, 3 * C)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        a, lse = ops._attn_fwd(q, k, v, cfg.pack, cfg.scale, cfg.p_att, seed ^ (salt * 0x9E3779B1), sd)
        F = s_w1.shape[0]
        cfg.out_qkv = None
        # (past 4 096 rows `own` is off -- the library's GEMMs take the layer's products -- but the forward chain has its 64-row
        #  form there: csrc/chain.hip, layer_chain_fwd_big_kernel; the backward then runs the library's launches)
        use_chain = (not stock and (own or (R > 4096 and _CHAIN_BIG[0])) and A == torch.bfloat16 and cfg.packed is not None
                     and _chain_ok(C, F, x, a, s_bo, s_b1, s_b2, n1w, n1b, nxw, nxb, *cfg.packed))
        # pre-LN chain (round 4): out-projection ... second residual add in one launch, + the NEXT layer's self_attention_norm
        # and QKV projection when the model named that layer (cfg.next_norm / next_qkv); the backward is the chain's too
        need_bwd = any(ctx.needs_input_grad)
        bwd_ok = bool(_CHAIN_BWD[0] and cfg.packed_t is not None and len(cfg.packed_t) > 3
                      and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in cfg.packed_t))
        stock_chain = (stock and own and A == torch.bfloat16 and cfg.packed is not None and (bwd_ok or not need_bwd)
                       and _wgrad_hip(A, F, C, R) and _chain_ok(C, F, x, a, s_bo, s_b1, s_b2, n1w, n1b, *cfg.packed))
        ctx.stock_chain = stock_chain
        ctx.chained_in = chained_in
        if chained_in and not stock_chain:
            raise RuntimeError("mobgt fused layer (pre-LN): the layer below chained into this one, which cannot run its chain kernels")
        ctx.fuse_ln = use_chain and own
        # (past 4 096 rows both chains are the 64-row kernels of csrc/chain.hip; the backward hosts the upper layer's tail only)
        ctx.chain_bwd = bool(use_chain and _CHAIN_BWD[0] and cfg.packed_t is not None and (R <= 4096 or _CHAIN_BIG[0])
                             and all(t.is_contiguous() and t.data_ptr() % 16 == 0 for t in cfg.packed_t))
        # the layer below produced this layer's qkv in ITS chain launch and will run chain_bwd: it can host what this layer's
        # backward leaves undone (see _PENDING_TAIL)
        ctx.below_hosts = bool(cfg.from_layer and qkv_pre is not None and ctx.chain_bwd and len(cfg.packed_t) > 3)
        if use_chain:               # everything row-local of the layer (+ the next layer's QKV projection) in one launch
            bf = dict(dtype=A, device=dev)
            x1, x2, out = torch.empty(R, C, **f32), torch.empty(R, C, **f32), torch.empty(R, C, **f32)
            z, out_a = torch.empty(R, C, **bf), torch.empty(R, C, **bf)
            u, h = torch.empty(R, F, **bf), torch.empty(R, F, **bf)
            nq = cfg.next_qkv
            if nq is not None and not (nq[0].dtype == A and tuple(nq[0].shape) == (3 * C, C) and nq[0].is_contiguous()
                                       and nq[0].data_ptr() % 16 == 0):
                nq = None
            qkv_next = torch.empty(R, 3 * C, **bf) if nq is not None else None
            p_wo, p_w1, p_w2 = cfg.packed
            check(_lib.lib().mobgt_layer_chain_fwd(_p(a), _p(x), _p(p_wo), _p(s_bo), _p(n1w), _p(n1b), _p(p_w1), _p(s_b1), _p(p_w2),
                                                   _p(s_b2), _p(nxw), _p(nxb), _p(nq[0] if nq else None),
                                                   _p(nq[1] if nq else None), _p(x1), _p(z), _p(u), _p(h), _p(x2), _p(out),
                                                   _p(out_a), _p(qkv_next), _p(stats[2]), _p(stats[3]), _p(stats[4]),
                                                   _p(stats[5]), R, C, F, cfg.p, seed, _p(sd), (salt + 1) & 0xFFFFFFFF,
                                                   (salt + 2) & 0xFFFFFFFF, _p(chain_workspace(dev, C, R)), _stream()), "mobgt_layer_chain_fwd")
            cfg.out_act, cfg.out_qkv = out_a, qkv_next
        elif stock_chain:
            bf = dict(dtype=A, device=dev)
            x1, x2 = torch.empty(R, C, **f32), torch.empty(R, C, **f32)
            z = torch.empty(R, C, **bf)
            u, h = torch.empty(R, F, **bf), torch.empty(R, F, **bf)
            nq = cfg.next_qkv if nnw is not None else None
            if nq is not None and not (nq[0].dtype == A and tuple(nq[0].shape) == (3 * C, C) and nq[0].is_contiguous()
                                       and nq[0].data_ptr() % 16 == 0 and nnw.is_contiguous() and nnb is not None):
                nq = None
            ctx.chain_next = nq is not None
            out_a = torch.empty(R, C, **bf) if nq is not None else None
            qkv_next = torch.empty(R, 3 * C, **bf) if nq is not None else None
            p_wo, p_w1, p_w2 = cfg.packed
            check(_lib.lib().mobgt_layer_chain_fwd(_p(a), _p(x), _p(p_wo), _p(s_bo), _p(n1w), _p(n1b), _p(p_w1), _p(s_b1), _p(p_w2),
                                                   _p(s_b2), _p(nnw if nq is not None else None), _p(nnb if nq is not None else None),
                                                   _p(nq[0] if nq else None), _p(nq[1] if nq else None), _p(x1), _p(z), _p(u), _p(h),
                                                   _p(x2), _p(None), _p(out_a), _p(qkv_next), _p(stats[2]), _p(stats[3]), _p(stats[4]),
                                                   _p(stats[5]), R, C, F, cfg.p, seed, _p(sd), (salt + 1) & 0xFFFFFFFF,
                                                   (salt + 2) & 0xFFFFFFFF, _p(chain_workspace(dev, C, R)), _stream()), "mobgt_layer_chain_fwd")
            out = x2                                                  # the residual stream passes the next layer's norm by
            if nq is not None:
                cfg.out_act, cfg.out_qkv, cfg.out_preln = out_a, qkv_next, True
        else:
            out, x1, z, u, h, x2 = _FusedLayerFn._tail_launches(ctx, cfg, x, a, stats, shadows, own, stock, n1w, n1b, nxw, nxb,
                                                                R, C, A, act, dev)
        # destinations for the four weight gradients in the trainer's flat buffer, if it registered any (q/k/v are
        # adjacent there exactly when they are adjacent in the fused [3C, C] parameter storage)
        sq, sk, sv = ops.grad_sink(wq), ops.grad_sink(wk), ops.grad_sink(wv)
        s_qkv = None
        if sq is not None and sk is not None and sv is not None and sq.is_contiguous() \
                and sk.data_ptr() == sq.data_ptr() + 4 * sq.numel() and sv.data_ptr() == sk.data_ptr() + 4 * sk.numel():
            s_qkv = torch.as_strided(sq, (3 * C, C), (C, 1))
        ctx.sinks = (s_qkv, ops.grad_sink(wo), ops.grad_sink(w1), ops.grad_sink(w2))
        # the block of small gradients [dbq dbk dbv | dbo | db1 | db2 | dn1w | dn1b | dnxw | dnxb]: one slice of the flat
        # buffer when the trainer laid these parameters out in that order (train.flat_order)
        ctx.small_sink = None
        # (pre-LN chain: the two norm gradients at the end of the block belong to ANOTHER layer when this layer's chain applies its
        #  successor's norm, and do not exist when the layer below applied this layer's: the block is the first eight, the norms'
        #  gradients have destinations of their own)
        chain = [ops.grad_sink(t) for t in ((bq, bk, bv, bo, b1, b2, n1w, n1b) if stock_chain else (bq, bk, bv, bo, b1, b2, n1w, n1b, nxw, nxb))]
        if all(t is not None and t.is_contiguous() for t in chain) and \
                all(chain[j + 1].data_ptr() == chain[j].data_ptr() + 4 * chain[j].numel() for j in range(len(chain) - 1)):
            ctx.small_sink = torch.as_strided(chain[0], (sum(t.numel() for t in chain),), (1,))
        if stock_chain:
            ctx.nx_sinks = (ops.grad_sink(nxw), ops.grad_sink(nxb)) if nxw is not None else (None, None)
            ctx.nn_sinks = (ops.grad_sink(nnw), ops.grad_sink(nnb)) if getattr(ctx, "chain_next", False) else (None, None)
        ctx.cfg = cfg
        ctx.shapes = (G, T, C)
        ctx.save_for_backward(x, xa, qkv, a, lse, x1, z, u, h, x2, stats, s_wqkv, s_wo, s_w1, s_w2, n1w, nxw,
                              nnw if getattr(ctx, "chain_next", False) else None)
        return out.view(G, T, C)

    @staticmethod
    def _tail_launches(ctx, cfg, x, a, stats, shadows, own, stock, n1w, n1b, nxw, nxb, R, C, A, act, dev):
        """out-proj ... second LayerNorm as separate launches (every configuration the chain kernel does not cover)."""
        s_wqkv, s_bqkv, s_wo, s_bo, s_w1, s_b1, s_w2, s_b2 = shadows
        seed, sd, salt = cfg.seed, cfg.seed_dev, cfg.salt
        f32 = dict(dtype=torch.float32, device=dev)
        y = ops.layer_gemm(a.view(R, C), s_wo, s_bo) if own else torch.addmm(s_bo, a.view(R, C), s_wo.t())
        x1 = torch.empty(R, C, **f32)
        z = torch.empty(R, C, dtype=A, device=dev)
        fuse_ln = own and A == torch.bfloat16 and _ln_gemm_ok(C, x, y, s_w1, s_b1, s_w2, s_wo)
        ctx.fuse_ln = fuse_ln
        if fuse_ln:             # x1 = x + dropout(y), z = LN(x1), u = z W1^T + b1, h = gelu(u): one launch
            u, h = _ln_gemm_fwd(x, y, x1, n1w, n1b, z, stats[2], stats[3], R, C, cfg.p, seed, sd, salt + 1, s_w1, s_b1,
                                ops.GEMM_GELU)
            f = ops.layer_gemm(h, s_w2, s_b2)
        elif own:
            _k1_fwd(x, y, x1, n1w, n1b, z, None, stats[2], stats[3], R, C, cfg.p, seed, sd, salt + 1, act)
            u, h = ops.layer_gemm(z, s_w1, s_b1, epilogue=ops.GEMM_GELU)
            f = ops.layer_gemm(h, s_w2, s_b2)
        else:
            _k1_fwd(x, y, x1, n1w, n1b, z, None, stats[2], stats[3], R, C, cfg.p, seed, sd, salt + 1, act)
            u = torch.addmm(s_b1, z, s_w1.t())
            h = torch.empty_like(u)
            check(_lib.lib().mobgt_gelu_fwd(_p(u), _p(h), u.numel(), act, _stream()), "mobgt_gelu_fwd")
            f = torch.addmm(s_b2, h, s_w2.t())
        x2 = torch.empty(R, C, **f32)
        if stock:                                                     # x = x + dropout(ffn(...))  (model.py:485-488)
            _k1_fwd(x1, f, x2, None, None, None, None, None, None, R, C, cfg.p, seed, sd, salt + 2, act)
            out = x2
        else:                                                         # ... then ffn_norm2 (model_fqandtoyo.py:1742)
            out = torch.empty(R, C, **f32)
            out_a = torch.empty(R, C, dtype=A, device=dev) if A != torch.float32 else None
            _k1_fwd(x1, f, x2, nxw, nxb, out_a, out, stats[4], stats[5], R, C, cfg.p, seed, sd, salt + 2, act)
            cfg.out_act = out_a                                       # bf16 copy for the next layer's QKV GEMM
        return out, x1, z, u, h, x2

    @staticmethod
    def _bwd_launches(ctx, cfg, wb, dout, x1, z, u, h, x2, a, stats, s_wo, s_w1, s_w2, n1w, nxw, dbo, db1, db2, dn1w, dn1b, dnxw,
                      dnxb, G, T, C, F, A, act, dev, own, stock, db1_in_wgrad):
        """d(out) ... d(attention out) as separate launches (every configuration the backward chain does not cover)."""
        R = G * T
        seed, sd, salt = cfg.seed, cfg.seed_dev, cfg.salt
        k_qkv, k_wo, k_w1, k_w2 = ctx.sinks
        df = torch.empty(R, C, dtype=A, device=dev)
        fuse_ln = ctx.fuse_ln and _LN_GEMM_BWD[0]
        du = None
        if stock:
            dx2 = dout                                                # grad at x2 = x1 + dropout(f)
            _k1_bwd(None, None, dout, x2, None, None, None, None, df, None, None, db2, R, C, cfg.p, seed, sd, salt + 2, act)
        else:
            dx2 = torch.empty(R, C, dtype=torch.float32, device=dev)  # through ffn_norm2
            if fuse_ln and db1_in_wgrad:    # ffn_norm2' + dropout' -> df, du = (df W2) * gelu'(u): one launch
                du = _ln_gemm_bwd(None, dout, None, x2, stats[4], stats[5], nxw, dx2, df, dnxw, dnxb, db2, R, C, cfg.p,
                                  seed, sd, salt + 2, s_w2, ops.GEMM_GELU_BWD, u)
            else:
                _k1_bwd(None, dout, None, x2, stats[4], stats[5], nxw, dx2, df, dnxw, dnxb, db2, R, C, cfg.p, seed, sd,
                        salt + 2, act)
        dw2 = wb.add(df, h, sink=k_w2)
        if du is not None:
            pass
        elif own and db1_in_wgrad:
            du = ops.layer_gemm(df, s_w2, None, True, ops.GEMM_GELU_BWD, aux_in=u)      # (df W2) * gelu'(u), one launch
        else:
            dh = df @ s_w2
            du = torch.empty_like(u)
            check(_lib.lib().mobgt_gelu_bwd_colsum(_p(dh), _p(u), _p(du), _p(None if db1_in_wgrad else db1), R, F, act,
                                                   _stream()), "mobgt_gelu_bwd_colsum")
        dz = ops.layer_gemm(du, s_w1, None, True) if own else du @ s_w1
        dw1 = wb.add(du, z, db=db1 if db1_in_wgrad else None, sink=k_w1)
        dx1 = torch.empty(R, C, dtype=torch.float32, device=dev)
        dy = torch.empty(R, C, dtype=A, device=dev)
        if fuse_ln:             # ffn_norm1' + residual + dropout' -> dy, da = dy Wo: one launch
            da = _ln_gemm_bwd(dz, None, dx2, x1, stats[2], stats[3], n1w, dx1, dy, dn1w, dn1b, dbo, R, C, cfg.p, seed, sd,
                              salt + 1, s_wo, ops.GEMM_BIAS, None).view(G, T, C)
        else:
            _k1_bwd(dz, None, dx2, x1, stats[2], stats[3], n1w, dx1, dy, dn1w, dn1b, dbo, R, C, cfg.p, seed, sd, salt + 1, act)
            da = (ops.layer_gemm(dy, s_wo, None, True) if own else dy @ s_wo).view(G, T, C)
        dwo = wb.add(dy, a.view(R, C), sink=k_wo)
        return da, dx1, dw2, dw1, dwo

    @staticmethod
    def backward(ctx, dout):
        cfg = ctx.cfg
        G, T, C = ctx.shapes
        R = G * T
        x, xa, qkv, a, lse, x1, z, u, h, x2, stats, s_wqkv, s_wo, s_w1, s_w2, n1w, nxw, nnw = ctx.saved_tensors
        pend = getattr(ctx, "_mobgt_pending", None)     # (model.note_pending_backward: the layer's shadows may be rewritten again)
        if pend is not None:
            pend.done()
        if getattr(ctx, "stock_chain", False):
            return _FusedLayerFn._backward_preln_chain(ctx, dout, x, xa, qkv, a, lse, x1, z, u, h, x2, stats, s_wqkv, n1w, nxw, nnw)
        A = cfg.act_dtype
        act = _DT[A]
        dev = dout.device
        seed, sd, salt = cfg.seed, cfg.seed_dev, cfg.salt
        stock = cfg.variant == "stock"
        F = s_w1.shape[0]
        dout = dout.contiguous().view(R, C).float()
        # every small gradient of the layer in ONE zero-filled buffer:
        # [dbqkv 3C | dbo C | db1 F | db2 C | dn1w C | dn1b C | dnxw C | dnxb C]
        small = ctx.small_sink if ctx.small_sink is not None else ops.zeros_f32((3 * C + C + F + C + 4 * C,), dev)
        o = [0]

        def take(n):
            t = small[o[0]:o[0] + n]
            o[0] += n
            return t
        dbqkv, dbo, db1, db2, dn1w, dn1b, dnxw, dnxb = take(3 * C), take(C), take(F), take(C), take(C), take(C), take(C), take(C)
        own = ctx.own_gemm
        wb = _WgradBatch()
        k_qkv, k_wo, k_w1, k_w2 = ctx.sinks                          # gradient sinks (or None)
        db1_in_wgrad = _wgrad_hip(A, F, C, R)                         # then b1's gradient rides on the dW1 kernel
        if _PENDING_TAIL or _PENDING_CB[0] is not None:
            _drop_stale_pending()
        pend = _PENDING_TAIL.pop(_pending_key(dout), None) if _PENDING_TAIL else None
        # (the 64-row backward chain, R > 4096: b1's gradient is a column sum of du behind it -- the weight gradients are the
        #  library's there)
        chain_b = getattr(ctx, "chain_bwd", False) and (db1_in_wgrad or R > 4096) and not stock
        host = pend is not None and chain_b and pend["R"] == R
        if pend is not None and not host:
            _complete_pending(pend)                                   # this layer cannot host it: finish it right here
            pend = None
        if chain_b:
            # d(out) -> ffn_norm2' -> dropout' -> (W2, gelu') -> W1 -> ffn_norm1' + residual -> dropout' -> Wo: one launch
            bf = dict(dtype=A, device=dev)
            df, dy, da = torch.empty(R, C, **bf), torch.empty(R, C, **bf), torch.empty(R, C, **bf)
            du = torch.empty(R, F, **bf)
            dx1 = torch.empty(R, C, dtype=torch.float32, device=dev)
            w2t, w1t, wot = cfg.packed_t[:3]
            vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
            if pend is not None:
                # ... and, hosted for the layer above: its input gradient is finished in front of the first norm (`dout` holds
                # its dx1), its four weight gradients run as extra workgroups of this launch
                it = pend["items"]
                n = len(it)
                extra = (_p(pend["dqkv"]), _p(pend["wqt"]), n, (vp * n)(*[t[0].data_ptr() for t in it]),
                         (i64 * n)(*[t[0].stride(0) for t in it]), (vp * n)(*[t[1].data_ptr() for t in it]),
                         (i64 * n)(*[t[1].stride(0) for t in it]), (vp * n)(*[t[2].data_ptr() for t in it]),
                         (i64 * n)(*[t[2].shape[1] for t in it]),
                         (vp * n)(*[(t[3].data_ptr() if t[3] is not None else None) for t in it]),
                         (ci * n)(*[t[0].shape[1] for t in it]), (ci * n)(*[t[1].shape[1] for t in it]))
            else:
                extra = (None, None, 0, None, None, None, None, None, None, None, None, None)
            if R > 4096 and (pend is None or not pend["items"]):
                # the 64-row form: b1's gradient is summed inside (db1 is zero-filled: `small`); it hosts the upper layer's tail
                # (dout = that layer's dx1; dout + dqkv Wqkv is finished in front of the first norm), not its weight gradients
                check(_lib.lib().mobgt_layer_chain_bwd_big(_p(dout), _p(x2), _p(x1), _p(u), _p(stats[2]), _p(stats[3]), _p(stats[4]),
                                                           _p(stats[5]), _p(n1w), _p(nxw), _p(w2t), _p(w1t), _p(wot), _p(df), _p(du),
                                                           _p(dy), _p(da), _p(dx1), _p(dnxw), _p(dnxb), _p(db2), _p(dn1w), _p(dn1b),
                                                           _p(dbo), _p(None if db1_in_wgrad else db1), R, C, F, cfg.p, seed, _p(sd),
                                                           (salt + 1) & 0xFFFFFFFF, (salt + 2) & 0xFFFFFFFF,
                                                           _p(pend["dqkv"]) if pend is not None else None,
                                                           _p(pend["wqt"]) if pend is not None else None, _stream()),
                      "mobgt_layer_chain_bwd_big")
            else:
                check(_lib.lib().mobgt_layer_chain_bwd(_p(dout), _p(x2), _p(x1), _p(u), _p(stats[2]), _p(stats[3]), _p(stats[4]),
                                                       _p(stats[5]), _p(n1w), _p(nxw), _p(w2t), _p(w1t), _p(wot), _p(df), _p(du),
                                                       _p(dy), _p(da), _p(dx1), _p(dnxw), _p(dnxb), _p(db2), _p(dn1w), _p(dn1b),
                                                       _p(dbo), R, C, F, cfg.p, seed, _p(sd), (salt + 1) & 0xFFFFFFFF,
                                                       (salt + 2) & 0xFFFFFFFF, *extra, _p(chain_workspace(dev, C, R)), _stream()), "mobgt_layer_chain_bwd")
                if not db1_in_wgrad:
                    check(_lib.lib().mobgt_colsum(_p(du), _p(db1), R, F, act, _stream()), "mobgt_colsum")
            da = da.view(G, T, C)
            big_items = None
            if R > 4096 and _WGRAD_BIG[0] and A == torch.bfloat16 and not db1_in_wgrad:
                # past 4 096 rows: all four weight gradients of the layer (and dbqkv) as ONE launch behind the attention backward
                # (csrc/wgradbig.hip, round 6; before: three library split-K GEMMs, the grouped kernel for dWo, a column-sum launch)
                big_items = [(df, h, None, k_w2), (du, z, None, k_w1), (dy, a.view(R, C), None, k_wo)]
                if not ops.layer_wgrad_big_ok(big_items):
                    big_items = None
            if big_items is None:
                dw2 = wb.add(df, h, sink=k_w2)
                dw1 = wb.add(du, z, db=db1 if db1_in_wgrad else None, sink=k_w1)
                dwo = wb.add(dy, a.view(R, C), sink=k_wo)
        else:
            big_items = None
            da, dx1, dw2, dw1, dwo = _FusedLayerFn._bwd_launches(ctx, cfg, wb, dout, x1, z, u, h, x2, a, stats, s_wo, s_w1, s_w2,
                                                                n1w, nxw, dbo, db1, db2, dn1w, dn1b, dnxw, dnxb, G, T, C, F, A,
                                                                act, dev, own, stock, db1_in_wgrad)
        dqkv = torch.empty(G, T, 3 * C, dtype=A, device=dev)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        ops._attn_bwd(q, k, v, a, lse, da, dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:], cfg.pack, cfg.scale,
                      cfg.p_att, seed ^ (salt * 0x9E3779B1), sd)
        dqkv2 = dqkv.view(R, 3 * C)
        if big_items is not None and ops.layer_wgrad_big_ok([(dqkv2, xa.view(R, C), dbqkv, k_qkv)]):
            dw2, dw1, dwo, dwqkv = ops.layer_wgrad_big(big_items + [(dqkv2, xa.view(R, C), dbqkv, k_qkv)], R)
        else:
            if big_items is not None:
                dw2, dw1, dwo = ops.layer_wgrad_big(big_items, R)
            dwqkv = wb.add(dqkv2, xa, db=dbqkv, sink=k_qkv)
        defer = (_DEFER[0] and getattr(ctx, "below_hosts", False) and own and not stock and _TAIL[0] and len(wb.items) == 4
                 and R <= _DEFER_MAX_R[0] and dx1.data_ptr() % 16 == 0 and dqkv2.is_contiguous() and ctx.small_sink is not None
                 and all(k is not None for k in ctx.sinks))
        big_defer = (not defer and _DEFER[0] and getattr(ctx, "below_hosts", False) and not stock and _TAIL[0] and R > 4096
                     and dx1.data_ptr() % 16 == 0 and dqkv2.is_contiguous() and dqkv2.data_ptr() % 16 == 0)
        if big_defer:
            # past 4 096 rows: the 64-row chain of the layer below finishes dx = dx1 + dqkv Wqkv in front of its first norm (a
            # 20 us library GEMM otherwise); the weight gradients are the library's and are issued here
            wb.flush(tail=None)
            _PENDING_TAIL[_pending_key(dx1)] = dict(dx1=dx1[:], dqkv=dqkv2, wqkv=s_wqkv, wqt=cfg.packed_t[3], items=[], R=R, big=True)
            if _PENDING_CB[0] != _task_id():
                _PENDING_CB[0] = _task_id()
                torch.autograd.Variable._execution_engine.queue_callback(_pending_check)
            rode = True
            dx = dx1
        elif defer:
            # nothing more is launched for this layer: the chain launch of the layer below finishes dx and the weight gradients
            # (parked as FRESH views of the gradient buffers: autograd's AccumulateGrad clones a returned gradient that anything
            # else still references, and the clone -- taken before the buffers are filled -- would later be copied over them)
            items = [(g_, x_, dw_[:], db_[:] if db_ is not None else None) for g_, x_, dw_, db_ in wb.items]
            _PENDING_TAIL[_pending_key(dx1)] = dict(dx1=dx1[:], dqkv=dqkv2, wqkv=s_wqkv, wqt=cfg.packed_t[3], items=items, R=R)
            wb.items = []
            if _PENDING_CB[0] != _task_id():
                _PENDING_CB[0] = _task_id()
                torch.autograd.Variable._execution_engine.queue_callback(_pending_check)
            rode = True
            dx = dx1
        elif (stock and ops._WGRAD_DEFER["on"] and wb.items and R <= _DEFER_MAX_R[0] and all(k is not None for k in ctx.sinks)
              and all(g_.dtype == torch.bfloat16 and x_.dtype == torch.bfloat16 for g_, x_, _, _ in wb.items)):
            # stock variant inside a train step: nothing downstream reads a weight gradient and all four land in their sinks, so
            # they join the step's ONE grouped launch (ops.flush_deferred_wgrads) instead of one 7 us launch per layer -- parked
            # as FRESH views (see above: AccumulateGrad clones a returned gradient that anything else still references)
            for g_, x_, dw_, db_ in wb.items:
                ops._WGRAD_DEFER["items"].append((g_, x_, None, None, (1.0, 1.0, 1.0), dw_[:], db_[:] if db_ is not None else None, False))
            wb.items = []
            rode = False
        else:
            # fq variant, own GEMMs: dx = dx1 + dqkv Wqkv rides in the weight-gradient launch
            rode = wb.flush(tail=(dqkv2, s_wqkv, dx1) if (own and not stock and _TAIL[0]) else None)
        if rode:
            dx = dx1
        elif stock:                                                     # back through self_attention_norm
            dz0 = ops.layer_gemm(dqkv2, s_wqkv, None, True) if own else dqkv2 @ s_wqkv
            dx = torch.empty(R, C, dtype=torch.float32, device=dev)
            _k1_bwd(dz0, None, dx1, x.view(R, C), stats[0], stats[1], nxw, dx, None, dnxw, dnxb, None, R, C, 0.0, seed, sd,
                    salt, act)
        elif A == torch.float32:                                      # dx1 is this function's own buffer: accumulate onto it
            dx = torch.addmm(dx1, dqkv2, s_wqkv, out=dx1)
        elif own:
            dx = ops.layer_gemm(dqkv2, s_wqkv, None, True, ops.GEMM_ADD, aux_in=dx1)     # dx1 + dqkv Wqkv, f32, in place
        else:
            dx = _addmm_f32(dx1, dqkv2, s_wqkv, inplace=True)
        return (dx.view(G, T, C), None, None, None, None, None, dwqkv[:C], dbqkv[:C], dwqkv[C:2 * C], dbqkv[C:2 * C], dwqkv[2 * C:],
                dbqkv[2 * C:], dwo, dbo, dn1w, dn1b, dnxw, dnxb, dw1, db1, dw2, db2) + ((None, None) if stock else ())

    @staticmethod
    def _backward_preln_chain(ctx, dout, x, xa, qkv, a, lse, x1, z, u, h, x2, stats, s_wqkv, n1w, nxw, nnw):
        """model.py:479-489 backwards through the chain kernel (mobgt_layer_chain_bwd_preln):
            [hosted for the layer above:  t = dqkv_above Wqkv_above;  dx2 = dout + norm_above'(t)]    else dx2 = dout
            df = dropout'(dx2);  du = (df W2) gelu'(u);  dz = du W1;  dx1 = dx2 + ffn_norm'(dz);  dy = dropout'(dx1);  da = dy Wo
        then the attention backward; then either this layer's own norm (a layer without a chained input: dx = dx1 +
        self_attention_norm'(dqkv Wqkv), two launches) or nothing -- the layer below finishes dx in ITS chain launch."""
        cfg = ctx.cfg
        G, T, C = ctx.shapes
        R = G * T
        A = cfg.act_dtype
        act = _DT[A]
        dev = dout.device
        seed, sd, salt = cfg.seed, cfg.seed_dev, cfg.salt
        F = u.shape[1]
        dout = dout.contiguous().view(R, C).float()
        n8 = 3 * C + C + F + C + 2 * C
        small = ctx.small_sink if ctx.small_sink is not None else ops.zeros_f32((n8,), dev)
        o = [0]

        def take(n):
            t = small[o[0]:o[0] + n]
            o[0] += n
            return t
        dbqkv, dbo, db1, db2, dn1w, dn1b = take(3 * C), take(C), take(F), take(C), take(C), take(C)

        def dst(sink):
            return sink[:] if sink is not None else ops.zeros_f32((C,), dev)
        dnxw, dnxb = (dst(ctx.nx_sinks[0]), dst(ctx.nx_sinks[1])) if nxw is not None else (None, None)
        dnnw, dnnb = (dst(ctx.nn_sinks[0]), dst(ctx.nn_sinks[1])) if nnw is not None else (None, None)
        wb = _WgradBatch()
        k_qkv, k_wo, k_w1, k_w2 = ctx.sinks
        if _PENDING_TAIL or _PENDING_CB[0] is not None:
            _drop_stale_pending()
        pend = _PENDING_TAIL.pop(_pending_key(dout), None) if _PENDING_TAIL else None
        if pend is not None and not (pend.get("preln") and nnw is not None and pend["R"] == R):
            _complete_pending(pend)                    # (not this layer's kind of guest: finish it as its own launches)
            pend = None
        bf = dict(dtype=A, device=dev)
        df, dy, da = torch.empty(R, C, **bf), torch.empty(R, C, **bf), torch.empty(R, C, **bf)
        du = torch.empty(R, F, **bf)
        dx1 = torch.empty(R, C, dtype=torch.float32, device=dev)
        w2t, w1t, wot = cfg.packed_t[:3]
        tail = (_p(pend["dqkv"]), _p(pend["wqt"])) if pend is not None else (None, None)
        check(_lib.lib().mobgt_layer_chain_bwd_preln(_p(dout), _p(x2), _p(x1), _p(u), _p(stats[2]), _p(stats[3]), _p(stats[4]),
                                                     _p(stats[5]), _p(n1w), _p(nnw), _p(w2t), _p(w1t), _p(wot), _p(df), _p(du),
                                                     _p(dy), _p(da), _p(dx1), _p(dnnw), _p(dnnb), _p(db2), _p(dn1w), _p(dn1b),
                                                     _p(dbo), R, C, F, cfg.p, seed, _p(sd), (salt + 1) & 0xFFFFFFFF,
                                                     (salt + 2) & 0xFFFFFFFF, *tail, 0, None, None, None, None, None, None, None,
                                                     None, None, _p(chain_workspace(dev, C, R)), _stream()), "mobgt_layer_chain_bwd_preln")
        dw2 = wb.add(df, h, sink=k_w2)
        dw1 = wb.add(du, z, db=db1, sink=k_w1)
        dwo = wb.add(dy, a.view(R, C), sink=k_wo)
        dqkv = torch.empty(G, T, 3 * C, dtype=A, device=dev)
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        ops._attn_bwd(q, k, v, a, lse, da.view(G, T, C), dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:], cfg.pack, cfg.scale,
                      cfg.p_att, seed ^ (salt * 0x9E3779B1), sd)
        dqkv2 = dqkv.view(R, 3 * C)
        dwqkv = wb.add(dqkv2, xa, db=dbqkv, sink=k_qkv)
        # the four weight gradients: inside a train step they join the step's ONE grouped launch (as the separate-launch stock
        # layer's do), else their own grouped launch now
        if (ops._WGRAD_DEFER["on"] and wb.items and R <= _DEFER_MAX_R[0] and all(k_ is not None for k_ in ctx.sinks)
                and all(g_.dtype == torch.bfloat16 and x_.dtype == torch.bfloat16 for g_, x_, _, _ in wb.items)):
            for g_, x_, dw_, db_ in wb.items:
                ops._WGRAD_DEFER["items"].append((g_, x_, None, None, (1.0, 1.0, 1.0), dw_[:], db_[:] if db_ is not None else None, False))
            wb.items = []
        else:
            wb.flush()
        if ctx.chained_in:
            # nothing more for this layer: the layer below opens its chain launch with  dx2 = dx1 + norm'(dqkv Wqkv)
            _PENDING_TAIL[_pending_key(dx1)] = dict(dx1=dx1[:], dqkv=dqkv2, wqkv=s_wqkv, wqt=cfg.packed_t[3], items=[], R=R, preln=True)
            if _PENDING_CB[0] != _task_id():
                _PENDING_CB[0] = _task_id()
                torch.autograd.Variable._execution_engine.queue_callback(_pending_check)
            dx = dx1
        else:                                                           # back through this layer's own self_attention_norm
            dz0 = ops.layer_gemm(dqkv2, s_wqkv, None, True)
            dx = torch.empty(R, C, dtype=torch.float32, device=dev)
            _k1_bwd(dz0, None, dx1, x.view(R, C), stats[0], stats[1], nxw, dx, None, dnxw, dnxb, None, R, C, 0.0, seed, sd, salt, act)
        return (dx.view(G, T, C), None, None, None, None, None, dwqkv[:C], dbqkv[:C], dwqkv[C:2 * C], dbqkv[C:2 * C], dwqkv[2 * C:],
                dbqkv[2 * C:], dwo, dbo, dn1w, dn1b, dnxw, dnxb, dw1, db1, dw2, db2, dnnw, dnnb)


def fused_encoder_layer(x, pack, cfg, shadows, params, xa_pre=None, qkv_pre=None):
    cfg.out_act = cfg.out_qkv = None
    cfg.out_preln = False
    out = _FusedLayerFn.apply(x, pack.token, cfg, shadows, xa_pre, qkv_pre, *params)
    if cfg.out_preln:
        out._mobgt_preln = True               # pre-LN: `_mobgt_act` is the NEXT layer's normed input (model.fused_layer_forward)
    if cfg.out_act is not None:
        out._mobgt_act = cfg.out_act          # picked up by the next fused layer (same Python tensor object)
    if cfg.out_qkv is not None:
        out._mobgt_qkv = cfg.out_qkv          # ... and its QKV projection, already computed by this layer's chain kernel
        out._mobgt_from_layer = True          # (the consumer's backward may leave its tail to this layer's: _PENDING_TAIL)
    return out

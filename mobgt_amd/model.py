"""Drop-in counterparts of `graphormer/model.py` (stock Graphormer) on MI355X.

Same constructor arguments, `forward` signatures and state_dict names as the reference
(`model.py:393-489` for FeedForwardNetwork / MultiHeadAttention / EncoderLayer, `model.py:23-217` for
Graphormer), so reference checkpoints load unchanged.  What differs is where the work runs:

* the score/bias/softmax/dropout/PV chain of `MultiHeadAttention.forward` (model.py:436-455) is ONE HIP
  kernel (`mobgt_attn_bias_fwd`) and its autograd is two more (`mobgt_attn_bias_bwd`);
* the bias assembly of `Graphormer.forward` (model.py:126-190: table gathers, permutes, slice-adds and
  the multi-hop edge reduce with its [G,N,N,D,H] temporaries) is `mobgt_build_bias`;
* node features (model.py:193-203) are one gather-sum kernel;
* the dense projections / FFN stay on hipBLASLt through torch (they are library GEMMs, not hot ops).

Lightning glue, FLAG, ogb/ZINC branches of the reference are out of scope.
"""
import math
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def init_bert_params(module, n_layers):
    """model.py:14-20"""
    if isinstance(module, nn.Linear):
        module.weight.data.normal_(mean=0.0, std=0.02 / math.sqrt(n_layers))
        if module.bias is not None:
            module.bias.data.zero_()
    if isinstance(module, nn.Embedding):
        module.weight.data.normal_(mean=0.0, std=0.02)


def no_grad_row0(weight):
    """nn.Embedding(padding_idx=0): row 0 is read like any other row but never receives a gradient."""
    return torch.cat([weight[:1].detach(), weight[1:]], dim=0)


def hop_table_from(edge_weight, edge_dis_weight, H, D, fp16_roundtrip=False):
    """[D, n_edge, H] table T[d,e,h] = sum_h' edge_encoder[e,h'] * edge_dis_encoder[d,h',h]
    (model.py:166-176: embedding -> bmm with `edge_dis_encoder.weight.reshape(-1,H,H)[:D]`).  Because the
    per-pair work is linear in the gathered rows, gathering from this product replaces the reference's
    gather + [G*N*N, H] x [H, H] bmm per hop.  Row 0 of the edge table is `padding_idx` (no gradient).
    fp16_roundtrip reproduces model_fqandtoyo.py:1178-1198: operands rounded to fp16, fp32 accumulate,
    product rounded to fp16 (exact for F == 1, which is every MobGT dataset)."""
    if edge_weight.is_cuda and edge_weight.dtype == torch.float32 and edge_dis_weight.dtype == torch.float32:
        from . import ops
        return ops.hop_table(edge_weight, edge_dis_weight, H, D, fp16_roundtrip)     # one launch each way
    W = edge_dis_weight.reshape(-1, H, H)[:D]
    enc = no_grad_row0(edge_weight)
    if fp16_roundtrip:
        return torch.matmul(enc.half().float().unsqueeze(0), W.half().float()).half().float()
    return torch.matmul(enc.unsqueeze(0), W)


class FeedForwardNetwork(nn.Module):
    """model.py:393-405"""

    def __init__(self, hidden_size, ffn_size, dropout_rate):
        super().__init__()
        self.layer1 = nn.Linear(hidden_size, ffn_size)
        self.gelu = nn.GELU()
        self.layer2 = nn.Linear(ffn_size, hidden_size)

    def forward(self, x):
        return self.layer2(self.gelu(self.layer1(x)))


_layer_counter = [0]


class MultiHeadAttention(nn.Module):
    """model.py:408-460.  `attn_bias` may be a torch tensor broadcastable to [G,H,T,T] (reference call
    convention) or an `ops.PackedBias` (what Graphormer.forward passes, already in kernel layout)."""

    def __init__(self, hidden_size, attention_dropout_rate, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.att_size = att_size = hidden_size // num_heads
        self.scale = att_size ** -0.5
        self.linear_q = nn.Linear(hidden_size, num_heads * att_size)
        self.linear_k = nn.Linear(hidden_size, num_heads * att_size)
        self.linear_v = nn.Linear(hidden_size, num_heads * att_size)
        self.att_dropout = nn.Dropout(attention_dropout_rate)
        self.output_layer = nn.Linear(num_heads * att_size, hidden_size)
        _layer_counter[0] += 1
        self.set_layer_index(_layer_counter[0])     # stand-alone layers: a process-wide count; models renumber 1..L
        self._pack_cache = None
        self.seed_dev = None            # optional device int64 scalar: advanced by the trainer each step

    def set_layer_index(self, index):
        """Dropout masks are a pure function of (seed, step, layer index, element): a model numbers its layers
        1..L so that two instances built the same way draw the same masks."""
        self._layer_index = int(index)
        self._seed_salt = (self._layer_index * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF

    def fuse_qkv_storage(self):
        """Make linear_q/k/v.weight (and .bias) views of one [3C, C] (and [3C]) tensor so that the QKV projection
        is a single GEMM without a per-step concatenation.  Parameter objects, names and values are unchanged
        (state_dict / optimizer are unaffected); re-run after .to(device) / load_state_dict re-allocations."""
        ws = [self.linear_q.weight, self.linear_k.weight, self.linear_v.weight]
        bs = [self.linear_q.bias, self.linear_k.bias, self.linear_v.bias]
        C = ws[0].shape[0]
        fw = getattr(self, "_wqkv", None)
        ok = fw is not None and fw.device == ws[0].device and all(
            w.data_ptr() == fw.data_ptr() + i * C * w.shape[1] * 4 for i, w in enumerate(ws)) and all(
            b.data_ptr() == self._bqkv.data_ptr() + i * C * 4 for i, b in enumerate(bs))
        if not ok:
            with torch.no_grad():
                fw = torch.cat([w.detach().float() for w in ws], dim=0).contiguous()
                fb = torch.cat([b.detach().float() for b in bs], dim=0).contiguous()
                for i, (w, b) in enumerate(zip(ws, bs)):
                    w.data = fw[i * C:(i + 1) * C]
                    b.data = fb[i * C:(i + 1) * C]
            self._wqkv, self._bqkv = fw, fb
        return self._wqkv, self._bqkv

    def _packed(self, attn_bias, G, T, ref):
        """The bias in kernel layout.  A dense tensor is packed once per TENSOR OBJECT (the L layers of a reference-style
        `for layer in layers: layer(x, bias)` loop share the pack): the cache is keyed on the object's identity through a weak
        reference plus its version counter -- never on its address, which the caching allocator hands to the next batch's
        bias as soon as this one is freed (model.py:190 creates a fresh tensor per batch)."""
        if isinstance(attn_bias, ops.PackedBias):
            return attn_bias
        if attn_bias is None:
            attn_bias = torch.zeros(1, 1, T, T, device=ref.device)
        cache = MultiHeadAttention._shared_cache
        src = cache.get("src")
        grad_mode = attn_bias.requires_grad and torch.is_grad_enabled()
        if (src is None or src() is not attn_bias or cache.get("key") != (attn_bias._version, G, T, self.num_heads, grad_mode)
                or cache["pack"].spent):
            cache["pack"] = ops.pack_bias(attn_bias, G, self.num_heads, T)
            cache["src"] = weakref.ref(attn_bias)
            cache["key"] = (attn_bias._version, G, T, self.num_heads, grad_mode)
        return cache["pack"]

    _shared_cache = {}

    def _masked_forward(self, q, k, v, attn_bias, mask):
        """model.py:446-448: `x.masked_fill(mask.unsqueeze(1), 0)` puts the SCORE (bias included) of a masked pair to 0 -- not to
        -inf -- in front of the softmax, so it cannot be folded into an additive bias.  No caller of the reference passes a mask
        (model.py:208, model_fqandtoyo.py:1350 `mask=None`); the branch is kept in its dense form, in fp32, op by op."""
        G, Tq = q.shape[0], q.shape[1]
        H, d = self.num_heads, self.att_size
        qh = self.linear_q(q).float().view(G, -1, H, d).transpose(1, 2) * self.scale
        kh = self.linear_k(k).float().view(G, -1, H, d).transpose(1, 2).transpose(2, 3)
        vh = self.linear_v(v).float().view(G, -1, H, d).transpose(1, 2)
        s = torch.matmul(qh, kh)
        if attn_bias is not None:
            if isinstance(attn_bias, ops.PackedBias):
                attn_bias = attn_bias.dense()
            s = s + attn_bias.float()
        s = s.masked_fill(mask.unsqueeze(1), 0)
        p = self.att_dropout(torch.softmax(s, dim=3))
        x = torch.matmul(p, vh).transpose(1, 2).contiguous().view(G, Tq, H * d)
        return self.output_layer(x.to(self.output_layer.weight.dtype))

    def forward(self, q, k, v, attn_bias=None, mask=None):
        if mask is not None:
            x = self._masked_forward(q, k, v, attn_bias, mask)
            assert x.size() == q.size()
            return x
        orig_q_size = q.size()
        G, T = q.shape[0], q.shape[1]
        pack = self._packed(attn_bias, G, T, q)
        p_drop = self.att_dropout.p if self.training else 0.0
        seed = self._seed_salt
        if p_drop > 0 and self.seed_dev is None:
            seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
        if q is k and k is v:
            w = torch.cat([self.linear_q.weight, self.linear_k.weight, self.linear_v.weight], dim=0)
            b = torch.cat([self.linear_q.bias, self.linear_k.bias, self.linear_v.bias], dim=0)
            x = ops.attention_qkv(F.linear(q, w, b), pack, self.scale, p_drop, seed, self.seed_dev)
        else:
            x = ops.attention(self.linear_q(q), self.linear_k(k), self.linear_v(v), pack, self.scale, p_drop, seed,
                              self.seed_dev)
        x = self.output_layer(x)
        assert x.size() == orig_q_size
        return x


def fused_layer_forward(layer, variant, x, attn_bias, n1, nx, next_layer=None):
    """Run one EncoderLayer as a single fused autograd node (mobgt_amd/fused_layer.py).  `n1` is the LayerNorm in
    front of the FFN, `nx` the variant's other LayerNorm (self_attention_norm for model.py, ffn_norm2 for fq).
    `next_layer`: the EncoderLayer the caller will run next ON THIS LAYER'S OUTPUT (fq variant): its QKV projection then
    rides in this layer's last launch."""
    from .fused_layer import LayerConfig, fused_encoder_layer
    mha = layer.self_attention
    G, T, C = x.shape
    pack = mha._packed(attn_bias, G, T, x)
    wqkv, bqkv = mha.fuse_qkv_storage()
    act = getattr(layer, "act_dtype", torch.float32)
    # AMP (the reference trains with --precision 16, README.md:62): under torch.autocast the layer takes its bf16
    # configuration -- bf16 GEMM operands / activations, fp32 residual stream, statistics and master weights -- whatever
    # `act_dtype` says; fp16 autocast maps to bf16 too (no fp16 instantiation exists; bf16 is the MFMA operand type).
    # The node itself then runs with autocast off: its dtypes are explicit.
    amp = x.is_cuda and torch.is_autocast_enabled("cuda")
    if amp:
        act = torch.bfloat16
    masters = (wqkv, bqkv, mha.output_layer.weight, mha.output_layer.bias, layer.ffn.layer1.weight, layer.ffn.layer1.bias,
               layer.ffn.layer2.weight, layer.ffn.layer2.bias)
    if act == torch.float32:
        shadows = tuple(m.detach() for m in masters)
    else:
        sh = getattr(layer, "_shadows", None)
        if sh is None or sh[0].device != x.device or sh[0].dtype != act:
            sh = tuple(torch.empty_like(m, dtype=act) for m in masters)
            layer._shadows = sh
            layer._shadow_fresh = False
            layer._shadow_ver = None
        if not getattr(layer, "_shadow_fresh", False) or act != getattr(layer, "act_dtype", torch.float32):
            # (stand-alone layers, and AMP on a layer configured for fp32: copy here; a model refreshes all layers in one call.
            #  Unchanged weights are not copied again -- refresh_shadows says why)
            ver = _weights_version(layer)
            if ((getattr(layer, "_shadow_ver", None) != ver or not _may_skip_shadow_copy(layer))
                    and not getattr(layer, "_shadow_external", False)):
                torch._foreach_copy_(list(sh), [m.detach() for m in masters])
                layer._shadow_ver = ver
                layer._packed_ver = None
        layer._shadow_fresh = False
        shadows = sh
    training = layer.training
    p = layer.self_attention_dropout.p if training else 0.0
    p_att = mha.att_dropout.p if training else 0.0
    seed = mha._seed_salt
    if (p > 0 or p_att > 0) and mha.seed_dev is None:
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    cfg = LayerConfig(variant, mha.num_heads, mha.scale, p, p_att, seed, mha.seed_dev, mha._layer_index * 8, pack, act)
    # (a layer the trainer cuts the backward pass in front of -- train.TrainStep's layer-wise gradient buckets -- finishes its own
    # input gradient: the layer below runs in another autograd pass and cannot host this one's tail)
    cfg.from_layer = bool(getattr(x, "_mobgt_from_layer", False)) and not getattr(layer, "_mobgt_cut", False)
    # the next layer's QKV projection rides in this layer's chain kernel (csrc/chain.hip) when that layer is fused, has
    # the same activation dtype and its bf16 shadows are current (refresh_shadows / the trainer refreshed ALL layers)
    nxt = next_layer
    stock = variant == "stock"
    if (nxt is not None and act != torch.float32 and not amp and getattr(nxt, "fused", False)
            and getattr(nxt, "act_dtype", None) == act and getattr(nxt, "_packed_fresh", False)
            and nxt._packed[0].device == x.device):
        if not stock:
            cfg.next_qkv = (nxt._packed[0], nxt._shadows[1])
        else:
            # pre-LN (model.py:479-489): the next layer's self_attention_norm AND its QKV projection ride in this layer's chain
            # launch; that layer then has no norm / projection of its own and leaves the norm's backward to this layer's chain
            # (fused_layer._PENDING_TAIL) -- so not across a cut of the trainer's backward pass, and not without deferral
            from . import fused_layer as _fl
            bwd_ok = (not torch.is_grad_enabled()) or (getattr(nxt, "_packed_t_fresh", False) and getattr(layer, "_packed_t_fresh", False))
            if _fl._DEFER[0] and not getattr(nxt, "_mobgt_cut", False) and getattr(layer, "_packed_fresh", False) and bwd_ok:
                cfg.next_qkv = (nxt._packed[0], nxt._shadows[1])
                cfg.next_norm = (nxt.self_attention_norm.weight, nxt.self_attention_norm.bias)
    if getattr(layer, "_packed_fresh", False) is True and act != torch.float32 and not amp:
        cfg.packed = layer._packed[1:]                    # (wo, w1, w2) in MFMA operand order
        if getattr(layer, "_packed_t_fresh", False):
            cfg.packed_t = layer._packed_t                # (w2^T, w1^T, wo^T)
    # like `_shadow_fresh` above: a pack is good for the ONE forward that follows the refresh (refresh_shadows /
    # pack_layer_weights run once per model forward); a later stand-alone call of this layer re-copies its shadows and
    # must not pair them with the old pack
    layer._packed_fresh = layer._packed_t_fresh = False
    # a pre-LN layer whose input was normed and projected by the chain launch of the layer below has no edge to its own
    # self_attention_norm: that layer's node owns it (next_norm above)
    chained_in = (stock and not amp and getattr(x, "_mobgt_preln", False) and getattr(x, "_mobgt_qkv", None) is not None
                  and getattr(x, "_mobgt_act", None) is not None and not getattr(layer, "_mobgt_cut", False))
    params = (mha.linear_q.weight, mha.linear_q.bias, mha.linear_k.weight, mha.linear_k.bias, mha.linear_v.weight,
              mha.linear_v.bias, mha.output_layer.weight, mha.output_layer.bias, n1.weight, n1.bias,
              None if chained_in else nx.weight, None if chained_in else nx.bias,
              layer.ffn.layer1.weight, layer.ffn.layer1.bias, layer.ffn.layer2.weight, layer.ffn.layer2.bias)
    if stock:
        nn_ = getattr(cfg, "next_norm", None)
        params = params + ((nn_[0], nn_[1]) if nn_ is not None else (None, None))
    if amp:
        with torch.autocast("cuda", enabled=False):
            out = fused_encoder_layer(x.float(), pack, cfg, shadows, params, xa_pre=getattr(x, "_mobgt_act", None))
    else:
        out = fused_encoder_layer(x.float(), pack, cfg, shadows, params, xa_pre=getattr(x, "_mobgt_act", None),
                                  qkv_pre=getattr(x, "_mobgt_qkv", None))
    if act != torch.float32 and out.requires_grad:
        note_pending_backward(layer, out)          # (its backward reads the bf16 shadows / packs saved above)
    return out


_PENDING_PACK = []          # jobs of a deferred weight pack (see pack_layer_weights(defer=True))


def take_pending_pack():
    """The deferred pack jobs, handed to the launch that will carry them (modelGNN._SmallGcnFn); the list is emptied."""
    jobs = list(_PENDING_PACK)
    del _PENDING_PACK[:]
    return jobs


def flush_pending_pack():
    """Launch a deferred pack that nobody picked up (call in front of the first consumer of `layer._packed`)."""
    jobs = take_pending_pack()
    if jobs:
        _launch_pack(jobs)


def _launch_pack(jobs):
    import ctypes
    from . import _lib
    from .ops import _stream
    for o in range(0, len(jobs), 96):
        part = jobs[o:o + 96]
        n = len(part)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        _lib.check(_lib.lib().mobgt_pack_mfma_b(n, (vp * n)(*[j[0].data_ptr() for j in part]), (vp * n)(*[j[1].data_ptr() for j in part]),
                                                (ci * n)(*[j[2] for j in part]), (ci * n)(*[j[3] for j in part]),
                                                (ci * n)(*[j[4] for j in part]), _stream()), "mobgt_pack_mfma_b")


class _PendingBackward:
    """One forward pass of a fused layer whose backward has not run yet: lives on the autograd node (`out.grad_fn`), so it goes
    away with the graph; `done()` is called by the layer's backward."""
    __slots__ = ("pending", "__weakref__")

    def __init__(self, pending):
        self.pending = pending
        pending.add(id(self))
        weakref.finalize(self, pending.discard, id(self))

    def done(self):
        self.pending.discard(id(self))


def note_pending_backward(layer, out):
    """`out` of a fused layer forward that saved the layer's bf16 shadows / packs for its backward."""
    fn = getattr(out, "grad_fn", None)
    if fn is None:
        return
    pend = layer.__dict__.setdefault("_pending_bwd", set())
    try:
        fn._mobgt_pending = _PendingBackward(pend)
    except Exception:                 # (a node that takes no Python attributes: the version check alone then decides)
        pass


def _may_skip_shadow_copy(layer):
    """Re-deriving the bf16 shadows is skipped on an unchanged version counter ONLY where rewriting them is the wrong thing to do:
    while a backward pass that saved them is pending (an eval forward between a training forward and its backward: the copy would
    trip autograd's version check for nothing), or when the owner declared the weights frozen (`layer._weights_frozen = True`: an
    evaluation loop that does not want 26 MB copied per batch).  Everywhere else the copy is unconditional: a version
    counter does not see `p.data.add_()` / `p.data.copy_()` (the model's own init_params, many third-party optimizers), raw-pointer
    writers or an optimizer step replayed inside a captured graph, and skipping there would silently train / evaluate on stale
    bf16 weights while the fp32 masters move."""
    return bool(getattr(layer, "_pending_bwd", None)) or bool(getattr(layer, "_weights_frozen", False))


def _weights_version(layer):
    """What the bf16 shadows / MFMA-order packs of a layer were derived from: (version counter, address) of each of its GEMM
    parameters.  In-place updates through the parameter (optimizers, load_state_dict, copy_) bump the counter; writes through
    `.data` or raw pointers do not -- which is why an unchanged version only spares the copy under _may_skip_shadow_copy."""
    mha = layer.self_attention
    ps = (mha.linear_q.weight, mha.linear_k.weight, mha.linear_v.weight, mha.linear_q.bias, mha.linear_k.bias, mha.linear_v.bias,
          mha.output_layer.weight, mha.output_layer.bias, layer.ffn.layer1.weight, layer.ffn.layer1.bias, layer.ffn.layer2.weight,
          layer.ffn.layer2.bias)
    return tuple((p._version, p.data_ptr()) for p in ps)


def refresh_shadows(layers, defer_pack=False, rows=None):
    """bf16 copies of every layer's GEMM weights in ONE multi-tensor copy (call once per forward).  `defer_pack`: the MFMA-order
    pack is not launched but left for the category GCN's forward launch to carry (take_pending_pack) -- the caller MUST call
    flush_pending_pack() in front of the first consumer of the packs.
    A layer whose weights have not changed since its shadows were made is left alone WHILE A BACKWARD PASS THAT SAVED THEM IS
    PENDING (an eval forward between a training forward and its backward: the copy would rewrite saved tensors and autograd's
    version check fires -- loudly, but for nothing) or when its owner set `layer._weights_frozen` (evaluation loops);
    otherwise the copy is unconditional -- see _may_skip_shadow_copy."""
    dst, src = [], []
    for layer in layers:
        if getattr(layer, "act_dtype", torch.float32) == torch.float32 or not layer.fused:
            continue
        if getattr(layer, "_shadow_external", False):       # kept current by the trainer's optimizer kernel
            layer._shadow_fresh = True
            continue
        mha = layer.self_attention
        wqkv, bqkv = mha.fuse_qkv_storage()
        masters = (wqkv, bqkv, mha.output_layer.weight, mha.output_layer.bias, layer.ffn.layer1.weight, layer.ffn.layer1.bias,
                   layer.ffn.layer2.weight, layer.ffn.layer2.bias)
        sh = getattr(layer, "_shadows", None)
        ver = _weights_version(layer)
        if sh is None or sh[0].device != wqkv.device or sh[0].dtype != layer.act_dtype:
            sh = tuple(torch.empty_like(m, dtype=layer.act_dtype) for m in masters)
            layer._shadows = sh
            layer._shadow_ver = None
        if getattr(layer, "_shadow_ver", None) != ver or not _may_skip_shadow_copy(layer):
            dst += list(sh)
            src += [m.detach() for m in masters]
            layer._shadow_ver = ver
            layer._packed_ver = None
        layer._shadow_fresh = True
    if dst:
        torch._foreach_copy_(dst, src)
    pack_layer_weights(layers, defer=defer_pack, rows=rows)


def pack_layer_weights(layers, defer=False, rows=None):
    """MFMA-operand-order copies of the fused layers' (fq post-LN and, since round 4, model.py's pre-LN) bf16 GEMM weights (csrc/chain.hip reads a wave's B operand as one
    contiguous KB), all layers in one launch (up to 96 weights): `layer._packed` = (wqkv, wo, w1, w2) packed for the
    forward chain and, when gradients are enabled, `layer._packed_t` = (w2^T, w1^T, wo^T, wqkv^T) for the backward chain."""
    from . import fused_layer
    flush_pending_pack()                                     # (a pack deferred earlier and never picked up)
    if not fused_layer._CHAIN[0]:
        return
    want_t = torch.is_grad_enabled() and fused_layer._CHAIN_BWD[0]
    # `rows`: the batch's token rows, when the caller knows them.  Past 4 096 rows the FORWARD runs the 64-row chain kernel
    # (csrc/chain.hip: layer_chain_fwd_big_kernel / layer_chain_bwd_big_kernel, round 5): the same packs as below them.
    # MOBGT_NO_CHAIN_BIG=1: the library's GEMMs forward as well -- nothing reads a pack then but the token-assembly launch, which
    # multiplies by the FIRST layer's QKV weight.
    big = rows is not None and rows > 4096
    qkv0_only = big and not fused_layer._CHAIN_BIG[0]
    if big and not fused_layer._CHAIN_BIG[0]:
        want_t = False
    jobs = []
    for li, layer in enumerate(layers):
        layer._packed_fresh = layer._packed_t_fresh = False
        sh = getattr(layer, "_shadows", None)
        if (sh is None or not getattr(layer, "fused", False) or not getattr(layer, "_shadow_fresh", False)
                or not (hasattr(layer, "ffn_norm2") or hasattr(layer, "self_attention_norm")) or sh[0].dtype != torch.bfloat16
                or not sh[0].is_cuda):
            continue
        C, F = sh[2].shape[0], sh[4].shape[0]
        if (C, F) not in ((128, 1024), (192, 1024), (256, 1024)) or not all(sh[i].is_contiguous() for i in (0, 2, 4, 6)):
            continue
        pk = getattr(layer, "_packed", None)
        if pk is None or pk[0].device != sh[0].device:
            pk = tuple(torch.empty_like(sh[i]) for i in (0, 2, 4, 6))
            layer._packed = pk
            layer._packed_ver = None
        want_t_l = bool(want_t and any(p.requires_grad for p in layer.parameters()))
        pt = getattr(layer, "_packed_t", None)
        if want_t_l and (pt is None or pt[0].device != sh[0].device):
            pt = tuple(torch.empty_like(sh[i]) for i in (6, 4, 2, 0))
            layer._packed_t = pt
            layer._packed_ver = None
        # packs made from these very shadows before (weights unchanged since: refresh_shadows) are reused, not rewritten -- a
        # pending backward pass may have saved them.  Shadows a trainer's optimizer kernel rewrites every step carry no version.
        ver = None if getattr(layer, "_shadow_external", False) else getattr(layer, "_shadow_ver", None)
        have = getattr(layer, "_packed_ver", None)
        what = "qkv" if qkv0_only else "all"
        reuse = ver is not None and have is not None and have[0] == ver and have[1] == what and (have[2] or not want_t_l)
        if qkv0_only:
            if li == 0:
                if not reuse:
                    jobs.append((sh[0], pk[0], sh[0].shape[0], sh[0].shape[1], 0))
                    layer._packed_ver = (ver, what, False)
                layer._packed_fresh = "qkv"                 # (model.fused_layer_forward hands on a pack only when this is True)
            continue
        if not reuse:
            for d, i in zip(pk, (0, 2, 4, 6)):
                jobs.append((sh[i], d, sh[i].shape[0], sh[i].shape[1], 0))
            if want_t_l:
                for d, i in zip(pt, (6, 4, 2, 0)):       # dX = dY W: W [K = out, N = in] is the operand, packed as [N][K]
                    jobs.append((sh[i], d, sh[i].shape[1], sh[i].shape[0], 1))
            layer._packed_ver = (ver, what, want_t_l)
        layer._packed_fresh = True
        layer._packed_t_fresh = want_t_l
    import os
    # (as a passenger only while the pack is short next to the 26 us host launch: 6.5 M elements at S-FSQ = 11.7 us alone.  At
    #  S-BIG -- 18.9 M -- the passengers outlasted the network: 102 us for 28 + 37, measured)
    small = sum(j[2] * j[3] for j in jobs) <= (8 << 20)
    if defer and jobs and len(jobs) <= 96 and small and os.environ.get("MOBGT_NO_PACK_PASSENGER") != "1":
        _PENDING_PACK.extend(jobs)
    else:
        _launch_pack(jobs)


def sync_external_shadows(model):
    """bf16 shadow weights owned by a trainer (train.TrainStep keeps them current from inside its optimizer kernel) go
    stale when something else writes the parameters; writers call this to have them re-derived."""
    for layer in getattr(model, "layers", []):          # (shadows / packs a layer made itself: derived again at its next forward)
        layer._shadow_ver = layer._packed_ver = None
    ref = model.__dict__.get("_shadow_sync")
    fn = ref() if ref is not None else None
    if fn is not None:
        fn()


class EncoderLayer(nn.Module):
    """model.py:463-489 (pre-LN)."""
    fused = True                 # one fused autograd node per layer on the GPU; False = op-by-op torch + HIP attention
    act_dtype = torch.float32    # dtype of the GEMM-facing activations in the fused path (fp32 or bf16)

    def __init__(self, hidden_size, ffn_size, dropout_rate, attention_dropout_rate, num_heads):
        super().__init__()
        self.self_attention_norm = nn.LayerNorm(hidden_size)
        self.self_attention = MultiHeadAttention(hidden_size, attention_dropout_rate, num_heads)
        self.self_attention_dropout = nn.Dropout(dropout_rate)
        self.ffn_norm = nn.LayerNorm(hidden_size)
        self.ffn = FeedForwardNetwork(hidden_size, ffn_size, dropout_rate)
        self.ffn_dropout = nn.Dropout(dropout_rate)

    def forward(self, x, attn_bias=None, mask=None, next_layer=None):
        if self.fused and x.is_cuda and mask is None:
            return fused_layer_forward(self, "stock", x, attn_bias, self.ffn_norm, self.self_attention_norm, next_layer=next_layer)
        y = self.self_attention_norm(x)
        y = self.self_attention(y, y, y, attn_bias, mask=mask)
        y = self.self_attention_dropout(y)
        x = x + y.to(x.dtype)          # same-dtype add: the mixed fp32+bf16 elementwise kernel is ~20x slower on ROCm
        y = self.ffn_norm(x)
        y = self.ffn(y)
        y = self.ffn_dropout(y)
        return x + y.to(x.dtype)


class Graphormer(nn.Module):
    """model.py:23-217 without the Lightning / ogb / FLAG branches.  `num_class` replaces the
    `get_dataset(dataset_name)["num_class"]` lookup (model.py:84-86)."""

    def __init__(self, n_layers, num_heads, hidden_dim, dropout_rate, intput_dropout_rate, weight_decay, ffn_dim,
                 dataset_name, warmup_updates, tot_updates, peak_lr, end_lr, edge_type, multi_hop_max_dist,
                 attention_dropout_rate, num_class=1, bias_dtype=torch.float32, act_dtype=torch.float32, fused_layers=True,
                 num_atoms=512 * 9 + 1, **_unused):
        """`num_atoms`: rows of `atom_encoder` (model.py:44 hard-codes 512 * 9 + 1, the OGB atom vocabulary; a POI universe
        larger than that -- bench.py --variant stock -- needs one row per POI id)."""
        super().__init__()
        self.num_heads = num_heads
        self.atom_encoder = nn.Embedding(num_atoms, hidden_dim, padding_idx=0)
        self.edge_encoder = nn.Embedding(512 * 3 + 1, num_heads, padding_idx=0)
        self.edge_type = edge_type
        if self.edge_type == "multi_hop":
            self.edge_dis_encoder = nn.Embedding(128 * num_heads * num_heads, 1)
        self.rel_pos_encoder = nn.Embedding(512, num_heads, padding_idx=0)
        self.in_degree_encoder = nn.Embedding(512, hidden_dim, padding_idx=0)
        self.out_degree_encoder = nn.Embedding(512, hidden_dim, padding_idx=0)
        self.input_dropout = nn.Dropout(intput_dropout_rate)
        self.layers = nn.ModuleList([EncoderLayer(hidden_dim, ffn_dim, dropout_rate, attention_dropout_rate, num_heads)
                                     for _ in range(n_layers)])
        for li, layer in enumerate(self.layers):
            layer.act_dtype, layer.fused = act_dtype, fused_layers
            layer.self_attention.set_layer_index(li + 1)
        self.final_ln = nn.LayerNorm(hidden_dim)
        self.downstream_out_proj = nn.Linear(hidden_dim, num_class)
        self.graph_token = nn.Embedding(1, hidden_dim)
        self.graph_token_virtual_distance = nn.Embedding(1, num_heads)
        self.dataset_name = dataset_name
        self.warmup_updates, self.tot_updates = warmup_updates, tot_updates
        self.peak_lr, self.end_lr, self.weight_decay = peak_lr, end_lr, weight_decay
        self.multi_hop_max_dist = multi_hop_max_dist
        self.hidden_dim = hidden_dim
        self.bias_dtype = bias_dtype
        self.apply(lambda module: init_bert_params(module, n_layers=n_layers))

    def _hop_depth(self, batched_data):
        D = batched_data.edge_input.shape[3]
        if self.multi_hop_max_dist > 0:
            D = min(D, self.multi_hop_max_dist)
        if self.edge_type != "multi_hop":
            raise NotImplementedError("only edge_type='multi_hop' is used by MobGT (README.md:62)")
        return D

    def assemble_bias(self, batched_data, hop=None):
        """model.py:126-190 -> ops.PackedBias (`hop`: the hop table when the caller already has it)"""
        H = self.num_heads
        edge_input = batched_data.edge_input
        D = self._hop_depth(batched_data)
        if hop is None:
            hop = hop_table_from(self.edge_encoder.weight, self.edge_dis_encoder.weight, H, D)
        rel = self.rel_pos_encoder.weight      # padding_idx = 0: build_bias_bwd never adds into row 0
        return ops.build_bias(batched_data.attn_bias, batched_data.rel_pos, None, edge_input, rel, None, hop,
                              self.graph_token_virtual_distance.weight, D, dtype=self.bias_dtype)

    def validate_batch(self, batched_data):
        """Index ranges nn.Embedding / F.cross_entropy check in the reference (model.py:193-203, 286: IndexError / a device
        assert there; the gather, token and loss kernels here treat an out-of-range index as "contributes nothing", which would
        train silently on masked inputs -- ADVICE r3).  One reduction + one host read per batch OBJECT, remembered on it: a
        pre-collated batch is checked in the eager dry run and costs nothing inside a captured step."""
        if getattr(batched_data, "_mobgt_validated", None) is self or not batched_data.x.is_cuda:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        lim = [("x", batched_data.x, self.atom_encoder.num_embeddings), ("in_degree", batched_data.in_degree, self.in_degree_encoder.num_embeddings),
               ("rel_pos", batched_data.rel_pos, self.rel_pos_encoder.num_embeddings),
               ("edge_input", batched_data.edge_input, self.edge_encoder.num_embeddings),
               ("y", batched_data.y, self.downstream_out_proj.out_features)]
        mx = torch.stack([t.max().long() if t.numel() else torch.zeros((), dtype=torch.long, device=t.device) for _, t, _ in lim]).tolist()
        mn = int(batched_data.y.min()) if batched_data.y.numel() else 0
        for (name, _, n), m in zip(lim, mx):
            if m >= n:
                raise IndexError(f"batch.{name} has index {m}, out of range for a table of {n} rows")
        if mn < 0:
            raise IndexError(f"batch.y has the negative class {mn}")
        try:
            batched_data._mobgt_validated = self
        except AttributeError:
            pass

    def index_limits(self):
        """Largest admissible value + 1 of the raw fields train.EpochLoop checks on the host for a fresh batch."""
        return dict(x=self.atom_encoder.num_embeddings - 1, y=self.downstream_out_proj.out_features - 1,
                    edge=self.edge_encoder.num_embeddings, deg=self.in_degree_encoder.num_embeddings)

    def forward(self, batched_data, perturb=None):
        self.validate_batch(batched_data)
        x = batched_data.x
        in_degree = out_degree = batched_data.in_degree            # model.py:118 (aliasing kept)
        n_graph = x.size(0)
        if x.shape[2] != 1:
            raise NotImplementedError("MobGT items have one feature column (wrapper.py:37)")
        # (one feature column: a view, not a strided copy)
        xi = x.reshape(x.shape[0], x.shape[1])
        if xi.dtype not in (torch.int64, torch.int32):
            xi = xi.long()
        tabs = (self.atom_encoder.weight, self.in_degree_encoder.weight, self.out_degree_encoder.weight)
        deg_raw = in_degree.reshape(xi.shape)
        one_launch = (ops.stock_tokens_ok(xi, *tabs, self.graph_token.weight)
                      and deg_raw.dtype in (torch.int64, torch.int32, torch.int16))
        if one_launch and os.environ.get("MOBGT_NO_STOCK_FRONT") != "1":
            # round 4: the hop table's forward and the layers' weight pack are left as jobs and ride in the launch of the encoder
            # input (three independent front launches as one grid); the bias build follows it
            ops.front_deferral(True)
            try:
                hop = hop_table_from(self.edge_encoder.weight, self.edge_dis_encoder.weight, self.num_heads, self._hop_depth(batched_data))
                refresh_shadows(self.layers, defer_pack=True)
                output = ops.stock_tokens(xi, deg_raw, deg_raw, *tabs, self.graph_token.weight, self.input_dropout.p, self.training, 0x1003)
            finally:
                ops.front_deferral(False)
                ops.flush_front()
                flush_pending_pack()
            bias = self.assemble_bias(batched_data, hop=hop)
            one_launch = None                                  # (done)
        else:
            bias = self.assemble_bias(batched_data)
            refresh_shadows(self.layers)
        if one_launch is None:
            pass
        elif one_launch:
            # gather + graph token + input dropout: one launch each way (same values, same mask as the three ops below); the
            # degrees go in in their own dtype
            output = ops.stock_tokens(xi, deg_raw, deg_raw, *tabs, self.graph_token.weight, self.input_dropout.p, self.training, 0x1003)
        else:
            deg = deg_raw if deg_raw.dtype == xi.dtype else deg_raw.to(xi.dtype)    # (the aliased degree tensor is widened ONCE)
            node_feature = ops.embed_gather_sum(list(tabs), [xi, deg, deg], padding_idx=[0, 0, 0])
            graph_token_feature = self.graph_token.weight.unsqueeze(0).expand(n_graph, -1, -1)      # (cat reads it strided: no copy)
            output = ops.dropout(torch.cat([graph_token_feature, node_feature], dim=1), self.input_dropout.p, self.training, 0x1003)
        self._bias_pack, self._cuts = bias, {}
        for li, enc_layer in enumerate(self.layers):
            if getattr(enc_layer, "_mobgt_cut", False):
                self._cuts[li] = output          # train.TrainStep: the backward pass is cut here (layer-wise gradient buckets)
            # (the layer that follows is named so that its norm and QKV projection can ride in this layer's chain launch)
            output = enc_layer(output, bias, mask=None, next_layer=self.layers[li + 1] if li + 1 < len(self.layers) else None)
        self._enc_out = output
        # (model.py:211-217 normalises every token and then reads the graph token: LayerNorm is per row, so only that row is
        # normalised here -- same value, same gradient)
        if ops.token_layer_norm_ok(output, self.final_ln.weight):
            tok = ops.token_layer_norm(output, self.final_ln.weight, self.final_ln.bias, self.final_ln.eps)    # one launch each way
        else:
            tok = self.final_ln(output[:, 0, :])
        proj = self.downstream_out_proj
        if ops.skinny_linear_ok(tok, proj.weight):
            # G <= 16 rows against thousands of classes: one pass over the weight per product (csrc/skinny.hip; the library's
            # M = 16 GEMM took 29 us of the S-FSQ step at K = 128), and the weight gradient lands in its sink
            return ops.skinny_linear(tok, proj.weight, proj.bias)
        return proj(tok)

    head_modules = ("final_ln", "downstream_out_proj")

    def training_step(self, batched_data, batch_idx=0):
        """model.py:218-285, the generic branch (`loss_fn(y_hat, y_gt)`), with the POI datasets' loss of data.py:76 / :98:
        NLLLoss(ignore_index=0) on log-probabilities over the classes == cross_entropy(ignore_index=0) on the logits."""
        logits = self(batched_data)
        y = batched_data.y.view(-1)
        if ops.cross_entropy_ok(logits, y):
            return ops.cross_entropy(logits, y, ignore_index=0)           # value + gradient in one launch (csrc/layer.hip)
        return F.cross_entropy(logits.float(), y.long(), ignore_index=0)

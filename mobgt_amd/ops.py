"""torch-facing wrappers of the C ABI (include/mobgt_hip.h): device pointers, strides and the current
HIP stream go in, autograd comes out.  PyTorch is plumbing here (memory, streams, autograd graph);
all arithmetic of the hot path happens in libmobgt_hip.so.  No CPU fallback exists.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import F32, BF16, I64, I32, I16, U8, check

_DT = {torch.float32: F32, torch.bfloat16: BF16}
_IT = {torch.int64: I64, torch.int32: I32, torch.int16: I16, torch.uint8: U8}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mobgt_amd ops run on the GPU only (there is no CPU fallback); got a CPU tensor")


# ---- launches whose workgroups wait for each other (cluster form of the chain kernels, the head's cluster, the one-launch GCN) ----
# They give up after a bounded wait and count the event instead of trapping (csrc/chain.hip WS_FAULT, head.hip, smallgcn.hip).
# SAFE_FORMS[0] = True makes every caller take the form WITHOUT cross-workgroup waits (one workgroup per row block, the head as
# three launches, the GCN layer by layer): what train.TrainStep.check_faults switches to before it re-runs a faulted step.
SAFE_FORMS = [os.environ.get("MOBGT_SAFE_FORMS") == "1"]


def _ws_word(ws, off):
    return ws.view(torch.int32)[off // 4:off // 4 + 1]


def _scrub_exchange_area(ws, fault_off):
    """After a faulted launch the exchange area may hold packets of members that arrived late: member 0 raises the block's
    generation whether or not it saw everybody, so a late member tags its packets gen + 2 -- the tag the NEXT launch waits
    for.  The fault word is the last of the generation words and the packets start right behind it: zero them
    (tag 0 is never a launch's; the generations stay).  The device is idle here (peer_wait_faults synchronised it)."""
    ws[fault_off + 4:].zero_()
    torch.cuda.synchronize()


def peer_wait_faults(reset=True):
    """{site: count} of workgroups that gave up waiting for their peers since the last reset, on the current device (empty dict
    = none).  Synchronises the device (it reads device words): call between steps, never during a capture."""
    from . import fused_layer
    lib = _lib.lib()
    dev = torch.cuda.current_device()
    torch.cuda.synchronize(dev)
    out = {}
    for key, ws in fused_layer._CHAIN_WS.items():           # (one workspace per device and launch geometry)
        if key[0] != dev:
            continue
        off = int(lib.mobgt_chain_ws_fault_offset())
        w = _ws_word(ws, off)
        n = int(w.item())
        if n:
            out["chain"] = out.get("chain", 0) + n
            if reset:
                w.zero_()
                _scrub_exchange_area(ws, off)
    ws = _HEAD_WS.get(dev)
    if ws is not None:
        off = int(lib.mobgt_head_chain_ws_fault_offset())
        w = _ws_word(ws, off)
        n = int(w.item())
        if n:
            out["head"] = n
            if reset:
                w.zero_()
                _scrub_exchange_area(ws, off)
    c = ctypes.c_uint32(0)
    check(lib.mobgt_small_gcn_faults(1 if reset else 0, ctypes.byref(c)), "mobgt_small_gcn_faults")
    if c.value:
        out["small_gcn"] = int(c.value)
    return out


def set_peer_wait_limit(rounds=0, gcn_ticks=0):
    """Test hook: poll rounds after which the cluster kernels give up (0 = default, seconds) and the one-launch GCN's barrier limit
    in 100 MHz ticks (0 = default, 2 s).  The workspaces must exist (an eager step creates them)."""
    from . import fused_layer
    lib = _lib.lib()
    dev = torch.cuda.current_device()
    rounds = int(rounds) & 0xFFFFFFFF                 # (0xffffffff: every wait reports a fault at once -- fault injection)
    rounds = rounds - (1 << 32) if rounds >= (1 << 31) else rounds
    for key, ws in fused_layer._CHAIN_WS.items():
        if key[0] == dev:
            _ws_word(ws, int(lib.mobgt_chain_ws_limit_offset())).fill_(int(rounds))
    ws = _HEAD_WS.get(dev)
    if ws is not None:
        _ws_word(ws, int(lib.mobgt_head_chain_ws_limit_offset())).fill_(int(rounds))
    check(lib.mobgt_small_gcn_set_wait_limit(int(gcn_ticks)), "mobgt_small_gcn_set_wait_limit")
    torch.cuda.synchronize()


class ZeroArena:
    """One f32 buffer zeroed ONCE per step from which the step's many small zero-initialised accumulators
    (bias / LayerNorm / table gradients that kernels add into) are carved, instead of one fill launch each."""

    def __init__(self, device, n=1 << 20):
        self.buf = torch.zeros(n, dtype=torch.float32, device=device)
        self.off = 0

    def reset(self):
        self.buf.zero_()
        self.off = 0

    def take(self, n):
        n4 = (n + 3) // 4 * 4                         # keep 16-byte alignment of every slice
        if self.off + n4 > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n]
        self.off += n4
        return t


_ARENA = [None]


def set_zero_arena(arena):
    _ARENA[0] = arena


def zeros_f32(shape, device):
    """torch.zeros(shape, f32) -- from the step's zero arena when one is active (see ZeroArena)."""
    n = 1
    for d in (shape if isinstance(shape, (tuple, list, torch.Size)) else (shape,)):
        n *= int(d)
    a = _ARENA[0]
    if a is not None and a.buf.device == torch.device(device):
        t = a.take(n)
        if t is not None:
            return t.view(shape)
    return torch.zeros(shape, dtype=torch.float32, device=device)


# ---- gradient sinks: where a trainer wants a parameter's gradient written (train.FlatGrads) ------------------
_GRAD_SINKS = {}


def set_grad_sinks(params, views):
    """params[i]'s gradient belongs at views[i] (a slice of the trainer's flat gradient buffer, zeroed before every
    backward).  Kernels that produce a large gradient write it THERE and return that view, so the trainer's
    gather has nothing to copy for it.  Keyed by the parameter's storage address; `None` clears."""
    import weakref
    _GRAD_SINKS.clear()
    if params is not None:
        for p, v in zip(params, views):
            _GRAD_SINKS[p.data_ptr()] = (weakref.ref(p), v)


# A sink is handed to the op that asks for it in its forward and written by that op's backward as if it were its own
# buffer (also LATER than the backward returns: deferred grouped launches).  That is only sound while ONE node per
# backward pass produces a parameter's gradient.  A parameter reached twice in one forward (tied weights, a module applied
# twice, one table in two ops) must therefore not have a sink at all: both nodes would write the same memory and autograd
# would add the two views of it.  The trainer's dry run counts the requests per forward pass
# (`sink_census_*`, train.used_parameters) and leaves such parameters out of `set_grad_sinks`: they get fresh buffers,
# autograd sums them, the gather copies the sum.
_SINK_CENSUS = {"on": False, "cur": {}, "max": {}}


def sink_census_begin():
    _SINK_CENSUS.update(on=True, cur={}, max={})


def sink_census_pass():
    """A forward pass ends / the next one begins: fold this pass's request counts into the per-parameter maximum."""
    c = _SINK_CENSUS
    for k, n in c["cur"].items():
        if n > c["max"].get(k, 0):
            c["max"][k] = n
    c["cur"] = {}


def sink_census_end():
    """-> {id(parameter): most requests seen in one forward pass}"""
    sink_census_pass()
    _SINK_CENSUS["on"] = False
    out, _SINK_CENSUS["max"] = _SINK_CENSUS["max"], {}
    return out


def grad_sink(param):
    """The registered destination of `param`'s gradient, or None.  The entry must belong to this very tensor
    object: addresses get reused, and a stale entry would redirect some other model's gradient."""
    if _SINK_CENSUS["on"]:
        _SINK_CENSUS["cur"][id(param)] = _SINK_CENSUS["cur"].get(id(param), 0) + 1
    e = _GRAD_SINKS.get(param.data_ptr())
    if e is not None and e[0]() is param and e[1].shape == param.shape:
        return e[1]
    return None


def round_up(x, m):
    return (x + m - 1) // m * m


class PackedBias:
    """The attention bias in kernel layout: `bias` [G,H,T,ld] row-major and `bias_t` (query/key
    transposed), ld = roundup(T,64) (rows start on 128-byte lines in bf16 and cover whole 64-key chunks: the attention
    kernels fetch a chunk as full row segments), pad columns -inf; plus the f32 dBias accumulator shared by all
    layers of one step and the autograd token that orders its consumer after every layer's backward."""

    def __init__(self, G, H, T, dtype, device):
        self.G, self.H, self.T = G, H, T
        self.ld = round_up(T, 64)
        self.dtype = dtype
        self.bias = torch.empty(G, H, T, self.ld, dtype=dtype, device=device)
        self.bias_t = torch.empty(G, H, T, self.ld, dtype=dtype, device=device)
        self.dbias = None
        self.n_use = 0            # attention forward passes that will want a bias gradient (one per layer)
        self.n_bwd = 0            # backward passes that have delivered theirs
        self.needs_grad = False
        self.token = None
        self.spent = False        # its gradient has been handed out (_PackFn.backward): a later forward packs afresh

    def dense(self):
        """[G,H,T,T] float32 copy (tests)."""
        return self.bias[..., : self.T].float()

    @property
    def sliced(self):
        """bf16 bias: every layer writes its own bf16 dBias slice (write-only, a quarter of the HBM traffic of
        the f32 read-modify-write accumulator) and the consumer sums them; f32 bias: one f32 accumulator."""
        return self.dtype == torch.bfloat16

    def grad_buffer(self):
        if self.dbias is None:
            # columns >= T are never written by the kernels; keep them zero
            shape = (self.G, self.H, self.T, self.ld)
            if self.sliced:
                # every element a consumer reads (columns < T) is written by the backward pass: no fill
                self.dbias = torch.empty((max(self.n_use, 1),) + shape, dtype=torch.bfloat16, device=self.bias.device)
            else:
                self.dbias = torch.zeros(shape, dtype=torch.float32, device=self.bias.device)
        return self.dbias

    def next_grad_slice(self):
        """(buffer for this backward pass, accumulate flag)."""
        buf = self.grad_buffer()
        i = self.n_bwd
        self.n_bwd += 1
        if not self.sliced:
            return buf, 1 if i > 0 else 0
        if i >= buf.shape[0]:
            raise RuntimeError("PackedBias: more attention backward passes than forward uses of this bias")
        return buf[i], 0

    def grad_total(self):
        """Sum over layers as f32 [G,H,T,T] (tests, caller-supplied-bias gradient)."""
        if self.dbias is None:
            return None
        if self.sliced:
            return self.dbias[: max(self.n_bwd, 1), ..., : self.T].float().sum(0)
        return self.dbias[..., : self.T]


# ------------------------------------------------------------------------------------- bias pack
class _PackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, pack):
        G, H, T = pack.G, pack.H, pack.T
        s = src.expand(G, H, T, T)
        st = s.stride()
        check(_lib.lib().mobgt_bias_pack(_p(s), _DT[s.dtype], st[0], st[1], st[2], st[3], _p(pack.bias), _p(pack.bias_t),
                                         _DT[pack.dtype], G, H, T, pack.ld, _stream()), "mobgt_bias_pack")
        ctx.pack = pack
        ctx.src_shape = src.shape
        ctx.src_dtype = src.dtype
        ctx.set_materialize_grads(False)
        return zeros_f32((1,), src.device)

    @staticmethod
    def backward(ctx, _g):
        pack = ctx.pack
        pack.spent = True
        if pack.dbias is None:
            return torch.zeros(ctx.src_shape, dtype=ctx.src_dtype, device=pack.bias.device), None
        g = pack.grad_total()
        if tuple(ctx.src_shape) != tuple(g.shape):          # broadcast source: reduce
            g = g.sum_to_size(ctx.src_shape)
        return g.to(ctx.src_dtype), None


def pack_bias(attn_bias, G, H, T, dtype=None):
    """Caller-supplied [G,H,T,T] (or broadcastable) bias -> PackedBias (model.py:445 semantics)."""
    _require_cuda(attn_bias)
    if attn_bias.dtype not in _DT:
        attn_bias = attn_bias.float()
    if attn_bias.dim() == 3:
        attn_bias = attn_bias.unsqueeze(1)
    pack = PackedBias(G, H, T, dtype or attn_bias.dtype, attn_bias.device)
    pack.needs_grad = attn_bias.requires_grad and torch.is_grad_enabled()
    pack.token = _PackFn.apply(attn_bias, pack)
    return pack


# ------------------------------------------------------------------------------------- build bias
# The bias tables' backward (mobgt_build_bias_bwd, 21.6 us of the S-FSQ step) depends on nothing but the dBias slices, which
# are complete once the last attention backward has run -- long before autograd reaches this node (created first, run last).
# The category GCN's backward launch (19 compute units busy for 25 us, modelGNN._SmallGcnFn) comes in between and carries it
# as passenger workgroups: _BuildBiasFn.forward leaves a job here, the GCN's backward takes it (take_bias_bwd_job) and stores
# the four gradients in it, _BuildBiasFn.backward then only hands them out.  MOBGT_NO_BIAS_BWD_PASSENGER=1: own launch.
_BIAS_BWD_JOB = {}


# Two tiny front-of-step launches -- the hop table's forward and the index derivation of the node features, 4.8 us each, all ramp --
# ride in the category GCN's forward launch as well: inside front_deferral(True) their ops only allocate their outputs and leave a
# job (take_front_jobs); the model calls the one-launch GCN next and flush_front() behind it (a job nobody took is launched
# alone there).  MOBGT_NO_FRONT_PASSENGERS=1: own launches.
_FRONT_DEFER = {"on": False, "hop": None, "ni": None}


def front_deferral(on):
    _FRONT_DEFER["on"] = bool(on) and os.environ.get("MOBGT_NO_FRONT_PASSENGERS") != "1"


def take_front_jobs():
    hop, ni = _FRONT_DEFER["hop"], _FRONT_DEFER["ni"]
    _FRONT_DEFER["hop"] = _FRONT_DEFER["ni"] = None
    return hop, ni


def flush_front():
    hop, ni = take_front_jobs()
    if hop is not None:
        check(_lib.lib().mobgt_hop_table_fwd(*hop[0], _stream()), "mobgt_hop_table_fwd")
    if ni is not None:
        check(_lib.lib().mobgt_node_index(*ni[0], _stream()), "mobgt_node_index")


def _bias_bwd_alloc(shapes, dev, sinks=(None, None, None, None)):
    """Destinations of the four table gradients (accumulated into): the parameters' gradient sinks where the trainer registered
    them (fresh views), else zeros."""
    return tuple(None if sh is None else (k[:] if (k is not None and tuple(k.shape) == tuple(sh)) else zeros_f32(sh, dev))
                 for sh, k in zip(shapes, sinks))


def take_bias_bwd_job():
    """The pending bias-backward job if its inputs are complete and it is the instantiation the passenger form covers
    (int16 indices, uint8 edge ids, 8 heads, one edge feature, <= 20 hops, a short batch); None otherwise."""
    job = _BIAS_BWD_JOB.pop("cur", None)
    if job is None or os.environ.get("MOBGT_NO_BIAS_BWD_PASSENGER") == "1":
        return None
    pack = job["pack"]()
    if pack is None or pack.dbias is None or not pack.sliced or pack.n_use < 1 or pack.n_bwd < pack.n_use:
        return None
    G, N, H, D_in, D, F, n_rel, n_poi, n_edge, ld, idx_dt, edge_dt = job["args"]
    if not (idx_dt == I16 and edge_dt == U8 and H == 8 and D > 0 and D <= 20 and F == 1 and G * (N + 1) * (N + 1) < (1 << 20)
            and job["idx"][3] is not None):
        return None
    return job


def bias_bwd_job_args(job):
    """-> (outputs, ctypes-ready argument list of mobgt_build_bias_bwd without the stream)."""
    pack = job["pack"]()
    outs = _bias_bwd_alloc(job["shapes"], pack.bias.device, job.get("sinks", (None,) * 4))
    attn_bias, rel_pos, poi_pos, edge_input = job["idx"]
    n_sl = max(pack.n_bwd, 1) if pack.sliced else 1
    stride = pack.dbias.stride(0) if pack.sliced else 0
    args = [_p(pack.dbias), _DT[pack.dbias.dtype], n_sl, stride, _p(attn_bias), _p(rel_pos), _p(poi_pos), _p(edge_input),
            _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), *job["args"]]
    return outs, args


class _BuildBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rel_table, poi_table, hop_table, vdist, pack, attn_bias, rel_pos, poi_pos, edge_input, D, sinks=(None,) * 4):
        G, N = rel_pos.shape[:2]
        H = rel_table.shape[1]
        has_edge = edge_input is not None and D > 0
        D_in = edge_input.shape[3] if has_edge else 0
        F = edge_input.shape[4] if has_edge else 1
        idx_dt = _IT[rel_pos.dtype]
        edge_dt = _IT[edge_input.dtype] if has_edge else U8
        n_poi = poi_table.shape[0] if poi_table is not None else 0
        n_edge = hop_table.shape[1] if has_edge else 0
        args = (G, N, H, D_in, D if has_edge else 0, F, rel_table.shape[0], n_poi, n_edge, pack.ld, idx_dt, edge_dt)
        check(_lib.lib().mobgt_build_bias(_p(attn_bias), _p(rel_pos), _p(poi_pos), _p(edge_input if has_edge else None), _p(rel_table),
                                          _p(poi_table), _p(hop_table if has_edge else None), _p(vdist), _p(pack.bias), _p(pack.bias_t),
                                          *args, _DT[pack.dtype], _stream()), "mobgt_build_bias")
        ctx.pack, ctx.args = pack, args
        ctx.set_materialize_grads(False)          # the token carries no gradient: no zero-fill launch to materialise one
        ctx.idx = (attn_bias, rel_pos, poi_pos, edge_input if has_edge else None)
        ctx.shapes = (rel_table.shape, None if poi_table is None else poi_table.shape,
                      hop_table.shape if has_edge else None, vdist.shape)
        import weakref
        # (weak: the pack's token belongs to this forward pass's autograd graph, which must not outlive its backward)
        ctx.job = dict(pack=weakref.ref(pack), args=args, idx=ctx.idx, shapes=ctx.shapes, done=None, sinks=sinks)
        _BIAS_BWD_JOB.pop("cur", None)
        if pack.needs_grad:                        # (build_bias() looked at the grad mode: it is off inside Function.forward)
            _BIAS_BWD_JOB["cur"] = ctx.job
        return zeros_f32((1,), rel_table.device)

    @staticmethod
    def backward(ctx, _g):
        pack = ctx.pack
        dev = pack.bias.device
        if ctx.job["done"] is not None:            # computed by the passengers of the category GCN's backward launch, or beside
            d_rel, d_poi, d_hop, d_vd = ctx.job["done"]         # the rest of the pass on the side stream (long batches)
            ev = ctx.job.pop("done_event", None)
            if ev is not None:
                torch.cuda.current_stream(pack.bias.device).wait_event(ev)
            ctx.job["done"] = None
            return d_rel, d_poi, d_hop, d_vd, None, None, None, None, None, None, None
        if _BIAS_BWD_JOB.get("cur") is ctx.job:
            del _BIAS_BWD_JOB["cur"]
        d_rel, d_poi, d_hop, d_vd = _bias_bwd_alloc(ctx.shapes, dev, ctx.job.get("sinks", (None,) * 4))
        if pack.dbias is not None:
            attn_bias, rel_pos, poi_pos, edge_input = ctx.idx
            a = ctx.args
            n_sl = max(pack.n_bwd, 1) if pack.sliced else 1
            stride = pack.dbias.stride(0) if pack.sliced else 0
            check(_lib.lib().mobgt_build_bias_bwd(_p(pack.dbias), _DT[pack.dbias.dtype], n_sl, stride,
                                                  _p(attn_bias), _p(rel_pos), _p(poi_pos), _p(edge_input),
                                                  _p(d_rel), _p(d_poi), _p(d_hop), _p(d_vd), *a, _stream()),
                  "mobgt_build_bias_bwd")
        return d_rel, d_poi, d_hop, d_vd, None, None, None, None, None, None, None


def build_bias(attn_bias, rel_pos, poi_pos, edge_input, rel_table, poi_table, hop_table, vdist, D, dtype=torch.float32):
    """Fused bias assembly (model.py:126-190 / model_fqandtoyo.py:1143-1216) -> PackedBias.
    `hop_table` [D, n_edge, H] is the (differentiable) product of the edge and hop-distance tables."""
    _require_cuda(attn_bias, rel_pos, rel_table)
    G, N = rel_pos.shape[:2]
    H = rel_table.shape[1]
    pack = PackedBias(G, H, N + 1, dtype, rel_table.device)
    pack.needs_grad = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (rel_table, poi_table, hop_table, vdist))
    f = lambda t: None if t is None else t.contiguous()
    # gradient sinks of the tables that are trained parameters (looked up on the parameter objects themselves)
    k_vd = grad_sink(vdist)
    sinks = (grad_sink(rel_table), grad_sink(poi_table) if poi_table is not None else None, None,
             k_vd.view(-1) if (k_vd is not None and k_vd.is_contiguous()) else None)
    pack.token = _BuildBiasFn.apply(f(rel_table.float()), f(None if poi_table is None else poi_table.float()),
                                    f(None if hop_table is None else hop_table.float()), f(vdist.float().reshape(-1)),
                                    pack, f(attn_bias.float()), f(rel_pos), f(poi_pos), f(edge_input), int(D), sinks)
    return pack


# -------------------------------------------------------------------------------------- attention
_ATTN_ONE_PASS = [os.environ.get("MOBGT_ATTN_TWO_PASS") != "1"]       # T > 64, bf16: one-pass backward (round 4); =1: the two passes


def _check_rows(*ts):
    for t in ts:
        if t.stride(-1) != 1 or t.stride(-2) % 8 != 0 or t.data_ptr() % 16 != 0:
            raise RuntimeError("mobgt attention: q/k/v rows must be unit-stride, 16-byte aligned, row stride % 8 == 0")


def _lse_alloc(G, H, T, C, dtype, device):
    """The forward's row statistics: [G, H, T] f32 log-sum-exp and, behind them in the SAME buffer for bf16 I/O, the bf16 rounding
    residual of the output ([G, T, C] bf16 = G T C / 2 floats; csrc/attn.hip, "consistent softmax": the backward's
    rowsum(dO O) needs O beyond bf16).  One tensor, so everything that saves / hands on `lse` carries the residual along."""
    return torch.empty(lse_numel(G, H, T, C, dtype), dtype=torch.float32, device=device)


def lse_numel(G, H, T, C, dtype):
    n = G * H * T
    extra = (G * T * C + 1) // 2 if dtype == torch.bfloat16 else 0
    return (n + 3) // 4 * 4 + extra                # (the residual starts on a 16-byte boundary)


def lse_rows(lse, G, H, T):
    """[G, H, T] view of the log-sum-exp rows inside the buffer `_attn_fwd` returns."""
    return lse[: G * H * T].view(G, H, T)


def _out_lo_ptr(lse, G, H, T, dtype):
    if dtype != torch.bfloat16:
        return None
    return lse.data_ptr() + 4 * ((G * H * T + 3) // 4 * 4)


def _attn_fwd(q, k, v, pack, scale, p_drop, seed, seed_dev):
    G, T, C = q.shape
    H = pack.H
    out = torch.empty(G, T, C, dtype=q.dtype, device=q.device)
    lse = _lse_alloc(G, H, T, C, q.dtype, q.device)
    _check_rows(q, k, v)
    if pack.needs_grad:
        pack.n_use += 1
    check(_lib.lib().mobgt_attn_bias_fwd(_p(q), _p(k), _p(v), _p(pack.bias), _p(out), _out_lo_ptr(lse, G, H, T, q.dtype), _p(lse), G, H, T, C // H,
                                         q.stride(1), k.stride(1), v.stride(1), C, pack.ld, scale, p_drop, seed,
                                         _p(seed_dev), _DT[q.dtype], _DT[pack.dtype], _stream()), "mobgt_attn_bias_fwd")
    return out, lse


_DQ_ACC = {}
_DQ_ACC_RETIRED = []      # accumulators a failed call left in an unknown state: never freed (captured graphs may replay their address)


def _attn_bwd(q, k, v, out, lse, dout, dq, dk, dv, pack, scale, p_drop, seed, seed_dev):
    G, T, C = q.shape
    H = pack.H
    #  `lse` is the flat buffer _attn_fwd returns: [G,H,T] f32 rows and, for bf16 I/O, the bf16 residual of `out` behind
    # them.  A caller-made [G,H,T] tensor (the contract of round 4; mobgt::attention_backward takes any tensor) would make the
    # kernels read G*T*C bf16 past its end.
    need = lse_numel(G, H, T, C, q.dtype) if q.dtype == torch.bfloat16 else G * H * T
    if lse.dtype != torch.float32 or lse.numel() < need:
        raise ValueError(f"attention backward: `lse` must be the float32 buffer of {need} elements the "
                         f"forward returned (log-sum-exp rows + the bf16 residual of `out`), got {lse.dtype} x {lse.numel()}")
    dbias = None
    acc = 0
    if pack.needs_grad:
        dbias, acc = pack.next_grad_slice()
    delta = torch.empty(G, H, T, dtype=torch.float32, device=q.device)
    args = (_p(q), _p(k), _p(v), _p(pack.bias), _p(pack.bias_t), _p(out), _out_lo_ptr(lse, G, H, T, q.dtype), _p(lse), _p(dout),
            _p(dq), _p(dk), _p(dv), _p(dbias), _p(delta), G, H, T, C // H,
            q.stride(1), k.stride(1), v.stride(1), C, dq.stride(1), dk.stride(1),
            dv.stride(1), pack.ld, scale, p_drop, seed, _p(seed_dev), acc,
            _DT[dbias.dtype] if dbias is not None else F32, _DT[q.dtype], _DT[pack.dtype])
    if (T > 64 and q.dtype == torch.bfloat16 and pack.dtype == torch.bfloat16 and dbias is not None
            and dbias.dtype == torch.bfloat16 and _ATTN_ONE_PASS[0]):
        # long graphs, training configuration: ONE pass over the bias (csrc/attn.hip: attn_bwd_one_kernel) -- needs an f32
        # scratch accumulator for dQ (summed over key blocks by atomics)
        # ONE accumulator per (device, STREAM, size), zero between calls: the pass adds into it, the finishing launch reads and
        # re-zeroes it.  Calls of one stream follow each other; two streams (a trainer's step graph beside an eager or eval
        # backward of the same size) each get their own -- their atomics must not meet in one buffer.  `busy` catches a
        # call that died between its launches: that buffer is retired, not freed (a captured graph may hold its address).
        # dQ is summed by f32 atomics over key blocks: not bitwise reproducible (MOBGT_ATTN_TWO_PASS=1: the two deterministic passes).
        key = (str(q.device), int(torch.cuda.current_stream(q.device).cuda_stream), G * T * C)
        ent = _DQ_ACC.get(key)
        if ent is None or ent["busy"]:
            if ent is not None:
                _DQ_ACC_RETIRED.append(ent["buf"])
            ent = _DQ_ACC[key] = dict(buf=torch.zeros(G * T * C, dtype=torch.float32, device=q.device), busy=False)
        ent["busy"] = True
        check(_lib.lib().mobgt_attn_bias_bwd_fused_z(*args, _p(ent["buf"]), _stream()), "mobgt_attn_bias_bwd_fused_z")
        ent["busy"] = False
        _bias_bwd_beside(pack)
        return
    check(_lib.lib().mobgt_attn_bias_bwd(*args, _stream()), "mobgt_attn_bias_bwd")
    _bias_bwd_beside(pack)


# Long batches (round 6): the bias tables' backward (mobgt_build_bias_bwd: 0.7 ms at S-BIG) depends on nothing but the dBias
# slices, which are complete when the LAST attention backward of the pass has run -- and behind that point the backward pass
# still has ~0.7 ms of small launches to go (the encoder input's, the embedding tables', the GCNs': latency-bound, a few
# workgroups each).  The moment the last slice has been written the job is therefore launched on a SIDE stream (an event fork;
# inside a hipGraph capture: a parallel branch of the graph), and _BuildBiasFn.backward -- which autograd runs last -- only waits
# for it.  Short batches keep the passenger form (the category GCN's backward launch carries the job: take_bias_bwd_job).
# _BIAS_BWD_BESIDE[0] = False (tests): the launch stays where autograd reaches it.
_BIAS_BWD_BESIDE = [True]
_SIDE_STREAMS = {}


def _bias_bwd_beside(pack):
    job = _BIAS_BWD_JOB.get("cur")
    if (job is None or not _BIAS_BWD_BESIDE[0] or job["pack"]() is not pack or not pack.sliced or pack.dbias is None
            or pack.n_use < 1 or pack.n_bwd < pack.n_use or job["done"] is not None):
        return
    G, N = job["args"][0], job["args"][1]
    if G * (N + 1) * (N + 1) < (1 << 20) or job["idx"][3] is None:     # (short batches: the passenger of the GCN's backward launch)
        return
    dev = pack.bias.device
    main = torch.cuda.current_stream(dev)
    side = _SIDE_STREAMS.get(dev)
    if side is None:
        side = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    del _BIAS_BWD_JOB["cur"]
    outs, args = bias_bwd_job_args(job)             # (allocated on the main stream: zero-filled sinks / arena views)
    fork = torch.cuda.Event()
    fork.record(main)
    side.wait_event(fork)
    # three quarters of the compute units for this launch (one persistent workgroup each), the rest for the main stream's small
    # launches beside it: S-BIG 7.17 ms (launch in autograd's order) / 7.25 (beside, all units) / 6.95 (beside, 192 of 256) /
    # 7.09 (160) / 7.39 (128), back to back on one box
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    lib = _lib.lib()
    lib.mobgt_build_bias_bwd_set_workgroups(max(1, cus * 3 // 4))
    try:
        check(lib.mobgt_build_bias_bwd(*args, ctypes.c_void_p(side.cuda_stream)), "mobgt_build_bias_bwd")
    finally:
        lib.mobgt_build_bias_bwd_set_workgroups(0)
    done = torch.cuda.Event()
    done.record(side)
    job["done"], job["done_event"] = outs, done


class _AttnFn(torch.autograd.Function):
    """q, k, v: [G,T,C] (possibly strided views); returns [G,T,C]."""

    @staticmethod
    def forward(ctx, q, k, v, token, pack, scale, p_drop, seed, seed_dev):
        out, lse = _attn_fwd(q, k, v, pack, scale, p_drop, seed, seed_dev)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.misc = (pack, scale, p_drop, seed, seed_dev)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        pack, scale, p_drop, seed, seed_dev = ctx.misc
        dout = dout.contiguous()
        dq, dk, dv = torch.empty_like(out), torch.empty_like(out), torch.empty_like(out)
        _attn_bwd(q, k, v, out, lse, dout, dq, dk, dv, pack, scale, p_drop, seed, seed_dev)
        return dq, dk, dv, None, None, None, None, None, None


class _AttnQKVFn(torch.autograd.Function):
    """qkv: [G,T,3C] (one fused projection); returns [G,T,C]; gradient comes back fused as well."""

    @staticmethod
    def forward(ctx, qkv, token, pack, scale, p_drop, seed, seed_dev):
        C = qkv.shape[2] // 3
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        out, lse = _attn_fwd(q, k, v, pack, scale, p_drop, seed, seed_dev)
        ctx.save_for_backward(qkv, out, lse)
        ctx.misc = (pack, scale, p_drop, seed, seed_dev)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        pack, scale, p_drop, seed, seed_dev = ctx.misc
        C = qkv.shape[2] // 3
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        dqkv = torch.empty_like(qkv)
        _attn_bwd(q, k, v, out, lse, dout.contiguous(), dqkv[..., :C], dqkv[..., C:2 * C], dqkv[..., 2 * C:], pack,
                  scale, p_drop, seed, seed_dev)
        return dqkv, None, None, None, None, None, None


def attention(q, k, v, pack, scale, p_drop=0.0, seed=0, seed_dev=None):
    _require_cuda(q, k, v)
    return _AttnFn.apply(q, k, v, pack.token, pack, float(scale), float(p_drop), int(seed), seed_dev)


def attention_qkv(qkv, pack, scale, p_drop=0.0, seed=0, seed_dev=None):
    _require_cuda(qkv)
    if qkv.shape[2] % 24 != 0:
        raise RuntimeError("fused qkv width must be 3*C with C % 8 == 0")
    return _AttnQKVFn.apply(qkv.contiguous(), pack.token, pack, float(scale), float(p_drop), int(seed), seed_dev)


def dropout_keep_mask(seed, G, H, T, p_drop):
    """Host replay of the attention kernels' keep rule (tests): bool [G,H,T,T] (mobgt_attn_dropout_mask_host)."""
    import numpy as np
    m = np.empty((G, H, T, T), dtype=np.uint8)
    check(_lib.lib().mobgt_attn_dropout_mask_host(int(seed) & 0xFFFFFFFFFFFFFFFF, G, H, T, float(p_drop), m.ctypes.data),
          "mobgt_attn_dropout_mask_host")
    return m.astype(bool)


def dropout_site_mask(seed, salt, R, C, p_drop, row0=0):
    """Host replay of the keep rule of every non-attention dropout site (tests): bool [R,C]; `seed` = host seed + device
    step counter, `salt` = the site's constant, rows numbered from `row0` (mobgt_dropout_mask_host)."""
    import numpy as np
    m = np.empty((R, C), dtype=np.uint8)
    check(_lib.lib().mobgt_dropout_mask_host(int(seed) & 0xFFFFFFFFFFFFFFFF, int(salt) & 0xFFFFFFFF, int(row0), R, C, float(p_drop),
                                             m.ctypes.data), "mobgt_dropout_mask_host")
    return m.astype(bool)


# ------------------------------------------------------------------------------------------- spd
def spd_batched(counts, n_nodes, D):
    """counts [G,N,N] int32 (device), n_nodes [G] int32 -> dict of device tensors (see mobgt_spd_batched)."""
    _require_cuda(counts, n_nodes)
    G, N = counts.shape[:2]
    dev = counts.device
    out = dict(
        spd=torch.empty(G, N, N, dtype=torch.int16, device=dev), path=torch.empty(G, N, N, dtype=torch.int16, device=dev),
        rel_pos=torch.empty(G, N, N, dtype=torch.int16, device=dev),
        edge_input=torch.empty(G, N, N, D, 1, dtype=torch.uint8, device=dev),
        in_degree=torch.empty(G, N, dtype=torch.int16, device=dev), out_degree=torch.empty(G, N, dtype=torch.int16, device=dev))
    work = torch.empty(int(_lib.lib().mobgt_spd_workspace_bytes(G, N)), dtype=torch.uint8, device=dev)
    check(_lib.lib().mobgt_spd_batched(_p(counts.contiguous()), _p(n_nodes.contiguous()), _p(out["spd"]), _p(out["path"]),
                                       _p(out["rel_pos"]), _p(out["edge_input"]), _p(out["in_degree"]),
                                       _p(out["out_degree"]), _p(work), G, N, D, _stream()), "mobgt_spd_batched")
    return out


# ----------------------------------------------------------------------------------------- embed
def _ptr_array(ts):
    arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    return arr


class _GatherSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, n_tables, *args):
        tables, idx = args[:n_tables], args[n_tables:]
        R = idx[0].numel()
        C = tables[0].shape[1]
        out = torch.empty(R, C, dtype=torch.float32, device=tables[0].device)
        check(_lib.lib().mobgt_embed_gather_sum(_ptr_array(tables), _ptr_array(idx), n_tables, _p(out), R, C, C,
                                                _IT[idx[0].dtype], _stream()), "mobgt_embed_gather_sum")
        ctx.idx, ctx.skip = idx, skip
        ctx.shapes = [t.shape for t in tables]
        # gradient sinks of the tables that are trained parameters (a table listed twice keeps separate buffers: autograd adds
        # the two results, which must not be one memory)
        ptrs = [t.data_ptr() for t in tables]
        ctx.sinks = [grad_sink(t) if ptrs.count(t.data_ptr()) == 1 else None for t in tables]
        return out

    @staticmethod
    def backward(ctx, dout):
        if dout.stride(1) != 1 or dout.stride(0) % 4 or dout.data_ptr() % 16:      # column slices of a wider matrix are fine
            dout = dout.contiguous()
        n = len(ctx.shapes)
        # (the scatter ACCUMULATES: a sink -- zeroed by the trainer's prologue -- serves as it is, a fresh view per call)
        grads = [k[:] if (k is not None and tuple(k.shape) == tuple(s)) else zeros_f32(tuple(s), dout.device)
                 for s, k in zip(ctx.shapes, ctx.sinks)]
        skip = (ctypes.c_int64 * n)(*ctx.skip)
        R, C = dout.shape
        check(_lib.lib().mobgt_embed_scatter_add(_ptr_array(grads), _ptr_array(ctx.idx), skip, n, _p(dout), R, C,
                                                 dout.stride(0), _IT[ctx.idx[0].dtype], _stream()), "mobgt_embed_scatter_add")
        return (None, None, *grads, *([None] * n))


class _HopTableFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, edge_weight, edge_dis_weight, H, D, fp16_roundtrip):
        E = edge_weight.shape[0]
        ew, dw = edge_weight.contiguous(), edge_dis_weight.contiguous()
        tab = torch.empty(D, E, H, dtype=torch.float32, device=ew.device)
        args = [_p(ew), _p(dw), _p(tab), D, E, H, int(fp16_roundtrip)]
        if _FRONT_DEFER["on"]:
            if _FRONT_DEFER["hop"] is not None:
                flush_front()
            _FRONT_DEFER["hop"] = (args, (ew, dw, tab))          # (rides in the category GCN's forward launch)
        else:
            check(_lib.lib().mobgt_hop_table_fwd(*args, _stream()), "mobgt_hop_table_fwd")
        ctx.save_for_backward(ew, dw)
        ctx.misc = (H, D, int(fp16_roundtrip), edge_dis_weight.shape)
        ctx.sinks = (grad_sink(edge_weight), grad_sink(edge_dis_weight))
        return tab

    @staticmethod
    def backward(ctx, dtab):
        ew, dw = ctx.saved_tensors
        H, D, rt, dis_shape = ctx.misc
        E = ew.shape[0]
        k_e, k_d = ctx.sinks
        d_ew = k_e[:] if k_e is not None else torch.empty_like(ew)          # (written in full)
        d_dw = k_d[:] if k_d is not None else zeros_f32(tuple(dis_shape), ew.device)     # rows of hop slots >= D get no gradient
        dtab = dtab.contiguous()
        # (parked only into a FREE slot and only when both gradients land in sinks: a second hop table in the same backward pass --
        #  two models, two tables -- would overwrite the first one's entry, whose gradient would then never be computed, and a
        #  buffer allocated here would reach autograd before the deferred launch has filled it: ADVICE r4)
        park = _WGRAD_DEFER["on"] and k_e is not None and k_d is not None
        if (park and "hop" not in _WGRAD_DEFER and H == 8 and E <= 256 and dtab.data_ptr() % 16 == 0 and dw.data_ptr() % 16 == 0):
            # rides in the step's grouped weight-gradient launch (flush_deferred_wgrads): nothing but the optimizer reads these
            _WGRAD_DEFER["hop"] = (dtab, ew, dw, d_ew, d_dw, D, E, rt)
            return d_ew[:], d_dw[:], None, None, None
        if (park and "hop_wide" not in _WGRAD_DEFER and H == 8 and E <= 2048 and dtab.data_ptr() % 16 == 0 and dw.data_ptr() % 16 == 0
                and os.environ.get("MOBGT_NO_STOCK_TAIL") != "1"):
            # too many edge ids for the grouped launch's hop slot (the stock variant's 1 537): parked until the flush, where it
            # shares ONE grid with the backward of the stock encoder input when that is parked too (csrc/layer.hip stock_tail_kernel)
            _WGRAD_DEFER["hop_wide"] = (dtab, ew, dw, d_ew, d_dw, D, E, rt)
            return d_ew[:], d_dw[:], None, None, None
        check(_lib.lib().mobgt_hop_table_bwd(_p(dtab), _p(ew), _p(dw), _p(d_ew), _p(d_dw), D, E, H, rt,
                                             _stream()), "mobgt_hop_table_bwd")
        return d_ew, d_dw, None, None, None


def hop_table(edge_weight, edge_dis_weight, H, D, fp16_roundtrip=False):
    """[D, n_edge, H] hop table T[d,e,:] = edge_encoder[e,:] . W_d (model.py:166-176) with the fq variant's fp16
    rounding points (model_fqandtoyo.py:1178-1198) and nn.Embedding(padding_idx=0)'s "row 0 gets no gradient"."""
    _require_cuda(edge_weight, edge_dis_weight)
    assert edge_weight.dtype == torch.float32 and edge_dis_weight.dtype == torch.float32
    assert edge_dis_weight.numel() >= D * H * H
    return _HopTableFn.apply(edge_weight, edge_dis_weight, H, D, fp16_roundtrip)


class _SkinnyLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x, w = x.contiguous(), weight.contiguous()
        if x.data_ptr() % 16:
            x = x.clone()
        G, K = x.shape
        V = w.shape[0]
        # measured at G = 16, K = 448, V = 7857: the library's forward / dx products 28.6 / 29 us, the kernels here
        # 38 / 32 us (one W row in flight per wave; 32-byte column pieces) -- but dW + db 9.5 us vs 26 us + a reduce.
        # So only the weight / bias gradient goes to csrc/skinny.hip; `MOBGT_SKINNY_ALL=1` routes all three (tests).
        ctx.all_hip = bool(os.environ.get("MOBGT_SKINNY_ALL"))
        if ctx.all_hip:
            y = torch.empty(G, V, dtype=torch.float32, device=x.device)
            check(_lib.lib().mobgt_skinny_linear_fwd(_p(x), _p(w), _p(bias), _p(y), G, K, V, _stream()),
                  "mobgt_skinny_linear_fwd")
        elif K % 64 == 0 and K <= 448:
            # one pass over W on the matrix cores (csrc/skinny.hip; the library's M = 16 GEMM: 9.2 us at K = 320, 29 us at K = 128)
            y = torch.empty(G, V, dtype=torch.float32, device=x.device)
            check(_lib.lib().mobgt_skinny_linear_fwd_mfma(_p(x), _p(w), _p(bias), _p(y), G, K, V, _stream()),
                  "mobgt_skinny_linear_fwd_mfma")
        else:
            y = torch.addmm(bias, x, w.t()) if bias is not None else x @ w.t()
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.sink = grad_sink(weight)
        ctx.sink_b = grad_sink(bias) if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        return _skinny_backward(ctx, dy)


def _skinny_backward(ctx, dy):
    """dx, dW, db of y = x W^T + b for `dy` (shared by _SkinnyLinearFn and _SkinnyLinearGtlFn; ctx: saved (x, w), all_hip,
    has_bias, sink, sink_b)."""
    x, w = ctx.saved_tensors[:2]
    G, K = x.shape
    V = w.shape[0]
    dy = dy.contiguous()
    dx = None
    dx_mfma = ctx.needs_input_grad[0] and not ctx.all_hip and K % 16 == 0 and K <= 512
    both = dx_mfma and ctx.needs_input_grad[1]
    if dx_mfma:     # one pass over W at the full L1 rate (the library's 16x16 tiles: 26 us at V = 7857, K = 448)
        dx = zeros_f32((G, K), x.device)
        if not both:
            check(_lib.lib().mobgt_skinny_linear_dx(_p(dy), _p(w), _p(dx), G, K, V, _stream()), "mobgt_skinny_linear_dx")
    elif ctx.needs_input_grad[0]:
        dx = torch.empty_like(x) if ctx.all_hip else dy @ w
    dw = None
    if ctx.needs_input_grad[1]:
        dw = ctx.sink[:] if ctx.sink is not None else torch.empty_like(w)      # (a fresh view object of the sink)
    db = None
    if ctx.has_bias and ctx.needs_input_grad[2]:      # (written in full by the kernel: the sink needs no zeroing for it)
        db = ctx.sink_b[:] if ctx.sink_b is not None else torch.empty(V, dtype=torch.float32, device=x.device)
    if both:        # dx and dW (+ db) share nothing but dy: one launch, the first workgroups run the dx body
        check(_lib.lib().mobgt_skinny_linear_bwd_both(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), G, K, V, _stream()),
              "mobgt_skinny_linear_bwd_both")
        return dx, dw, db
    check(_lib.lib().mobgt_skinny_linear_bwd(_p(dy), _p(x), _p(w), _p(dx if ctx.all_hip else None), _p(dw), _p(db),
                                             G, K, V, _stream()), "mobgt_skinny_linear_bwd")
    return dx, dw, db


class _SkinnyLinearGtlFn(torch.autograd.Function):
    """loss = GradientTailLoss(x W^T + b, targets + target_offset, alpha) in ONE launch (csrc/skinny.hip:
    mobgt_skinny_linear_gtl); the backward is the skinny Linear's, fed with the d loss / d logits the forward left behind."""

    @staticmethod
    def forward(ctx, x, weight, bias, targets, alpha, target_offset, logits_out):
        x, w = x.contiguous(), weight.contiguous()
        if x.data_ptr() % 16:
            x = x.clone()
        G, K = x.shape
        V = w.shape[0]
        dz = torch.empty(G, V, dtype=torch.float32, device=x.device)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        y = None
        if logits_out is not None:
            y = logits_out.t = torch.empty(G, V, dtype=torch.float32, device=x.device)
        check(_lib.lib().mobgt_skinny_linear_gtl(_p(x), _p(w), _p(bias), _p(targets.long().contiguous()), int(target_offset), _p(y), _p(dz),
                                                 _p(loss), G, K, V, float(alpha), _stream()), "mobgt_skinny_linear_gtl")
        ctx.save_for_backward(x, w, dz)
        ctx.all_hip = False
        ctx.has_bias = bias is not None
        ctx.sink = grad_sink(weight)
        ctx.sink_b = grad_sink(bias) if bias is not None else None
        return loss

    @staticmethod
    def backward(ctx, g):
        dz = ctx.saved_tensors[2]
        if g.data_ptr() != unit_grad(g.device).data_ptr():      # (the trainer supplies d loss / d loss = 1: no multiply)
            dz = dz * g
        return (*_skinny_backward(ctx, dz), None, None, None, None)


def skinny_linear_gtl_ok(x, weight):
    return (skinny_linear_ok(x, weight) and x.shape[1] % 64 == 0 and x.shape[1] <= 448)


def skinny_linear_gtl(x, weight, bias, targets, alpha=0.25, target_offset=0, logits_out=None):
    """gradient_tail_loss(F.linear(x, weight, bias), targets, alpha, target_offset) for the classifier head (G <= 16 rows).
    `logits_out`: an _OutRef whose `.t` receives the logits (detached), else they are never stored."""
    _require_cuda(x, weight, targets)
    return _SkinnyLinearGtlFn.apply(x, weight, bias, targets[: x.shape[0]].reshape(-1), alpha, target_offset, logits_out)


def skinny_linear_ok(x, weight):
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and 0 < x.shape[0] <= 16 and x.shape[1] % 4 == 0 and x.shape[1] <= 512 and weight.shape[0] >= 1024
            and weight.is_contiguous() and weight.data_ptr() % 16 == 0)


def skinny_linear(x, weight, bias=None):
    """F.linear(x, weight, bias) for a handful of rows and a wide output (the classifier head): one pass over the
    weight per product instead of a GEMM with M = 16."""
    _require_cuda(x, weight)
    return _SkinnyLinearFn.apply(x, weight, bias)


def gather_rows_t(a, rows):
    """(a[rows], a[rows]^T) for a bf16 matrix in one pass: [R,C] and [C,R] contiguous."""
    _require_cuda(a, rows)
    assert a.dtype == torch.bfloat16 and a.stride(1) == 1 and rows.dtype == torch.int64
    R, C = rows.numel(), a.shape[1]
    out = torch.empty(R, C, dtype=a.dtype, device=a.device)
    out_t = torch.empty(C, R, dtype=a.dtype, device=a.device)
    check(_lib.lib().mobgt_gather_rows_t(_p(a), a.stride(0), _p(rows.contiguous()), _p(out), _p(out_t), R, C, _stream()),
          "mobgt_gather_rows_t")
    return out, out_t


def target_rank(scores, target):
    """[G,2] int32: number of classes ranked ahead of target[g] in scores[g] (column 0: ties broken towards the
    lower index = stable top-k; column 1: towards the higher index = the reference's reversed argsort)."""
    _require_cuda(scores, target)
    scores = scores.float().contiguous()
    target = target.reshape(-1).long().contiguous()
    G, V = scores.shape
    rank = torch.empty(G, 2, dtype=torch.int32, device=scores.device)
    check(_lib.lib().mobgt_target_rank(_p(scores), _p(target), _p(rank), G, V, _stream()), "mobgt_target_rank")
    return rank


def node_index(x, time_normal, poi2cat, rows_only, in_degree=None, out_degree=None):
    """Row indices of the node-feature gathers (model_fqandtoyo.py:1259-1264, 1287-1298) in one launch.
    x [G,N] int64/int32 POI ids, time_normal [G,N] f32 -> (idx [8,G,N] int64, real [G,N] f32); rows of idx:
    POI row, time slot, category row, positional row (all -1 where there is none), GCN row max(x-1,0), zeros,
    in-degree, out-degree (the last two only when the [G,N] degree tensors are given; widened to int64)."""
    _require_cuda(x, time_normal, poi2cat)
    assert x.dtype in (torch.int64, torch.int32) and time_normal.dtype == torch.float32 and poi2cat.dtype == torch.int64
    G, N = x.shape
    idx = torch.empty(8, G, N, dtype=torch.int64, device=x.device)
    real = torch.empty(G, N, dtype=torch.float32, device=x.device)
    deg_dt = I64
    if in_degree is not None:
        in_degree, out_degree = in_degree.reshape(G, N).contiguous(), out_degree.reshape(G, N).contiguous()
        assert in_degree.dtype == out_degree.dtype and in_degree.dtype in _IT
        deg_dt = _IT[in_degree.dtype]
    args = [_p(x), _IT[x.dtype], x.stride(0), x.stride(1), _p(time_normal), time_normal.stride(0), time_normal.stride(1), _p(poi2cat),
            _p(in_degree), _p(out_degree), deg_dt, _p(idx), _p(real), G, N, int(bool(rows_only))]
    if _FRONT_DEFER["on"]:
        if _FRONT_DEFER["ni"] is not None:
            flush_front()
        _FRONT_DEFER["ni"] = (args, (x, time_normal, poi2cat, in_degree, out_degree, idx, real))     # (see front_deferral)
    else:
        check(_lib.lib().mobgt_node_index(*args, _stream()), "mobgt_node_index")
    return idx, real


def embed_gather_sum(tables, indices, padding_idx=None):
    """sum_t tables[t][indices[t]] -> [*indices[0].shape, C].  f32 tables of equal width; indices share
    dtype and shape; negative index = no contribution.  `padding_idx[t]` rows get no gradient."""
    _require_cuda(*tables, *indices)
    n = len(tables)
    shape = indices[0].shape
    skip = tuple(-1 if (padding_idx is None or padding_idx[t] is None) else int(padding_idx[t]) for t in range(n))
    idx = [i.contiguous().reshape(-1) for i in indices]
    tabs = [t.float().contiguous() for t in tables]
    out = _GatherSumFn.apply(skip, n, *tabs, *idx)
    return out.view(*shape, tabs[0].shape[1])


class _GatherConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skip, n_tables, *args):
        tables, idx = args[:n_tables], args[n_tables:]
        R = idx[0].numel()
        widths = [t.shape[1] for t in tables]
        ctot = sum(widths)
        out = torch.empty(R, ctot, dtype=torch.float32, device=tables[0].device)
        warr = (ctypes.c_int * n_tables)(*widths)
        check(_lib.lib().mobgt_embed_gather_concat(_ptr_array(tables), _ptr_array(idx), warr, n_tables, _p(out), R, ctot,
                                                   _IT[idx[0].dtype], _stream()), "mobgt_embed_gather_concat")
        ctx.idx, ctx.skip, ctx.widths = idx, skip, widths
        ctx.shapes = [t.shape for t in tables]
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        n = len(ctx.shapes)
        R, ctot = dout.shape
        grads = [zeros_f32(tuple(s), dout.device) for s in ctx.shapes]
        skip = (ctypes.c_int64 * n)(*ctx.skip)
        warr = (ctypes.c_int * n)(*ctx.widths)
        check(_lib.lib().mobgt_embed_scatter_concat(_ptr_array(grads), _ptr_array(ctx.idx), skip, warr, n, _p(dout), R, ctot,
                                                    _IT[ctx.idx[0].dtype], _stream()), "mobgt_embed_scatter_concat")
        return (None, None, *grads, *([None] * n))


def embed_gather_concat(tables, indices, padding_idx=None):
    """cat_t tables[t][indices[t]] along the feature axis -> [*shape, sum C_t] (widths multiples of 4)."""
    _require_cuda(*tables, *indices)
    n = len(tables)
    shape = indices[0].shape
    skip = tuple(-1 if (padding_idx is None or padding_idx[t] is None) else int(padding_idx[t]) for t in range(n))
    idx = [i.contiguous().reshape(-1) for i in indices]
    tabs = [t.float().contiguous() for t in tables]
    out = _GatherConcatFn.apply(skip, n, *tabs, *idx)
    return out.view(*shape, out.shape[1])


_ROW0_PENDING = {}      # table data_ptr -> [W] f32 gradient of its row 0 produced elsewhere, consumed by _GatherMultiFn.backward


# ------------------------------------------------------------------ several gathers of one position list, one launch
# ---- the encoder input's FORWARD as one launch (round 4; csrc/chain.hip token_fwd_chain_kernel) ---------------------------------------
# gather -> FuseEmbeddings-2 -> FuseEmbeddings-4 -> token assembly + first QKV are four autograd nodes in a row whose intermediate
# results nobody reads before the last one has run.  While `token_fwd_deferral(True)` is in force (model_fqandtoyo.node_features
# switches it on when the last node can take the fused launch), the first three only ALLOCATE their outputs and record their
# launch; _AssembleTokensFn then issues mobgt_token_fwd_chain, which fills every one of those buffers -- or, if what was
# recorded is not the chain it knows, `flush_token_fwd()` issues the recorded launches one by one.  The autograd graph, the saved
# tensors and therefore the whole backward pass are those of the separate launches.
_TOKEN_FWD = {"on": False, "gather": None, "f2": None, "f4": None, "fused_calls": 0}


def token_fwd_deferral(on):
    """Switch the recording on / off.  Switching it (either way) first launches whatever is still recorded."""
    flush_token_fwd()
    _TOKEN_FWD["on"] = bool(on) and os.environ.get("MOBGT_NO_TOKEN_FWD_CHAIN") != "1" and not _NAN_TRACE["on"]


def flush_token_fwd():
    """Issue the recorded launches separately, in order (the fallback of the fused launch)."""
    for k in ("gather", "f2", "f4"):
        rec, _TOKEN_FWD[k] = _TOKEN_FWD[k], None
        if rec is not None:
            rec["launch"]()


class _GatherMultiFn(torch.autograd.Function):
    """outs[o][r, coff : coff + W] (=, or += when accum) tables[t][idx[t][r], :] for the jobs `spec` = [(o, coff, accum,
    skip)], one per table; `outs_shape` = [(width_o)].  Columns of an output that no job writes are left unwritten (the
    caller fills them: FuseEmbeddings' first Linear writes its result next to the category rows)."""

    @staticmethod
    def forward(ctx, spec, out_widths, n, *args):
        tables, idx = args[:n], args[n:]
        R = idx[0].numel()
        dev = tables[0].device
        outs = [torch.empty(R, w, dtype=torch.float32, device=dev) for w in out_widths]
        ci, i64, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p

        def launch():
            check(_lib.lib().mobgt_embed_gather_multi(
                n, _ptr_array(tables), None, _ptr_array(idx), (i64 * n)(*[sp[3] for sp in spec]),
                (ci * n)(*[t.shape[1] for t in tables]), (ci * n)(*[sp[1] for sp in spec]), (ci * n)(*[int(sp[2]) for sp in spec]),
                (vp * n)(*[outs[sp[0]].data_ptr() for sp in spec]), (i64 * n)(*[outs[sp[0]].stride(0) for sp in spec]), R,
                _IT[idx[0].dtype], 0, None, 0, _stream()), "mobgt_embed_gather_multi")
        if _TOKEN_FWD["on"] and _TOKEN_FWD["gather"] is None and len(outs) == 3 and all(i_.dtype == idx[0].dtype for i_ in idx):
            _TOKEN_FWD["gather"] = dict(launch=launch, tables=tables, idx=idx, spec=spec, outs=outs, R=R)     # (see token_fwd_deferral)
        else:
            launch()
        ctx.idx, ctx.spec, ctx.n = idx, spec, n
        ctx.ptrs = [t.data_ptr() for t in tables]
        ctx.shapes = [t.shape for t in tables]
        ctx.sinks = [grad_sink(t) for t in tables]          # (a table that is a trained parameter: its slice of the flat gradient)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        n, spec = ctx.n, ctx.spec
        dev = ctx.idx[0].device
        R = ctx.idx[0].numel()
        gbuf = []
        for g in douts:
            if g is not None and (g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16 or g.dtype != torch.float32):
                g = g.float().contiguous()
            gbuf.append(g)
        grads = [(ctx.sinks[t][:] if ctx.sinks[t] is not None else zeros_f32(tuple(sh), dev))
                 if (ctx.needs_input_grad[3 + t] and gbuf[spec[t][0]] is not None) else None for t, sh in enumerate(ctx.shapes)]
        jobs = [t for t in range(n) if grads[t] is not None]
        if jobs:
            m = len(jobs)
            ci, i64, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
            # a row-0 gradient another node left for one of these tables (assemble_tokens: the graph token's pe[0])
            extra, extra_job = None, 0
            for k, t in enumerate(jobs):
                pend = _ROW0_PENDING.pop(ctx.ptrs[t], None)
                if pend is not None:
                    extra, extra_job = pend, k
            check(_lib.lib().mobgt_embed_gather_multi(
                m, None, (vp * m)(*[grads[t].data_ptr() for t in jobs]), (vp * m)(*[ctx.idx[t].data_ptr() for t in jobs]),
                (i64 * m)(*[spec[t][3] for t in jobs]), (ci * m)(*[ctx.shapes[t][1] for t in jobs]),
                (ci * m)(*[spec[t][1] for t in jobs]), None, (vp * m)(*[gbuf[spec[t][0]].data_ptr() for t in jobs]),
                (i64 * m)(*[gbuf[spec[t][0]].stride(0) for t in jobs]), R, _IT[ctx.idx[0].dtype], 1, _p(extra), extra_job,
                _stream()), "mobgt_embed_gather_multi")
        return (None, None, None, *grads, *([None] * n))


def embed_gather_multi(jobs, out_widths):
    """jobs = [(table, index, out, coff, accum, padding_idx)]: table[index] -> columns [coff, coff + W) of output `out`
    (copied, or added when `accum`; accum = 2: this table's row is added to the PREVIOUS job's before that job stores --
    a constant partial table of the same shape, no gradient) -- all in one launch each way.
    -> list of [R, out_widths[o]] f32 tensors."""
    n = len(jobs)
    tabs = [j[0].float().contiguous() for j in jobs]
    idx = [j[1].contiguous().reshape(-1) for j in jobs]
    _require_cuda(*tabs, *idx)
    # (accum 2: FOLDED into the job in front of it -- same index, same width -- i.e. that job gathers the sum of the tables)
    spec = tuple((int(j[2]), int(j[3]), int(j[4]), -1 if j[5] is None else int(j[5])) for j in jobs)
    for k, sp in enumerate(spec):
        assert sp[2] != 2 or (k > 0 and tabs[k].shape[1] == tabs[k - 1].shape[1] and not tabs[k].requires_grad
                              and (k < 4 or any(s_[2] != 2 for s_ in spec[k - 3:k]))), "fold job (at most three per gather)"
    return list(_GatherMultiFn.apply(spec, tuple(out_widths), n, *tabs, *idx))


class _JoinColsFn(torch.autograd.Function):
    """`whole` [R, W] whose leading columns were filled IN PLACE by the op that produced `part` (a view of those columns):
    makes the result depend on both; backward hands `part` its columns of the gradient (a view, no launch)."""

    @staticmethod
    def forward(ctx, whole, part):
        ctx.w = part.shape[1]
        return whole.detach().view_as(whole)

    @staticmethod
    def backward(ctx, g):
        return g, g[:, :ctx.w]


def join_cols(whole, part):
    return _JoinColsFn.apply(whole, part)


# ------------------------------------------------------------------------------------------ loss
class _GradientTailLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, alpha, target_offset):
        logits = logits.float().contiguous()
        G, V = logits.shape
        dlogits = torch.empty_like(logits)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        check(_lib.lib().mobgt_gradient_tail_loss(_p(logits), _p(targets.long().contiguous()), int(target_offset), _p(dlogits),
                                                  _p(loss), G, V, float(alpha), _stream()), "mobgt_gradient_tail_loss")
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        if g.data_ptr() == unit_grad(g.device).data_ptr():      # d(loss)/d(loss) = 1 supplied by the trainer: no multiply
            return dlogits, None, None, None
        return dlogits * g, None, None, None


class _StockTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, din, dout, atom, indeg, outdeg, gtok, p, seed, seed_dev, salt, padding_idx):
        G, N = x.shape
        C = atom.shape[1]
        y = torch.empty(G, N + 1, C, dtype=torch.float32, device=atom.device)
        tok = (_p(x), _p(din), _p(dout), _IT[x.dtype], _IT[din.dtype], _p(atom), _p(indeg), _p(outdeg), _p(gtok), _p(y),
               G, N, C, atom.shape[0], indeg.shape[0], outdeg.shape[0], p, seed, _p(seed_dev), salt)
        hop, ni = take_front_jobs() if _FRONT_DEFER["on"] else (None, None)
        if ni is not None:                                       # (not a job of the stock model: launched alone)
            check(_lib.lib().mobgt_node_index(*ni[0], _stream()), "mobgt_node_index")
        jobs = []
        if _FRONT_DEFER["on"]:
            from .model import take_pending_pack
            jobs = take_pending_pack()
        if hop is not None or jobs:
            # the stock step's front as one grid (csrc/layer.hip stock_front_kernel): these rows, the hop table's forward the
            # model deferred (front_deferral) and the weight pack it deferred (refresh_shadows(defer_pack=True))
            import ctypes
            vp, ci = ctypes.c_void_p, ctypes.c_int
            for o in range(96, len(jobs), 96):                   # (more than one launch's worth of pack jobs: the rest alone)
                from .model import _launch_pack
                _launch_pack(jobs[o:o + 96])
            part = jobs[:96]
            nj = len(part)
            pack = ((vp * nj)(*[j[0].data_ptr() for j in part]), (vp * nj)(*[j[1].data_ptr() for j in part]), (ci * nj)(*[j[2] for j in part]),
                    (ci * nj)(*[j[3] for j in part]), (ci * nj)(*[j[4] for j in part])) if nj else (None, None, None, None, None)
            hargs = [1] + hop[0] if hop is not None else [0, None, None, None, 0, 0, 0, 0]
            check(_lib.lib().mobgt_stock_front_fwd(*tok, nj, *pack, *hargs, _stream()), "mobgt_stock_front_fwd")
        else:
            check(_lib.lib().mobgt_stock_tokens_fwd(*tok, _stream()), "mobgt_stock_tokens_fwd")
        ctx.idx = (x, din, dout)
        ctx.misc = (p, seed, seed_dev, salt, padding_idx, [t.shape for t in (atom, indeg, outdeg, gtok)])
        # (a table listed twice keeps separate buffers: autograd adds the results, which must not be one memory)
        ptrs = [t.data_ptr() for t in (atom, indeg, outdeg, gtok)]
        ctx.sinks = [grad_sink(t) if ptrs.count(t.data_ptr()) == 1 else None for t in (atom, indeg, outdeg, gtok)]
        return y

    @staticmethod
    def backward(ctx, dy):
        x, din, dout = ctx.idx
        p, seed, seed_dev, salt, padding_idx, shapes = ctx.misc
        G, N = x.shape
        dy = dy.contiguous()
        C = dy.shape[2]
        grads = [None if not need else (k[:] if (k is not None and tuple(k.shape) == tuple(sh)) else zeros_f32(tuple(sh), dy.device))
                 for need, k, sh in zip(ctx.needs_input_grad[3:7], ctx.sinks, shapes)]
        args = [_p(dy), _p(x), _p(din), _p(dout), _IT[x.dtype], _IT[din.dtype], _p(grads[0]), _p(grads[1]), _p(grads[2]),
                _p(grads[3]), G, N, C, shapes[0][0], shapes[1][0], shapes[2][0], int(padding_idx), p, seed, _p(seed_dev), salt]
        in_sinks = all((not need) or (k is not None and tuple(k.shape) == tuple(sh))
                       for need, k, sh in zip(ctx.needs_input_grad[3:7], ctx.sinks, shapes))
        if _WGRAD_DEFER["on"] and in_sinks and "stock_tok" not in _WGRAD_DEFER and os.environ.get("MOBGT_NO_STOCK_TAIL") != "1":
            # trainer's backward, every table gradient lands in its sink: nothing reads them before the optimizer, so the launch is
            # parked until the flush and shares ONE grid with the hop table's backward there (mobgt_stock_tail_bwd); what
            # autograd gets back are fresh views (AccumulateGrad clones a returned gradient something else still references)
            _WGRAD_DEFER["stock_tok"] = (args, (dy, x, din, dout, seed_dev, grads))
            return (None, None, None, *[g_[:] if g_ is not None else None for g_ in grads], None, None, None, None, None)
        check(_lib.lib().mobgt_stock_tokens_bwd(*args, _stream()), "mobgt_stock_tokens_bwd")
        return (None, None, None, *grads, None, None, None, None, None)


def stock_tokens_ok(x, atom, indeg, outdeg, gtok):
    return (x.is_cuda and x.dim() == 2 and x.dtype in _IT and atom.shape[1] % 4 == 0
            and all(t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in (atom, indeg, outdeg, gtok)))


def stock_tokens(x, in_degree, out_degree, atom, indeg, outdeg, graph_token, p, training, salt, padding_idx=0):
    """The stock variant's encoder input [G, N+1, C] (model.py:193-205): graph token row + atom / in-degree / out-degree rows
    summed, then input dropout -- one launch each way (csrc/layer.hip).  x [G,N] indices; in_degree, out_degree [G,N] indices of
    one dtype of their own (int64 / int32 / int16 each: no cast launch)."""
    _require_cuda(x, atom)
    assert in_degree.dtype == out_degree.dtype and in_degree.dtype in _IT and in_degree.dtype != torch.uint8 and x.dtype != torch.uint8
    if not training:
        p = 0.0
    seed, seed_dev = _DROPOUT_STATE["seed"], _DROPOUT_STATE["seed_dev"]
    if seed_dev is None and p > 0:
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    return _StockTokensFn.apply(x.contiguous(), in_degree.contiguous(), out_degree.contiguous(), atom, indeg, outdeg, graph_token,
                                float(p), int(seed), seed_dev, int(salt) & 0xFFFFFFFF, int(padding_idx))


class _TokenLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, weight, bias, eps):
        enc = enc.contiguous()
        G, T, C = enc.shape
        y = torch.empty(G, C, dtype=torch.float32, device=enc.device)
        stat = torch.empty(2, G, dtype=torch.float32, device=enc.device)
        check(_lib.lib().mobgt_token_ln_fwd(_p(enc), _p(weight), _p(bias), _p(y), _p(stat[0]), _p(stat[1]), G, T, C, float(eps),
                                            _stream()), "mobgt_token_ln_fwd")
        ctx.save_for_backward(enc, weight, stat)
        ctx.sinks = (grad_sink(weight), grad_sink(bias))
        return y

    @staticmethod
    def backward(ctx, dy):
        enc, weight, stat = ctx.saved_tensors
        G, T, C = enc.shape
        denc = torch.empty_like(enc)
        # (the affine gradients ACCUMULATE: into the parameters' sinks -- zeroed by the trainer's prologue -- or into zeros)
        dg, db = (k[:] if k is not None else zeros_f32((C,), enc.device) for k in ctx.sinks)
        check(_lib.lib().mobgt_token_ln_bwd(_p(dy.contiguous()), _p(enc), _p(stat[0]), _p(stat[1]), _p(weight), _p(denc), _p(dg), _p(db),
                                            G, T, C, _stream()), "mobgt_token_ln_bwd")
        return denc, dg, db, None


def token_layer_norm_ok(enc, weight):
    return (enc.is_cuda and enc.dim() == 3 and enc.dtype == torch.float32 and weight.dtype == torch.float32
            and enc.shape[2] <= 1024)


def token_layer_norm(enc, weight, bias, eps=1e-5):
    """LayerNorm of the graph-token rows enc[:, 0, :] of an encoder output [G,T,C] -> [G,C]: one launch each way (the backward
    writes the whole d(enc), zero outside the token rows)."""
    _require_cuda(enc, weight, bias)
    return _TokenLayerNormFn.apply(enc, weight.contiguous(), bias.contiguous(), eps)


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, ignore_index):
        logits = logits.float().contiguous()
        G, V = logits.shape
        dlogits = torch.empty_like(logits)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        check(_lib.lib().mobgt_cross_entropy(_p(logits), _p(targets.long().contiguous()), int(ignore_index), _p(dlogits), _p(loss), G, V,
                                             _stream()), "mobgt_cross_entropy")
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        if g.data_ptr() == unit_grad(g.device).data_ptr():
            return dlogits, None, None
        return dlogits * g, None, None


def cross_entropy_ok(logits, targets):
    return (logits.is_cuda and logits.dim() == 2 and 0 < logits.shape[0] <= 4095 and logits.shape[1] <= 10240
            and targets.numel() == logits.shape[0])


def cross_entropy(logits, targets, ignore_index=-100):
    """F.cross_entropy(logits, targets, ignore_index=ignore_index) (mean over the rows that count): value and gradient from one
    HIP kernel (csrc/layer.hip: cross_entropy_kernel)."""
    _require_cuda(logits, targets)
    return _CrossEntropyFn.apply(logits, targets.reshape(-1), ignore_index)


_UNIT = {}


def unit_grad(device):
    """A persistent scalar 1.0 per device.  `loss.backward(gradient=unit_grad(dev))` saves autograd's ones_like fill
    and lets the loss node skip its multiplication by one."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else (torch.cuda.current_device() if device.type == "cuda" else 0))
    t = _UNIT.get(key)
    if t is None:
        t = torch.ones((), dtype=torch.float32, device=device)
        _UNIT[key] = t
    return t


def gradient_tail_loss(logits, targets, alpha=0.25, target_offset=0):
    """model_fqandtoyo.py:545-550 (beta = k = 1): value and gradient from one HIP kernel.  Row g's class is
    targets[g] + target_offset (the training step's `y - 1` without a launch for it)."""
    _require_cuda(logits, targets)
    return _GradientTailLossFn.apply(logits, targets[: logits.shape[0]].reshape(-1), alpha, target_offset)


# ------------------------------------------------------------------------------- small linear layers
class _OutRef:
    """A destination buffer handed to an autograd Function as a plain Python object (not as a tensor input)."""

    def __init__(self, t):
        self.t = t.detach()


# ---- the encoder input's backward as ONE launch (csrc/tokbwd.hip) ------------------------------------------------------------------
# assemble_tokens' backward and the data gradients of FuseEmbeddings-4 / -2 are three autograd nodes in a row (tokens <- nf <-
# x4 = [f2 | cat] <- pt).  Inside the trainer's backward (weight gradients deferred to the grouped launch) they co-operate: the
# first two only PARK their work and return the buffers their results will live in, the third launches mobgt_token_bwd_chain,
# which fills all of them.  The consumers of those buffers (the weight-gradient group, the gathers' scatter) run later in
# stream order.  flush_deferred_wgrads() raises if a parked chain was never completed.
_TOKEN_CHAIN = {}           # "cur": what model.node_features registered for the forward pass that just ran
_TOKEN_PENDING = {}         # address of the gradient buffer the next node will receive -> parked work


def register_token_chain(nf, x4, w2_width, w4, slope4, w2, slope2):
    """nf [R, C] = leaky(x4 W4^T + b4), x4[:, :w2_width] = f2 = leaky(pt W2^T + b2) written in place (ops.join_cols): the
    chain whose backward mobgt_token_bwd_chain covers.  No-op for other widths."""
    _TOKEN_CHAIN.pop("cur", None)
    if os.environ.get("MOBGT_NO_TOKEN_BWD_CHAIN") == "1":
        return
    if (nf.is_cuda and nf.dtype == torch.float32 and x4.dtype == torch.float32 and nf.dim() == 2 and nf.is_contiguous()
            and x4.is_contiguous() and tuple(nf.shape) == tuple(x4.shape) and nf.shape[1] == 192 and w2_width == 160
            and tuple(w4.shape) == (192, 192) and tuple(w2.shape) == (160, 160) and w4.is_contiguous() and w2.is_contiguous()
            and w4.dtype == torch.float32 and w2.dtype == torch.float32):
        # (detached: a reference to nf / x4 themselves would keep this forward pass's autograd graph alive past its backward --
        #  and a live eager graph makes a later hipGraph capture_end crash)
        _TOKEN_CHAIN["cur"] = dict(nf_ptr=nf.data_ptr(), nf=nf.detach(), x4=x4.detach(), w4=w4.detach(), slope4=float(slope4),
                                   w2=w2.detach(), slope2=float(slope2))


def _token_park_linear(g, w, y):
    """_LinearSplitKFn.backward inside a parked chain: FuseEmbeddings-4 parks again and returns the buffer dx4 will be written
    to; FuseEmbeddings-2 launches the kernel and returns d_pt.  None: not part of a parked chain."""
    pend = _TOKEN_PENDING.get(g.data_ptr())
    if pend is None:
        return None
    ent = pend["ent"]
    if pend["stage"] == 1 and w.data_ptr() == ent["w4"].data_ptr() and y is not None and y.data_ptr() == ent["nf_ptr"]:
        del _TOKEN_PENDING[g.data_ptr()]
        dx4 = torch.empty_like(ent["x4"])
        pend.update(stage=2, dx4=dx4)
        _TOKEN_PENDING[dx4.data_ptr()] = pend
        return dx4
    if pend["stage"] == 2 and w.data_ptr() == ent["w2"].data_ptr() and y is not None and y.data_ptr() == ent["x4"].data_ptr():
        del _TOKEN_PENDING[g.data_ptr()]
        G, N, C, p_pos, p_in, seed, seed_dev, salts = pend["misc"]
        W2 = ent["w2"].shape[0]
        d_pt = torch.empty(G * N, W2, dtype=torch.float32, device=g.device)
        dx4 = pend["dx4"]
        check(_lib.lib().mobgt_token_bwd_chain(_p(pend["dout"]), _p(pend["real"]), _p(ent["nf"]), _p(ent["x4"]), ent["x4"].stride(0),
                                               _p(ent["w4"]), _p(ent["w2"]), _p(pend["d_nf"]), _p(pend["d_add"]), _p(dx4), dx4.stride(0),
                                               _p(d_pt), _p(pend["d_tok"]), G, N, C, W2, ent["slope4"], ent["slope2"], p_pos, p_in, seed,
                                               _p(seed_dev), salts[0], salts[1], salts[2], _stream()), "mobgt_token_bwd_chain")
        return d_pt
    return None


class _LinearSplitKFn(torch.autograd.Function):
    """y = x W^T + b for a few hundred rows: the weight gradient g^T x has K = rows and a tiny output, which a
    plain GEMM call maps onto one workgroup (measured 60-150 us in fp32); evaluate it split-K as a batched GEMM.
    `slope`: y = leaky_relu(x W^T + b, slope) (FuseEmbeddings, model_fqandtoyo.py:452-455) -- the activation in the GEMM's
    epilogue, its derivative applied to the incoming gradient while the two backward products load it."""

    @staticmethod
    def forward(ctx, x, w, b, bf16_wgrad, slope, out=None):
        ctx.bf16_wgrad = bf16_wgrad
        ctx.slope = slope
        if _small_linear(x, w) & 1:
            # (`out`: an _OutRef to a column slice of a wider buffer; the result is a fresh view of it, not an input)
            gat = _TOKEN_FWD["gather"] if _TOKEN_FWD["on"] else None
            stage = None
            if gat is not None and slope is not None and b is not None and x.dtype == torch.float32 and w.dtype == torch.float32:
                if (_TOKEN_FWD["f2"] is None and out is not None and x.data_ptr() == gat["outs"][0].data_ptr()
                        and out.t.data_ptr() == gat["outs"][1].data_ptr()):
                    stage = "f2"
                elif (_TOKEN_FWD["f2"] is not None and _TOKEN_FWD["f4"] is None and out is None
                        and x.data_ptr() == gat["outs"][1].data_ptr() and x.is_contiguous()):
                    stage = "f4"
            if stage is not None:            # recorded, not launched (token_fwd_deferral): the output buffer exists, its values come later
                y = out.t if out is not None else torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
                _TOKEN_FWD[stage] = dict(launch=lambda: small_gemm(x, w, b, True, leaky=slope, out=y), x=x, w=w, b=b, slope=slope, y=y)
            else:
                y = small_gemm(x, w, b, True, leaky=slope, out=out.t if out is not None else None)
            if out is not None:
                y = y.view_as(y)
        else:
            y = torch.addmm(b, x, w.t())
            if slope is not None:
                y = torch.nn.functional.leaky_relu(y, slope)
        ctx.save_for_backward(x, w, y if slope is not None else None)
        ctx.sinks = (grad_sink(w), grad_sink(b) if b is not None else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        k_w, k_b = ctx.sinks

        def dst(sink, shape):           # a fresh view of the parameter's gradient sink, or zeros
            return sink[:] if sink is not None else zeros_f32(shape, g.device)
        if y is None or y.is_contiguous():
            g = g.contiguous()                  # (else: a strided gradient goes straight into leaky_relu_backward below)
        R = x.shape[0]
        hip_wgrad = (ctx.bf16_wgrad and g.dtype == torch.float32 and x.dtype == torch.float32 and g.shape[1] % 2 == 0
                     and x.shape[1] % 2 == 0 and R <= 4096 and g.data_ptr() % 8 == 0 and x.data_ptr() % 8 == 0)
        if y is not None:
            mv = act_mask_values(ctx.slope, 0.0)
            # (the head's 16 rows: a few hundred rows and the data gradient's [K,N] operand walk make csrc/sgemm.hip slower
            # than the library GEMM + one elementwise launch -- measured 10.4 vs 9.6 us at R = 608)
            if (hip_wgrad and R <= 64 and small_gemm_ok(g, w) and y.is_contiguous() and y.data_ptr() % 8 == 0
                    and max(w.shape) <= 512):
                db = dst(k_b, (g.shape[1],))
                dw = linear_wgrad_masked(g, x, g_mask=y, mask_vals=mv, db=db, leaf=True, dw=dst(k_w, tuple(w.shape)))
                dx = _token_park_linear(g, w, y) if _WGRAD_DEFER["on"] else None
                return (dx if dx is not None else small_gemm(g, w, a_mask=(y, *mv))), dw, (db[:] if _WGRAD_DEFER["on"] else db), None, None, None
            if (_WGRAD_DEFER["on"] and hip_wgrad and g.stride(1) == 1 and y.stride() == g.stride() and g.data_ptr() % 8 == 0
                    and y.data_ptr() % 8 == 0 and g.stride(0) % 2 == 0 and small_gemm_ok(g, w) and R <= _SMALL_ROWS
                    and max(w.shape) <= 512):
                # trainer's backward: the weight gradient is a leaf and joins the step's one grouped launch; the data gradient
                # then applies the activation's derivative itself while it loads g (no masked copy of g exists)
                db = dst(k_b, (g.shape[1],))
                dw = linear_wgrad_masked(g, x, g_mask=y, mask_vals=mv, db=db, leaf=True, dw=dst(k_w, tuple(w.shape)))
                dx = _token_park_linear(g, w, y)               # (a parked encoder-input chain: see register_token_chain)
                return (dx if dx is not None else small_gemm(g, w, a_mask=(y, *mv))), dw, db[:], None, None, None
            if (hip_wgrad and g.stride(1) == 1 and y.stride() == g.stride() and g.data_ptr() % 8 == 0 and y.data_ptr() % 8 == 0
                    and g.stride(0) % 2 == 0):
                # the weight-gradient kernel applies the activation's derivative while it loads g, sums the bias gradient and
                # leaves the masked g for the (library) data-gradient GEMM: no elementwise launch
                gm = torch.empty_strided(g.shape, g.stride(), dtype=torch.float32, device=g.device)
                db = zeros_f32((g.shape[1],), g.device)
                dw = linear_wgrad_masked(g, x, g_mask=y, mask_vals=mv, db=db, g_out=gm)
                # (own GEMM: the library's pick for this 608 x 192 x 192 product varied between 5 and 13 us from run to run)
                dx = small_gemm(gm, w) if (small_gemm_ok(gm, w) and R <= _SMALL_ROWS and max(w.shape) <= 512) else gm @ w
                return dx, dw, db, None, None, None
            g = torch.ops.aten.leaky_relu_backward(g, y, ctx.slope, True).contiguous()      # (from the activation's result)
        if hip_wgrad:
            # bf16 configuration: weight AND bias gradient from the split-K MFMA kernel (operands rounded to bf16 while
            # loading); the library's split-K path took 27-31 us + a reduce for these 224-wide layers
            dw, db = linear_wgrad(g, x, with_bias=True, leaf=True, dw=dst(k_w, tuple(w.shape)), db=dst(k_b, (g.shape[1],)))
            return (small_gemm(g, w) if _small_linear(g, w) & 2 else g @ w), dw, db, None, None, None
        s = max((c for c in (16, 8, 4, 2) if R % c == 0 and R // c >= 16), default=1)
        if s > 1:
            dw = torch.bmm(g.view(s, R // s, -1).transpose(1, 2), x.view(s, R // s, -1)).sum(0)
        else:
            dw = g.t() @ x
        return g @ w, dw, colsum(g), None, None, None


_SMALL_LINEAR = 1
_SMALL_ROWS = 4096       # rows up to which csrc/sgemm.hip takes these layers


def _small_linear(x, w):
    """Bit mask (1: forward, 2: data gradient) -- FuseEmbeddings-sized Linear layers on csrc/sgemm.hip."""
    ok = (x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and x.shape[0] <= _SMALL_ROWS
          and max(w.shape) <= 512 and x.stride(1) == 1 and w.is_contiguous())
    if not ok:
        return 0
    return _SMALL_LINEAR if x.shape[0] <= 64 else (_SMALL_LINEAR & 1)     # data gradient: the head's 16 rows only


def linear_splitk(x, weight, bias, bf16_wgrad=False, slope=None, out=None):
    """F.linear(x, weight, bias) [-> leaky_relu(slope)] for a few hundred rows (see _LinearSplitKFn).  `out`: a 2-D f32
    view (unit column stride) that receives the result -- only honoured on the csrc/sgemm.hip path, check the return."""
    shape = x.shape
    x2 = x.reshape(-1, shape[-1]).contiguous()
    if out is not None and not (_small_linear(x2, weight) & 1 and out.shape == (x2.shape[0], weight.shape[0])):
        out = None
    y = _LinearSplitKFn.apply(x2, weight, bias, bf16_wgrad, slope, _OutRef(out) if out is not None else None)
    return y if out is not None else y.view(*shape[:-1], weight.shape[0])


# --------------------------------------------------------------------------------------- dropout
_DROPOUT_STATE = {"seed_dev": None, "seed": 0x5DEECE66D}


def set_dropout_state(seed_dev=None, seed=None):
    """Device seed scalar (advanced by the trainer, visible to captured graphs) and host seed for ops.dropout."""
    _DROPOUT_STATE["seed_dev"] = seed_dev
    if seed is not None:
        _DROPOUT_STATE["seed"] = int(seed)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, seed_dev, salt):
        x = x.contiguous()
        y = torch.empty_like(x)
        row = x.shape[-1] if x.dim() > 1 else x.numel()
        check(_lib.lib().mobgt_dropout(_p(x), _p(y), x.numel(), row, p, seed, _p(seed_dev), salt, _stream()), "mobgt_dropout")
        ctx.misc = (p, seed, seed_dev, salt, row)
        return y

    @staticmethod
    def backward(ctx, g):
        p, seed, seed_dev, salt, row = ctx.misc
        g = g.contiguous()
        dx = torch.empty_like(g)
        check(_lib.lib().mobgt_dropout(_p(g), _p(dx), g.numel(), row, p, seed, _p(seed_dev), salt, _stream()), "mobgt_dropout")
        return dx, None, None, None, None


def dropout(x, p, training, salt):
    """nn.Dropout replacement for the model-level sites; mask = f(seed state, salt, element index)."""
    if not training or p <= 0.0:
        return x
    _require_cuda(x)
    seed, seed_dev = _DROPOUT_STATE["seed"], _DROPOUT_STATE["seed_dev"]
    if seed_dev is None:
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    return _DropoutFn.apply(x.float(), float(p), int(seed), seed_dev, int(salt) & 0xFFFFFFFF)


class _BiasActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bias, slope, p_drop, seed, seed_dev, salt, yt=None):
        x = x.contiguous()
        R, C = x.shape
        y = torch.empty_like(x)
        check(_lib.lib().mobgt_bias_act_fwd_t(_p(x), _p(bias), _p(y), _p(yt.t if yt is not None else None),
                                              yt.t.stride(0) if yt is not None else 0, R, C, slope, p_drop, seed, _p(seed_dev), salt,
                                              _stream()), "mobgt_bias_act_fwd_t")
        ctx.save_for_backward(y)
        ctx.misc = (slope, p_drop, seed, seed_dev, salt, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        slope, p_drop, seed, seed_dev, salt, has_bias = ctx.misc
        R, C = y.shape
        dx = torch.empty_like(y)
        db = zeros_f32((C,), y.device) if has_bias else None
        check(_lib.lib().mobgt_bias_act_bwd(_p(dy.contiguous()), _p(y), _p(dx), _p(db), R, C, slope, p_drop, seed, _p(seed_dev),
                                            salt, _stream()), "mobgt_bias_act_bwd")
        return dx, db, None, None, None, None, None, None


def dropout_seed(p_drop):
    """(host seed, device step counter or None) for one dropout site -- what `dropout` / `bias_act` use: without a device
    counter every call draws a fresh host seed."""
    seed, seed_dev = _DROPOUT_STATE["seed"], _DROPOUT_STATE["seed_dev"]
    if seed_dev is None and p_drop > 0:
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    return int(seed), seed_dev


def bias_act(x, bias, slope, p_drop, training, salt, yt=None):
    """dropout(leaky_relu(x + bias, slope)) on a 2-D f32 tensor in one launch; the backward also yields bias.grad.
    `yt`: a bf16 [C, ld >= R] buffer that also receives the result transposed (modelGNN.xt_workspace)."""
    _require_cuda(x)
    if not training:
        p_drop = 0.0
    seed, seed_dev = _DROPOUT_STATE["seed"], _DROPOUT_STATE["seed_dev"]
    if seed_dev is None and p_drop > 0:
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    return _BiasActFn.apply(x.float(), bias, float(slope), float(p_drop), int(seed), seed_dev, int(salt) & 0xFFFFFFFF,
                            _OutRef(yt) if yt is not None else None)


class _HeadActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, w, b, eps, slope, p_drop, seed, seed_dev, salt):
        u, w, b = u.contiguous(), w.contiguous(), b.contiguous()
        R, C = u.shape
        out = torch.empty_like(u)
        stats = torch.empty(2, R, dtype=torch.float32, device=u.device)
        check(_lib.lib().mobgt_head_act_fwd(_p(u), _p(w), _p(b), _p(out), _p(stats[0]), _p(stats[1]), R, C, eps, slope, p_drop,
                                            seed, _p(seed_dev), salt, _stream()), "mobgt_head_act_fwd")
        ctx.save_for_backward(u, w, b, stats)
        ctx.misc = (eps, slope, p_drop, seed, seed_dev, salt)
        return out

    @staticmethod
    def backward(ctx, dout):
        u, w, b, stats = ctx.saved_tensors
        eps, slope, p_drop, seed, seed_dev, salt = ctx.misc
        R, C = u.shape
        du = torch.empty_like(u)
        dg, db = zeros_f32((C,), u.device), zeros_f32((C,), u.device)
        check(_lib.lib().mobgt_head_act_bwd(_p(dout.contiguous()), _p(u), _p(w), _p(b), _p(stats[0]), _p(stats[1]), _p(du),
                                            _p(dg), _p(db), R, C, eps, slope, p_drop, seed, _p(seed_dev), salt, _stream()),
              "mobgt_head_act_bwd")
        return du, dg, db, None, None, None, None, None, None


def head_act(u, ln_weight, ln_bias, eps, slope, p_drop, training, salt):
    """dropout(ELU(LayerNorm(LeakyReLU(u)))) on a few rows (the classifier head's activation chain) in one launch."""
    _require_cuda(u, ln_weight, ln_bias)
    if not training:
        p_drop = 0.0
    seed, seed_dev = _DROPOUT_STATE["seed"], _DROPOUT_STATE["seed_dev"]
    if seed_dev is None and p_drop > 0:
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    return _HeadActFn.apply(u.float(), ln_weight, ln_bias, float(eps), float(slope), float(p_drop), int(seed), seed_dev,
                            int(salt) & 0xFFFFFFFF)


class _HeadChainFn(torch.autograd.Function):
    """csrc/head.hip: [token row | user row] -> FuseEmbeddings' Linear -> LeakyReLU -> LayerNorm -> ELU -> dropout, one launch
    each way (head_input + linear_splitk + head_act as separate launches: 16 + 17 us of the S-FSQ step for 16 rows)."""

    @staticmethod
    def forward(ctx, enc, table, user, offset, w3, b3, ln_w, ln_b, eps, slope, p_drop, seed, seed_dev, salt, bf16_wgrad):
        G, T, C = enc.shape
        U = table.shape[1]
        W = C + U
        dev = enc.device
        x3, u3, out = (torch.empty(G, W, dtype=torch.float32, device=dev) for _ in range(3))
        stats = torch.empty(2, G, dtype=torch.float32, device=dev)
        check(_lib.lib().mobgt_head_chain_fwd(_p(enc), _p(user), _IT[user.dtype], offset, _p(table), table.shape[0], _p(w3), _p(b3),
                                              _p(ln_w), _p(ln_b), _p(x3), _p(u3), _p(out), _p(stats[0]), _p(stats[1]), G, T, C, U,
                                              eps, slope, p_drop, seed, _p(seed_dev), salt, _p(_head_chain_ws(dev)), _stream()),
              "mobgt_head_chain_fwd")
        ctx.save_for_backward(x3, u3, stats, w3, ln_w, ln_b)
        ctx.user = user
        ctx.misc = (G, T, C, U, offset, tuple(table.shape), eps, slope, p_drop, seed, seed_dev, salt, bf16_wgrad)
        ctx.sink = grad_sink(table)
        ctx.psinks = (grad_sink(w3), grad_sink(b3), grad_sink(ln_w), grad_sink(ln_b))
        return out

    @staticmethod
    def backward(ctx, dout):
        x3, u3, stats, w3, ln_w, ln_b = ctx.saved_tensors
        G, T, C, U, offset, tshape, eps, slope, p_drop, seed, seed_dev, salt, bf16_wgrad = ctx.misc
        dev = dout.device
        dout = dout.contiguous()
        du3 = torch.empty_like(u3)
        denc = torch.empty(G, T, C, dtype=torch.float32, device=dev)
        dtable = ctx.sink[:] if ctx.sink is not None else zeros_f32(tshape, dev)
        k_w3, k_b3, k_g, k_beta = ctx.psinks
        dg = k_g[:] if k_g is not None else zeros_f32((C + U,), dev)
        dbeta = k_beta[:] if k_beta is not None else zeros_f32((C + U,), dev)
        check(_lib.lib().mobgt_head_chain_bwd(_p(dout), _p(u3), _p(stats[0]), _p(stats[1]), _p(ctx.user), _IT[ctx.user.dtype], offset,
                                              tshape[0], _p(w3), _p(ln_w), _p(ln_b), _p(du3), _p(denc), _p(dtable), _p(dg), _p(dbeta),
                                              G, T, C, U, eps, slope, p_drop, seed, _p(seed_dev), salt, _stream()),
              "mobgt_head_chain_bwd")
        if bf16_wgrad:                                                     # (operands rounded to bf16 while loading)
            dw, db = linear_wgrad(du3, x3, with_bias=True, leaf=True, dw=k_w3[:] if k_w3 is not None else None,
                                  db=k_b3[:] if k_b3 is not None else None)
        else:
            dw, db = du3.t() @ x3, colsum(du3)
        return denc, dtable, None, None, dw, db, dg, dbeta, None, None, None, None, None, None, None


_HEAD_WS = {}


def _head_chain_ws(dev):
    """Exchange area of csrc/head.hip's forward (include/mobgt_hip.h: mobgt_head_chain_ws_bytes): one per device, zeroed once,
    for one stream at a time; it must exist before a graph capture starts (an eager warm-up step creates it)."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ws = _HEAD_WS.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the head kernel's workspace must be allocated before graph capture: run one eager step first")
        ws = _HEAD_WS[key] = torch.zeros(int(_lib.lib().mobgt_head_chain_ws_bytes()), dtype=torch.uint8, device=dev)
    return ws


def head_chain_ok(enc, table, user, w3):
    W = enc.shape[-1] + table.shape[1]
    return (enc.is_cuda and enc.dim() == 3 and enc.dtype == torch.float32 and table.dtype == torch.float32 and table.is_contiguous()
            and W in (320, 384) and tuple(w3.shape) == (W, W) and w3.dtype == torch.float32 and w3.is_contiguous()
            and user.dtype in (torch.int64, torch.int32) and user.numel() == enc.shape[0] and enc.shape[0] <= 160
            and enc.shape[-1] % 16 == 0 and table.shape[1] % 16 == 0 and not SAFE_FORMS[0])


def head_chain(enc, user_table, user, user_offset, w3, b3, ln_weight, ln_bias, eps, slope, p_drop, training, salt, bf16_wgrad=False):
    """tok [G, C+U] = dropout(ELU(LayerNorm(LeakyReLU(Linear([enc[:, 0] | user_table[user + user_offset]]))))) -- the classifier
    head in front of out_proj (model_fqandtoyo.py:1239-1240, 1353-1364) in one launch each way.  Where head_chain_ok says no (and under
    ops.SAFE_FORMS) callers fall back to head_input + linear_splitk + head_act."""
    _require_cuda(enc, user_table, user, w3)
    if not training:
        p_drop = 0.0
    seed, seed_dev = _DROPOUT_STATE["seed"], _DROPOUT_STATE["seed_dev"]
    if seed_dev is None and p_drop > 0:
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    return _HeadChainFn.apply(enc.contiguous(), user_table, user.reshape(-1).contiguous(), int(user_offset), w3, b3.contiguous(),
                              ln_weight.contiguous(), ln_bias.contiguous(), float(eps), float(slope), float(p_drop), int(seed),
                              seed_dev, int(salt) & 0xFFFFFFFF, bool(bf16_wgrad))


class _HeadInputFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, table, user, offset):
        G, T, C = enc.shape
        U = table.shape[1]
        x3 = torch.empty(G, C + U, dtype=torch.float32, device=enc.device)
        check(_lib.lib().mobgt_head_input_fwd(_p(enc), _p(user), _IT[user.dtype], offset, _p(table), table.shape[0], _p(x3),
                                              G, T, C, U, _stream()), "mobgt_head_input_fwd")
        ctx.user, ctx.misc = user, (G, T, C, U, offset, table.shape)
        ctx.sink = grad_sink(table)
        return x3

    @staticmethod
    def backward(ctx, dx3):
        G, T, C, U, offset, tshape = ctx.misc
        dx3 = dx3.contiguous()
        denc = torch.empty(G, T, C, dtype=torch.float32, device=dx3.device)
        dtable = ctx.sink[:] if ctx.sink is not None else zeros_f32(tuple(tshape), dx3.device)
        check(_lib.lib().mobgt_head_input_bwd(_p(dx3), _p(ctx.user), _IT[ctx.user.dtype], offset, _p(denc), _p(dtable),
                                              tshape[0], G, T, C, U, _stream()), "mobgt_head_input_bwd")
        return denc, dtable, None, None


def head_input(enc, user_table, user, user_offset=0):
    """[G, C+U] input of the classifier head: the graph-token row of the encoder output next to the user's embedding row
    `user_table[user + user_offset]` (model_fqandtoyo.py:1239-1240, 1353-1358) -- one launch each way."""
    _require_cuda(enc, user_table, user)
    assert enc.dtype == torch.float32 and user_table.dtype == torch.float32 and user.dtype in (torch.int64, torch.int32)
    assert user_table.is_contiguous()
    return _HeadInputFn.apply(enc.contiguous(), user_table, user.reshape(-1).contiguous(), int(user_offset))


def _token_fwd_fused(nf, real, add, token, pe0, out, out16, qkv_w, qkv, G, N, C, p_pos, p_in, seed, seed_dev, salts):
    """The recorded gather + FuseEmbeddings-2 / -4 (token_fwd_deferral) together with this token assembly + QKV projection as
    ONE launch -> True; anything else: the recorded launches are issued separately -> False (the caller launches its own)."""
    gat, f2, f4 = _TOKEN_FWD["gather"], _TOKEN_FWD["f2"], _TOKEN_FWD["f4"]
    ok = (gat is not None and f2 is not None and f4 is not None and C == 192 and gat["R"] == G * N and len(gat["outs"]) == 3
          and nf.data_ptr() == f4["y"].data_ptr() and add.data_ptr() == gat["outs"][2].data_ptr()
          and tuple(f2["w"].shape) == (160, 160) and tuple(f4["w"].shape) == (192, 192)
          and gat["outs"][0].shape[1] == 160 and gat["outs"][1].shape[1] == 192 and gat["outs"][2].shape[1] == 192
          and all(t.is_contiguous() for t in (f2["w"], f2["b"], f4["w"], f4["b"], *gat["outs"]))
          and (pe0.dim() == 1 or pe0.shape[0] >= 1))
    if not ok:
        flush_token_fwd()
        return False
    tables, idx, spec = gat["tables"], gat["idx"], gat["spec"]
    n = len(tables)
    ci = ctypes.c_int
    rc = (_lib.lib().mobgt_token_fwd_chain(
        n, _ptr_array(tables), _ptr_array(idx), (ci * n)(*[t.shape[1] for t in tables]), (ci * n)(*[sp[1] for sp in spec]),
        (ci * n)(*[int(sp[2]) for sp in spec]), (ci * n)(*[sp[0] for sp in spec]), _IT[idx[0].dtype],
        _p(gat["outs"][0]), _p(gat["outs"][1]), _p(gat["outs"][2]), _p(f4["y"]), 160,
        _p(f2["w"]), _p(f2["b"]), float(f2["slope"]), _p(f4["w"]), _p(f4["b"]), float(f4["slope"]),
        _p(real), _p(token), _p(pe0), _p(out), _p(out16), _p(qkv_w[0].t), _p(qkv_w[1].t), _p(qkv), G, N, C, p_pos, p_in, seed,
        _p(seed_dev), salts[0], salts[1], salts[2], _stream()))
    if rc == -1:                          # MOBGT_EBADDIM: a job list the kernel's gather stage is not laid out for (nothing was launched)
        flush_token_fwd()
        return False
    check(rc, "mobgt_token_fwd_chain")
    for k in ("gather", "f2", "f4"):
        _TOKEN_FWD[k] = None
    _TOKEN_FWD["fused_calls"] += 1
    return True


class _AssembleTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nf, real, add, token, pe0, p_pos, p_in, seed, seed_dev, salts, side, row0_via_gather=False, qkv_w=None, tok_sink=None):
        ctx.tok_sink = tok_sink
        ctx.pe_ptr = pe0.data_ptr() if (row0_via_gather and pe0.dim() == 2 and pe0.shape[0] > 1) else None
        ctx.nf_ptr = nf.data_ptr()
        G, N, C = nf.shape
        nf, add, real = nf.contiguous(), add.contiguous(), real.contiguous()
        shapes = (token.shape, pe0.shape)
        token = token.reshape(-1).contiguous()
        # pe0: row 0 of the positional table, or the whole [L, C] table (then its gradient is one [L, C] tensor with
        # row 0 filled in -- no select_backward fill + copy)
        pe0 = pe0.contiguous() if pe0.dim() == 2 and pe0.shape[0] > 1 else pe0.reshape(-1).contiguous()
        out = torch.empty(G, N + 1, C, dtype=torch.float32, device=nf.device)
        out16 = torch.empty(G, N + 1, C, dtype=torch.bfloat16, device=nf.device) if side is not None else None
        if qkv_w is not None and out16 is not None and C in (192, 256):
            # ... and the first encoder layer's QKV projection from the same launch (csrc/chain.hip): qkv_w = _OutRefs of the
            # packed [3C, C] weight and the bf16 bias
            qkv = torch.empty(G * (N + 1), 3 * C, dtype=torch.bfloat16, device=nf.device)
            if not _token_fwd_fused(nf, real, add, token, pe0, out, out16, qkv_w, qkv, G, N, C, p_pos, p_in, seed, seed_dev, salts):
                check(_lib.lib().mobgt_assemble_tokens_qkv(_p(nf), _p(real), _p(add), _p(token), _p(pe0), _p(out), _p(out16),
                                                           _p(qkv_w[0].t), _p(qkv_w[1].t), _p(qkv), G, N, C, p_pos, p_in, seed,
                                                           _p(seed_dev), salts[0], salts[1], salts[2], _stream()),
                      "mobgt_assemble_tokens_qkv")
            side.append(out16)
            side.append(qkv)
        else:
            flush_token_fwd()
            check(_lib.lib().mobgt_assemble_tokens_fwd(_p(nf), _p(real), _p(add), _p(token), _p(pe0), _p(out), _p(out16), G, N, C,
                                                       p_pos, p_in, seed, _p(seed_dev), salts[0], salts[1], salts[2], _stream()),
                  "mobgt_assemble_tokens_fwd")
            if side is not None:
                side.append(out16)
        ctx.save_for_backward(real)
        ctx.misc = (G, N, C, p_pos, p_in, seed, seed_dev, salts, shapes)
        return out

    @staticmethod
    def backward(ctx, dout):
        (real,) = ctx.saved_tensors
        G, N, C, p_pos, p_in, seed, seed_dev, salts, (tshape, pshape) = ctx.misc
        dout = dout.contiguous()
        d_nf = torch.empty(G, N, C, dtype=torch.float32, device=dout.device)
        d_add = torch.empty(G, N, C, dtype=torch.float32, device=dout.device)
        k_tok = ctx.tok_sink
        if ctx.pe_ptr is not None:
            d_tok = k_tok.view(-1)[:] if (k_tok is not None and k_tok.numel() == C) else zeros_f32((C,), dout.device)
            d_pe = None
        elif len(pshape) == 2 and pshape[0] > 1:
            d_pe = zeros_f32(tuple(pshape), dout.device)
            d_tok = d_pe[0]                       # d(token) = d(pe[0]): the same column sums
        else:
            d_tok = zeros_f32((C,), dout.device)
            d_pe = d_tok.view(pshape)
        ent = _TOKEN_CHAIN.get("cur")
        if (ent is not None and ent["nf_ptr"] == ctx.nf_ptr and _WGRAD_DEFER["on"] and C == ent["nf"].shape[1]
                and G * N == ent["nf"].shape[0] and not os.environ.get("MOBGT_NO_TOKEN_BWD_CHAIN")):
            # park: FuseEmbeddings-2's backward, two autograd nodes further down, launches the one kernel that fills d_nf / d_add
            _TOKEN_PENDING[d_nf.data_ptr()] = dict(stage=1, ent=ent, dout=dout, real=real, d_nf=d_nf, d_add=d_add, d_tok=d_tok,
                                                   misc=(G, N, C, p_pos, p_in, seed, seed_dev, salts))
        else:
            check(_lib.lib().mobgt_assemble_tokens_bwd(_p(dout), _p(real), _p(d_nf), _p(d_add), _p(d_tok), G, N, C, p_pos, p_in, seed,
                                                       _p(seed_dev), salts[0], salts[1], salts[2], _stream()),
                  "mobgt_assemble_tokens_bwd")
        if ctx.pe_ptr is not None:
            # the positional table's other consumer (the gather of pe[1..n]) adds this row-0 share inside ITS scatter launch:
            # one gradient producer for the table, no [L, C] zero table here and no table-sized add after
            _ROW0_PENDING[ctx.pe_ptr] = d_tok
            return d_nf, None, d_add, d_tok.view(tshape), None, None, None, None, None, None, None, None, None, None
        return d_nf, None, d_add, d_tok.view(tshape), d_pe, None, None, None, None, None, None, None, None, None


def assemble_tokens(nf, real, add, token, pe0, p_pos, p_in, training, salts=(0x1001, 0x1002, 0x1003), bf16_copy=False,
                    pe_row0_via_gather=False, first_qkv=None):
    """[G,N+1,C] encoder input: graph token row (+ pe[0]) and the node features (* real + add), each through the
    positional dropout and then the input dropout -- one launch forward, one backward (see mobgt_assemble_tokens_fwd).
    `token` is [C]-sized; `pe0` is pe[0] or the whole positional table [L, C] (row 0 is used); the gradient of both is
    the same per-column sum over graphs.  `bf16_copy`: the kernel also writes the result in bf16 and hangs it on the
    returned tensor as `_mobgt_act` (what the first fused encoder layer feeds its QKV GEMM)."""
    _require_cuda(nf, add, token, pe0)
    if not training:
        p_pos = p_in = 0.0
    seed, seed_dev = _DROPOUT_STATE["seed"], _DROPOUT_STATE["seed_dev"]
    if seed_dev is None and (p_pos > 0 or p_in > 0):
        seed = (seed + int(torch.randint(0, 2 ** 31 - 1, (1,)).item())) & 0x7FFFFFFFFFFFFFFF
    side = [] if bf16_copy else None
    # `pe_row0_via_gather`: `pe0` is the whole positional table AND its rows 1.. reach `add` through ops.embed_gather_multi in
    # the same autograd graph: that node's backward then adds the token row's gradient to pe[0] (see _ROW0_PENDING)
    out = _AssembleTokensFn.apply(nf.float(), real.float(), add.float(), token.float(), pe0.float(), float(p_pos),
                                  float(p_in), int(seed), seed_dev, tuple(int(s) & 0xFFFFFFFF for s in salts), side,
                                  bool(pe_row0_via_gather and pe0.requires_grad and torch.is_grad_enabled()),
                                  (_OutRef(first_qkv[0]), _OutRef(first_qkv[1])) if (first_qkv is not None and bf16_copy) else None,
                                  grad_sink(token) if token.is_contiguous() else None)
    if side:
        out._mobgt_act = side[0]          # bf16 copy for the first fused layer's QKV GEMM (no cast launch)
        if len(side) > 1:
            out._mobgt_qkv = side[1]      # `first_qkv` = (packed wqkv, bqkv) of that layer: its QKV projection, from this launch
    return out


# ------------------------------------------------------------------- small f32 GEMMs (GCN / fuse / head)
def small_gemm(a, b, bias=None, b_is_nk=False, out=None, out_dtype=torch.float32, leaky=None, drop=None, a_mask=None, ct=None,
               k_b=None):
    """a [M,K] @ (b.T if b_is_nk else b) (+ bias) -> f32 [M,N] on csrc/sgemm.hip (one wave per 16-row tile, f32 MFMA).
    No autograd.  Operands may be row-strided views (unit column stride).
    `leaky` (slope): LeakyReLU on the way out; `drop` = (p, seed, seed_dev, salt): then dropout (mobgt_bias_act_fwd's mask).
    `a_mask` = (y, pos, neg, zero): a is multiplied elementwise by m(y) while it is loaded (y: a's shape and row stride).
    `ct` = (buffer bf16 [N, ld], row_scale f32 [M] or None, only): the result (times row_scale) also as bf16 TRANSPOSED into
    `buffer` (modelGNN.xt_workspace: the bitmask adjacency product's operand); `only`: no [M,N] result at all (returns None).
    `k_b`: b ([K,N] form) has only k_b <= a.shape[1] rows: a's trailing columns are zero padding."""
    _require_cuda(a, b)
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2
    assert a.stride(1) == 1 and b.stride(1) == 1
    pre = _FRONT_SGEMM.get("done")
    if pre is not None:
        # this very product rode in the bias assembly's launch (front_small_gemm): its result exists
        if (pre["key"] == (a.data_ptr(), b.data_ptr(), bias.data_ptr() if bias is not None else 0, leaky, k_b,
                           ct[0].data_ptr() if ct else 0) and drop is None and a_mask is None and out is None and not b_is_nk):
            del _FRONT_SGEMM["done"]
            return pre["c"]
    M, K = a.shape
    N = b.shape[0] if b_is_nk else b.shape[1]
    assert (b.shape[1] if b_is_nk else b.shape[0]) == (K if k_b is None else k_b) and (k_b is None or (not b_is_nk and k_b <= K))
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    only_t = ct is not None and ct[2]
    c = None if only_t else (out if out is not None else torch.empty(M, N, dtype=out_dtype, device=a.device))
    if leaky is None and a_mask is None and ct is None and k_b is None:
        check(_lib.lib().mobgt_small_gemm_f32(_p(a), a.stride(0), _p(b), b.stride(0), int(b_is_nk), _p(bias), _p(c), c.stride(0),
                                              _DT[c.dtype], M, N, K, _stream()), "mobgt_small_gemm_f32")
        return c
    y, pos, neg, zer = a_mask if a_mask is not None else (None, 1.0, 1.0, 1.0)
    if y is not None:
        assert y.dtype == torch.float32 and y.shape == a.shape and y.stride(1) == 1 and y.stride(0) == a.stride(0)
    p_drop, seed, seed_dev, salt = drop if drop is not None else (0.0, 0, None, 0)
    check(_lib.lib().mobgt_small_gemm_f32_act(_p(a), a.stride(0), _p(y), float(pos), float(neg), float(zer), _p(b), b.stride(0),
                                              int(b_is_nk), _p(bias), int(leaky is not None), float(leaky or 0.0), float(p_drop),
                                              int(seed), _p(seed_dev), int(salt) & 0xFFFFFFFF, _p(c), c.stride(0) if c is not None else N,
                                              _DT[c.dtype] if c is not None else F32, _p(ct[0] if ct else None),
                                              ct[0].stride(0) if ct else 0, _p(ct[1] if ct else None), M, N, K, int(k_b or 0),
                                              _stream()),
          "mobgt_small_gemm_f32_act")
    return c


# ---- a small GEMM riding in the bias assembly's launch (round 4) --------------------------------------------------------------------
_FRONT_SGEMM = {}


def front_small_gemm(a, b, bias, leaky, k_b, ct):
    """Leave  leaky_relu(a @ b[:k_b] + bias)  (+ transposed bf16 copy into ct) for the NEXT short-batch mobgt_build_bias launch
    (mobgt_front_sgemm_job); `front_small_gemm_flush()` behind that launch turns it into the result `small_gemm` hands out when it
    is called with these very arguments -- or launches it alone if no launch took it.  False: not a shape of that form."""
    _FRONT_SGEMM.clear()
    if (os.environ.get("MOBGT_NO_L0_RIDE") == "1" or not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
            and a.is_contiguous() and b.is_contiguous() and bias is not None and bias.is_contiguous())):
        return False
    M, K = a.shape
    N = b.shape[1]
    c = torch.empty(M, N, dtype=torch.float32, device=a.device)
    rc = _lib.lib().mobgt_front_sgemm_job(_p(a), a.stride(0), _p(b), b.stride(0), _p(bias), 1, float(leaky), _p(c), c.stride(0),
                                          _p(ct), ct.stride(0) if ct is not None else 0, M, N, K, int(k_b))
    if rc != 0:
        return False
    _FRONT_SGEMM["job"] = dict(key=(a.data_ptr(), b.data_ptr(), bias.data_ptr(), leaky, k_b, ct.data_ptr() if ct is not None else 0),
                               c=c, args=(a, b, bias, leaky, k_b, ct))
    return True


def front_small_gemm_flush():
    job = _FRONT_SGEMM.pop("job", None)
    if job is None:
        return
    if _lib.lib().mobgt_front_sgemm_pending():             # no launch took it along: drop the job, launch the product alone
        check(_lib.lib().mobgt_front_sgemm_job(None, 0, None, 0, None, 0, 0.0, None, 0, None, 0, 0, 0, 0, 0), "mobgt_front_sgemm_job")
        a, b, bias, leaky, k_b, ct = job["args"]
        job["c"] = small_gemm(a, b, bias, leaky=leaky, k_b=k_b, ct=(ct, None, False) if ct is not None else None)
    _FRONT_SGEMM["done"] = job


def front_small_gemm_drop():
    """Forget a job that is still waiting for a launch (an exception between front_small_gemm and its flush) and an unclaimed result."""
    if _FRONT_SGEMM.pop("job", None) is not None:
        check(_lib.lib().mobgt_front_sgemm_job(None, 0, None, 0, None, 0, 0.0, None, 0, None, 0, 0, 0, 0, 0), "mobgt_front_sgemm_job")
    _FRONT_SGEMM.pop("done", None)


def act_mask_values(slope, p_drop):
    """(pos, neg, zero) of m(y) = d dropout(leaky_relu(u)) / du read off the OUTPUT y: a kept positive is scaled by 1/keep, a
    kept negative by slope/keep; y == 0 is a dropped element (gradient 0) when dropout is on, else LeakyReLU'(0) = slope."""
    if p_drop and p_drop > 0.0:
        thr = int(p_drop * 65536.0 + 0.5)
        ik = 1.0 / (1.0 - thr / 65536.0)
        return ik, slope * ik, 0.0
    return 1.0, slope, slope


# ---- leaf weight gradients issued together (round 3) ----------------------------------------------------------------------------
# Nothing but the optimizer reads a weight gradient.  While `wgrad_deferral(True)` is in force (train.TrainStep switches it on
# around a backward pass and flushes before it gathers the gradients), `linear_wgrad(..., leaf=True)` /
# `linear_wgrad_masked(..., leaf=True)` only RECORD their problem -- operands and destinations are kept alive by the record --
# and `flush_deferred_wgrads()` issues all of them as ONE launch (mobgt_linear_wgrad_multi).  What a caller gets back is a
# fresh VIEW of the (zero-initialised) destination: autograd's AccumulateGrad steals a returned gradient only if nothing else
# references that tensor object, and the record holds the base -- the kernel's late writes land in the tensor the parameter
# ends up with (the rule fused_layer.py follows for its parked tails).  Off by default: eager callers get the launch at once.
_WGRAD_DEFER = {"on": False, "items": []}


def wgrad_deferral(on):
    """Switch the recording on / off.  Switching off DROPS anything still recorded: the trainer flushes explicitly after a
    successful backward pass, so leftovers exist only when that pass raised -- their buffers belong to a dead graph."""
    _WGRAD_DEFER["items"] = []
    for k in ("hop", "hop_wide", "stock_tok", "psum"):
        _WGRAD_DEFER.pop(k, None)
    _WGRAD_DEFER["on"] = bool(on)


def defer_partial_sum(part, dst):
    """Park `dst = part.sum(0)` ([s, M, N] f32 split-K partial products of a library batched GEMM; dst: the gradient's sink)
    until flush_deferred_wgrads, which issues all of a step's as ONE launch (mobgt_partial_sum_multi).  -> a fresh view of dst,
    or None when the sum cannot be parked (no recording, layouts): the caller then sums now."""
    if not (_WGRAD_DEFER["on"] and dst is not None and part.dtype == torch.float32 and dst.dtype == torch.float32
            and part.is_contiguous() and dst.is_contiguous() and part.dim() == 3 and dst.numel() == part[0].numel()
            and dst.numel() % 4 == 0 and part.data_ptr() % 16 == 0 and dst.data_ptr() % 16 == 0
            and os.environ.get("MOBGT_NO_PSUM_DEFER") != "1"):
        return None
    _WGRAD_DEFER.setdefault("psum", []).append((part, dst))
    return dst[:]


def layer_wgrad_big(items, R):
    """The weight gradients of one encoder layer past 4 096 rows as ONE launch (csrc/wgradbig.hip, mobgt_layer_wgrad_big):
    items = [(g [R, M] bf16, x [R, N] bf16, db [M] f32 zero-filled or None, sink or None)], at most four -> [dW [M, N] f32].
    The launch leaves split-K partial products [S, M, N]; inside a train step their sums are parked for the step's one
    reduction launch into the gradients' sinks (defer_partial_sum), else they are summed here."""
    lib = _lib.lib()
    n = len(items)
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    tiles = sum(lib.mobgt_layer_wgrad_big_tiles(g.shape[1], x.shape[1]) for g, x, _, _ in items)
    S = lib.mobgt_layer_wgrad_big_splits(R, tiles)
    dev = items[0][0].device
    parts = [torch.empty(S, g.shape[1], x.shape[1], dtype=torch.float32, device=dev) for g, x, _, _ in items]
    check(lib.mobgt_layer_wgrad_big(n, (vp * n)(*[t[0].data_ptr() for t in items]), (i64 * n)(*[t[0].stride(0) for t in items]),
                                    (vp * n)(*[t[1].data_ptr() for t in items]), (i64 * n)(*[t[1].stride(0) for t in items]),
                                    (vp * n)(*[q.data_ptr() for q in parts]),
                                    (vp * n)(*[(t[2].data_ptr() if t[2] is not None else None) for t in items]),
                                    (ci * n)(*[t[0].shape[1] for t in items]), (ci * n)(*[t[1].shape[1] for t in items]), R, S,
                                    _stream()), "mobgt_layer_wgrad_big")
    outs = []
    for part, (_, _, _, sink) in zip(parts, items):
        parked = defer_partial_sum(part, sink) if sink is not None else None
        outs.append(parked if parked is not None else part.sum(0))
    return outs


def layer_wgrad_big_ok(items):
    return (1 <= len(items) <= 4 and all(
        g.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and g.dim() == 2 and x.dim() == 2 and g.shape[0] == x.shape[0]
        and g.stride(1) == 1 and x.stride(1) == 1 and g.shape[1] % 8 == 0 and x.shape[1] % 8 == 0 and g.stride(0) % 8 == 0
        and x.stride(0) % 8 == 0 and g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
        and g.shape[0] * g.stride(0) * 2 < (1 << 31) and x.shape[0] * x.stride(0) * 2 < (1 << 31) for g, x, _, _ in items))


def _wgrad_defer(g, x, g_mask, x_mask, mask_vals, dw, db, db_of_x):
    _WGRAD_DEFER["items"].append((g, x, g_mask, x_mask, tuple(float(v) for v in mask_vals), dw, db, bool(db_of_x)))
    return dw[:]


def step_state_leftovers():
    """What the step's hand-over registries still hold -- {} when a forward + backward pass has consumed everything it parked.
    The registries (the bias tables' backward job, the front-of-step jobs, the parked encoder-input chain, deferred weight
    gradients, the fused layers' parked tails, the deferred weight pack) are process-wide by design: a job parked by the forward
    pass on the caller's thread is taken by a launch the autograd engine issues from ITS thread, so they cannot be thread-local;
    ONE step is in flight per process.  The trainer asserts this is empty behind every step (`TrainStep._fwd_bwd`):
    a job left behind -- an exception on the way, a second model's forward in between that took or overwrote it -- is an error
    of that step, not a surprise of the next one."""
    from . import fused_layer, model as _model
    left = {}
    if _BIAS_BWD_JOB:
        left["bias_bwd_job"] = list(_BIAS_BWD_JOB)
    if _FRONT_DEFER["hop"] is not None or _FRONT_DEFER["ni"] is not None:
        left["front_jobs"] = [k for k in ("hop", "ni") if _FRONT_DEFER[k] is not None]
    if _TOKEN_PENDING:
        left["token_pending"] = len(_TOKEN_PENDING)
    extra = [k for k in _WGRAD_DEFER if k not in ("on", "items")]
    if _WGRAD_DEFER["items"] or extra:
        left["wgrad_defer"] = dict(items=len(_WGRAD_DEFER["items"]), slots=extra)
    if fused_layer._PENDING_TAIL:
        left["layer_tails"] = len(fused_layer._PENDING_TAIL)
    if _model._PENDING_PACK:
        left["weight_pack"] = len(_model._PENDING_PACK)
    return left


def flush_deferred_wgrads():
    if _TOKEN_PENDING:
        stages = [p_["stage"] for p_ in _TOKEN_PENDING.values()]
        _TOKEN_PENDING.clear()
        _WGRAD_DEFER["items"] = []
        for k in ("hop", "hop_wide", "stock_tok", "psum"):
            _WGRAD_DEFER.pop(k, None)
        raise RuntimeError(f"a parked encoder-input backward chain was never completed (stages {stages}): gradients of this "
                           "step are invalid -- set MOBGT_NO_TOKEN_BWD_CHAIN=1 and report")
    items, _WGRAD_DEFER["items"] = _WGRAD_DEFER["items"], []
    hop = _WGRAD_DEFER.pop("hop", None)
    psum = _WGRAD_DEFER.pop("psum", None) or []
    for o in range(0, len(psum), 48):            # the sums over the library's split-K partial weight gradients: one launch
        part = psum[o:o + 48]
        n = len(part)
        check(_lib.lib().mobgt_partial_sum_multi(n, (ctypes.c_void_p * n)(*[p_[0].data_ptr() for p_ in part]),
                                                 (ctypes.c_void_p * n)(*[p_[1].data_ptr() for p_ in part]),
                                                 (ctypes.c_int * n)(*[p_[0].shape[0] for p_ in part]),
                                                 (ctypes.c_int64 * n)(*[p_[1].numel() for p_ in part]), _stream()),
              "mobgt_partial_sum_multi")
    tok, wide = _WGRAD_DEFER.pop("stock_tok", None), _WGRAD_DEFER.pop("hop_wide", None)
    if tok is not None and wide is not None:     # the stock step's tail: both in one grid
        dtab, ew, dw_, d_ew, d_dw, D, E, rt = wide
        check(_lib.lib().mobgt_stock_tail_bwd(*tok[0], _p(dtab), _p(ew), _p(dw_), _p(d_ew), _p(d_dw), D, E, 8, rt, _stream()),
              "mobgt_stock_tail_bwd")
    elif tok is not None:
        check(_lib.lib().mobgt_stock_tokens_bwd(*tok[0], _stream()), "mobgt_stock_tokens_bwd")
    elif wide is not None:
        dtab, ew, dw_, d_ew, d_dw, D, E, rt = wide
        check(_lib.lib().mobgt_hop_table_bwd(_p(dtab), _p(ew), _p(dw_), _p(d_ew), _p(d_dw), D, E, 8, rt, _stream()), "mobgt_hop_table_bwd")
    vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    if hop is not None and not items:            # nobody to ride with
        dtab, ew, dw_, d_ew, d_dw, D, E, rt = hop
        check(_lib.lib().mobgt_hop_table_bwd(_p(dtab), _p(ew), _p(dw_), _p(d_ew), _p(d_dw), D, E, 8, rt, _stream()), "mobgt_hop_table_bwd")
        hop = None
    for o in range(0, len(items), 32):
        part = items[o:o + 32]
        n = len(part)
        if hop is not None:                      # (with the first group)
            dtab, ew, dw_, d_ew, d_dw, D, E, rt = hop
            hop_args = [1, _p(dtab), _p(ew), _p(dw_), _p(d_ew), _p(d_dw), D, E, rt]
            hop = None
        else:
            hop_args = [0, None, None, None, None, None, 0, 0, 0]
        ptr = lambda t: t.data_ptr() if t is not None else None
        mv = []
        for it in part:
            mv += list(it[4])
        check(_lib.lib().mobgt_linear_wgrad_multi_hop(
            n, (vp * n)(*[ptr(it[0]) for it in part]), (i64 * n)(*[it[0].stride(0) for it in part]),
            (vp * n)(*[ptr(it[1]) for it in part]), (i64 * n)(*[it[1].stride(0) for it in part]),
            (vp * n)(*[ptr(it[2]) for it in part]), (vp * n)(*[ptr(it[3]) for it in part]), (cf * (3 * n))(*mv),
            (vp * n)(*[ptr(it[5]) for it in part]), (i64 * n)(*[it[5].stride(0) for it in part]),
            (vp * n)(*[ptr(it[6]) for it in part]), (ci * n)(*[int(it[7]) for it in part]),
            (i64 * n)(*[it[0].shape[0] for it in part]), (ci * n)(*[it[0].shape[1] for it in part]),
            (ci * n)(*[it[1].shape[1] for it in part]), (ci * n)(*[1 if it[0].dtype == torch.float32 else 0 for it in part]),
            *hop_args, _stream()), "mobgt_linear_wgrad_multi_hop")


def linear_wgrad_masked(g, x, g_mask=None, x_mask=None, mask_vals=(1.0, 1.0, 1.0), db=None, db_of_x=False, dw=None, g_out=None,
                        leaf=False):
    """dW [M,N] = (g * m(g_mask))^T (x * m(x_mask)) for f32 row-major g [R,M], x [R,N] (operands rounded to bf16 while
    loading, f32 accumulate); db (zero-initialised f32): += column sums of the masked g ([M]) or, `db_of_x`, x ([N]).
    `g_out` (f32, g's shape and strides): also receives g * m(g_mask) (for the data-gradient GEMM that follows)."""
    _require_cuda(g, x)
    R, M = g.shape
    N = x.shape[1]
    assert g.dtype == torch.float32 and x.dtype == torch.float32 and g.stride(1) == 1 and x.stride(1) == 1
    for m, t in ((g_mask, g), (x_mask, x)):
        assert m is None or (m.dtype == torch.float32 and m.shape == t.shape and m.stride() == t.stride())
    if dw is None:
        dw = zeros_f32((M, N), g.device)
    assert g_out is None or (g_mask is not None and g_out.shape == g.shape and g_out.stride() == g.stride())
    if leaf and _WGRAD_DEFER["on"] and g_out is None and M % 2 == 0 and N % 2 == 0:
        return _wgrad_defer(g, x, g_mask, x_mask, mask_vals, dw, db, db_of_x)
    check(_lib.lib().mobgt_linear_wgrad_masked(_p(g), g.stride(0), _p(x), x.stride(0), _p(g_mask), _p(x_mask), float(mask_vals[0]),
                                               float(mask_vals[1]), float(mask_vals[2]), _p(g_out), _p(dw), N, _p(db), int(db_of_x),
                                               R, M, N, _stream()), "mobgt_linear_wgrad_masked")
    return dw


def small_gemm_ok(a, b):
    return (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2
            and a.stride(1) == 1 and b.stride(1) == 1)


# ------------------------------------------------------------------- the encoder layer's small GEMMs
GEMM_BIAS, GEMM_GELU, GEMM_GELU_BWD, GEMM_ADD = 0, 1, 2, 3


def layer_gemm_ok(a, b, b_is_kn=False):
    """Shapes / layouts mobgt_layer_gemm takes: bf16, row-major, K % 32 == 0, N % 8 == 0, rows of at most 64 k elements
    of work per CU-filling launch (the library's big-tile kernels win on large M)."""
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.dim() != 2 or b.dim() != 2:
        return False
    M, K = a.shape
    N = b.shape[1] if b_is_kn else b.shape[0]
    if (b.shape[0] if b_is_kn else b.shape[1]) != K:
        return False
    return (K % 32 == 0 and N % 8 == 0 and a.stride(1) == 1 and b.stride(1) == 1 and a.stride(0) % 8 == 0
            and b.stride(0) % (2 if b_is_kn else 8) == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0
            and M <= 4096)


def layer_gemm(a, b, bias=None, b_is_kn=False, epilogue=GEMM_BIAS, aux_in=None):
    """acc = a @ (b if b_is_kn else b.T) (+ bias) with the epilogues of mobgt_layer_gemm (include/mobgt_hip.h):
    GEMM_BIAS -> C bf16;  GEMM_GELU -> (u, gelu(u)) bf16;  GEMM_GELU_BWD -> acc * gelu'(aux_in) bf16;
    GEMM_ADD -> acc + aux_in, f32, written IN PLACE into aux_in.  No autograd: the fused layer calls it both ways."""
    _require_cuda(a, b)
    M, K = a.shape
    N = b.shape[1] if b_is_kn else b.shape[0]
    dev = a.device
    aux_out = None
    if epilogue == GEMM_ADD:
        assert aux_in.dtype == torch.float32 and aux_in.shape == (M, N) and aux_in.is_contiguous()
        c = aux_in
    else:
        c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        if epilogue == GEMM_GELU:
            aux_out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        elif epilogue == GEMM_GELU_BWD:
            assert aux_in.dtype == torch.bfloat16 and aux_in.shape == (M, N) and aux_in.is_contiguous()
    if bias is not None:
        assert bias.dtype == torch.bfloat16 and bias.numel() == N and bias.is_contiguous()
    check(_lib.lib().mobgt_layer_gemm(_p(a), a.stride(0), _p(b), b.stride(0), int(b_is_kn), _p(bias), _p(c), N, epilogue,
                                      _p(aux_in), _p(aux_out), M, N, K, _stream()), "mobgt_layer_gemm")
    return (c, aux_out) if epilogue == GEMM_GELU else c


# ------------------------------------------------------------------------------ debug: NaN tracer
_NAN_TRACE = {"flags": None, "names": [], "on": False}


def nan_trace_enable(device, slots=256):
    _NAN_TRACE.update(flags=torch.zeros(slots, dtype=torch.float32, device=device), names=[], on=True)


def trace_nan(name, t, grad=True):
    """Debug aid (off by default): record inside the running stream / captured graph whether `t` (and its
    gradient) contains a NaN, without keeping `t` alive."""
    if not _NAN_TRACE["on"] or t is None:
        return t
    names, flags = _NAN_TRACE["names"], _NAN_TRACE["flags"]
    if name not in names:
        names.append(name)
        if grad:
            names.append("d/" + name)
    i = names.index(name)
    flags[i:i + 1].copy_(torch.isnan(t.detach()).any().float().reshape(1))
    if grad and t.requires_grad:
        j = names.index("d/" + name)
        def _hook(g, j=j):
            flags[j:j + 1].copy_(torch.isnan(g).any().float().reshape(1))
            return None
        t.register_hook(_hook)
    return t


def nan_trace_report():
    f = _NAN_TRACE["flags"].cpu().tolist()
    return [(n, bool(f[i])) for i, n in enumerate(_NAN_TRACE["names"])]


def linear_wgrad(g, x, with_bias=False, db=None, out_bias=None, leaf=False, dw=None):
    """(dW, db) of y = x W^T + b from g = dL/dy: dW [M,N] = g^T x (f32) and db [M] = g.sum(0) (f32, or None),
    for bf16 row-major g [R,M], x [R,N] (row strides may exceed the width: column slices are fine).
    `db`: an existing zero-initialised f32 [M] to accumulate the bias gradient into.
    `out_bias` (f32 [N]): added to every row of the product (the kernel used as a skinny `A^T B + bias`)."""
    _require_cuda(g)
    R, M = g.shape
    N = x.shape[1]
    assert g.dtype == x.dtype and g.dtype in (torch.bfloat16, torch.float32) and g.stride(1) == 1 and x.stride(1) == 1
    if dw is None:                    # (`dw` / `db`: zeroed f32 destinations the products are ADDED to -- e.g. gradient sinks)
        dw = zeros_f32((M, N), g.device)
    if out_bias is not None:
        assert db is None and not with_bias and out_bias.dtype == torch.float32 and out_bias.numel() == N
        check(_lib.lib().mobgt_linear_wgrad_bias(_p(g), g.stride(0), _p(x), x.stride(0), _p(out_bias.contiguous()), _p(dw), N, R, M, N,
                                                 _DT[g.dtype], _stream()), "mobgt_linear_wgrad_bias")
        return dw, None
    if db is None and with_bias:
        db = zeros_f32((M,), g.device)
    if leaf and _WGRAD_DEFER["on"] and M % 2 == 0 and N % 2 == 0:
        return _wgrad_defer(g, x, None, None, (1.0, 1.0, 1.0), dw, db, False), (db[:] if db is not None else None)
    check(_lib.lib().mobgt_linear_wgrad(_p(g), g.stride(0), _p(x), x.stride(0), _p(dw), N, _p(db), R, M, N, _DT[g.dtype],
                                        _stream()), "mobgt_linear_wgrad")
    return dw, db


def colsum(g):
    """Column sums of a 2-D f32/bf16 tensor as f32 [C] (bias gradient of a Linear / GraphConvolution)."""
    _require_cuda(g)
    g = g.contiguous()
    R, C = g.shape
    out = zeros_f32((C,), g.device)
    check(_lib.lib().mobgt_colsum(_p(g), _p(out), R, C, _DT[g.dtype], _stream()), "mobgt_colsum")
    return out

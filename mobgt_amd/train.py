"""Train-step driver: what Lightning's fit loop + DDP plugin do for the reference (`entry.py:141-161`,
`model_fqandtoyo.py:1434-1478, 1599-1616`), written for one process per GPU over RCCL.

* `FlatGrads`: every trainable parameter's `.grad` is a view into ONE flat fp32 buffer, so zeroing is one
  memset and the data-parallel exchange is ONE all-reduce (RCCL over xGMI).  Parameters the model never
  uses (25 tensors in the fq variant) keep `grad = None`, exactly as under the reference, so
  AdamW skips them the same way.  DEVIATION (documented in DESIGN.md section 7): a parameter that is used by the model but
  receives no gradient on ONE particular batch has a zero-filled slot that step and is updated like any other slot (weight
  decay, moment decay, one global step count), where torch.optim.AdamW would skip it for that step and keep a per-parameter
  step count.  In the fq / stock models every trained parameter is reached on every batch (tables are reached row-wise:
  an untouched ROW has a zero gradient under the reference too), so the two agree; a model for which that does not hold
  should use `configure_optimizers()` instead of the flat kernel.
* `TrainStep`: forward + loss + backward (+ all-reduce) + AdamW + PolynomialDecayLR.  With
  `use_graph=True` the forward/backward and the optimizer step are captured in hipGraphs per batch
  (static shapes per pre-collated batch); the learning rate and the dropout seed live in device scalars
  so a replay sees new values without re-capture.
"""
import os

import torch
import torch.distributed as dist

from . import ops
from .model import sync_external_shadows
from .lr import polynomial_decay_lr


def flat_offsets(params, align=8):
    """Element offset of every parameter inside the flat buffers (each slot starts on a multiple of `align` elements,
    so f32 slices are 32-byte and bf16 shadow slices 16-byte aligned) and the total length.  A parameter may ask for unused
    elements behind its slot (`p._mobgt_flat_slack`): a weight-gradient kernel whose operand is zero-padded by a column (the
    distance GCN's 303-wide first layer as 304 columns) writes one more row of (zero) products than the parameter has, straight
    into its gradient sink.  (Slack behind EVERY slot would separate q / k / v and the small per-layer gradients, which the fused
    layers need adjacent.)"""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + int(getattr(p, "_mobgt_flat_slack", 0)) + align - 1) // align * align
    return offs, off


class FlatGrads:
    def __init__(self, params, dtype=torch.float32):
        self.params = [p for p in params if p.requires_grad]
        self.offsets, n = flat_offsets(self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=dtype, device=dev)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        self.attach()

    def attach(self):
        for p, v in zip(self.params, self.views):
            p.grad = v

    def zero(self):
        self.flat.zero_()

    def release(self):
        """Before backward: drop the views so autograd hands over its freshly computed gradient tensors
        instead of launching one `grad += new` kernel per parameter (~130 tiny launches per step)."""
        for p in self.params:
            p.grad = None

    def gather(self, start=0, stop=None, grads=None):
        """After backward: one multi-tensor copy of the gradients of params[start:stop] into the flat buffer
        (`grads`: explicit gradient tensors, e.g. from torch.autograd.grad), then re-attach those views."""
        stop = len(self.params) if stop is None else stop
        ps, vs = self.params[start:stop], self.views[start:stop]
        if grads is None:
            grads = [p.grad for p in ps]
        for k in getattr(self, "zero_if_missing", ()):        # slots the per-step zeroing skips (TrainStep._skip)
            if start <= k < stop and grads[k - start] is None:
                vs[k - start].zero_()
        # nothing to copy for a gradient that was written in place (ops.grad_sink) or that does not exist (the flat
        # buffer is zeroed before every backward)
        todo = [(v, g) for g, v in zip(grads, vs) if g is not None and g.data_ptr() != v.data_ptr()]
        if todo:
            torch._foreach_copy_([v for v, _ in todo], [g for _, g in todo])
        for p, v in zip(ps, vs):
            p.grad = v

    def all_reduce_mean(self, group=None):
        """mean of the gradients over ranks (DDP semantics); no-op for world size 1."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))


def head_parameters(model, used):
    """The used parameters of `model.head_modules` (gradient complete once the backward has reached the encoder
    output), in `used` order; [] when the model does not declare a head."""
    names = getattr(model, "head_modules", ())
    ids = set()
    for n in names:
        m = getattr(model, n, None)
        if m is not None:
            ids |= {id(p) for p in m.parameters()}
    return [p for p in used if id(p) in ids]


def flat_order(model, used):
    """Order of the used parameters inside the flat buffers: the head's parameters first (one contiguous bucket
    whose all-reduce can start early), then the encoder layers from the LAST to the FIRST -- the order in which the backward
    pass completes their gradients, so that "everything up to layer k" is one growing prefix of the buffer and can be
    all-reduced while the layers below are still in their backward (TrainStep's layer-wise buckets) -- each layer as one
    contiguous run: q/k/v weights adjacent (the fused [3C, C] projection storage is then one slice of the flat parameter
    buffer), the block of small gradients, then the layer's remaining weights; then the rest."""
    used_ids = {id(p) for p in used}
    order, seen = [], set()
    for p in head_parameters(model, used):
        order.append(p)
        seen.add(id(p))
    def take(grp):
        if all(id(p) in used_ids and id(p) not in seen for p in grp):
            for p in grp:
                order.append(p)
                seen.add(id(p))
            return True
        return False
    layers = [m for m in model.modules() if hasattr(m, "self_attention") and hasattr(m, "ffn")]
    attn_done = set()
    for layer in reversed(layers):
        # an encoder layer: q/k/v weights, then -- in the order of the fused backward's block of small gradients
        # (fused_layer._FusedLayerFn.backward) -- q/k/v biases, output bias, FFN biases, the two LayerNorms: that block
        # is then ONE slice of the flat gradient buffer and nothing of it needs a gather copy
        m = layer.self_attention
        if not all(hasattr(m, a) for a in ("linear_q", "linear_k", "linear_v", "output_layer")):
            continue
        n1, nx = ((layer.ffn_norm1, layer.ffn_norm2) if hasattr(layer, "ffn_norm1")
                  else (getattr(layer, "ffn_norm", None), getattr(layer, "self_attention_norm", None)))
        if n1 is None or nx is None:
            continue
        if take([m.linear_q.weight, m.linear_k.weight, m.linear_v.weight, m.linear_q.bias, m.linear_k.bias, m.linear_v.bias,
                 m.output_layer.bias, layer.ffn.layer1.bias, layer.ffn.layer2.bias, n1.weight, n1.bias, nx.weight, nx.bias]):
            attn_done.add(id(m))
            for p in layer.parameters():              # the rest of this layer (Wo, W1, W2): behind its group, same bucket
                if id(p) in used_ids and id(p) not in seen:
                    order.append(p)
                    seen.add(id(p))
    for m in model.modules():
        if id(m) not in attn_done and all(hasattr(m, a) for a in ("linear_q", "linear_k", "linear_v")):
            take([m.linear_q.weight, m.linear_k.weight, m.linear_v.weight, m.linear_q.bias, m.linear_k.bias, m.linear_v.bias])
    return order + [p for p in used if id(p) not in seen]


class FlatParams:
    """All trained parameters as views of ONE fp32 buffer (same order as FlatGrads): AdamW becomes a single fused
    launch over one tensor instead of a multi-tensor sweep over ~130.  Parameter objects, names and values are
    unchanged (state_dict keeps working); untrained parameters are left alone, as the reference's optimizer
    skips them (their grad is None)."""

    def __init__(self, model, params):
        self.params = params
        self.offsets, n = flat_offsets(params)
        flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        with torch.no_grad():
            for p, off in zip(params, self.offsets):
                v = flat[off:off + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
        self.tensor = torch.nn.Parameter(flat)
        # re-point the attention modules' fused-QKV handles at the new storage
        for m in model.modules():
            if hasattr(m, "fuse_qkv_storage") and hasattr(m, "_wqkv"):
                w, b = m.linear_q.weight, m.linear_q.bias
                C = w.shape[0]
                nxt_w = m.linear_k.weight.data_ptr() == w.data_ptr() + w.numel() * 4
                nxt_b = m.linear_k.bias.data_ptr() == b.data_ptr() + b.numel() * 4
                if nxt_w and nxt_b:
                    wo = (w.data_ptr() - flat.data_ptr()) // 4
                    bo = (b.data_ptr() - flat.data_ptr()) // 4
                    m._wqkv = flat[wo:wo + 3 * C * w.shape[1]].view(3 * C, w.shape[1])
                    m._bqkv = flat[bo:bo + 3 * C]


def used_parameters(model, loss_fns):
    """Dry-run backward passes: the parameters that receive a gradient from ANY of them (the others stay grad=None,
    as under the reference, whose optimizer then skips them).  `loss_fns`: one callable per pre-collated batch -- a
    parameter may be reached on some batches only."""
    if callable(loss_fns):
        loss_fns = [loss_fns]
    hit = set()
    ops.sink_census_begin()
    try:
        for fn in loss_fns:
            for p in model.parameters():
                p.grad = None
            ops.sink_census_pass()
            fn().backward()
            hit |= {id(p) for p in model.parameters() if p.grad is not None}
    finally:
        census = ops.sink_census_end()
    for p in model.parameters():
        p.grad = None
    used = [p for p in model.parameters() if id(p) in hit]
    # parameters whose gradient has MORE than one producer per pass (tied / re-applied modules): no in-place sink
    # (ops.grad_sink: the producers would overwrite each other and autograd would add a buffer to itself)
    for p in used:
        p._mobgt_multi_use = census.get(id(p), 0) > 1
    return used


def recommended_env():
    """Environment a data-parallel process should set BEFORE `init_process_group("nccl")` when its steps are captured graphs
    (bench.py and the tests' workers do): c10d's event cache off -- a recycled event that was last recorded inside a capture must
    never reach the watchdog's polling (see TrainStep._quiesce_collectives)."""
    return {"TORCH_NCCL_CUDA_EVENT_CACHE": "0"}


def broadcast_parameters(model, src=0):
    """DDP's construction-time broadcast: every rank starts from rank `src`'s weights."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src)
        sync_external_shadows(model)


def assert_same_across_ranks(digest, what, device=None):
    """All-gather `digest` (bytes, same length everywhere) over the default process group; every rank raises RuntimeError
    naming the ranks that differ from rank 0.  No-op without a process group or with one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return
    world = dist.get_world_size()
    mine = torch.frombuffer(bytearray(digest), dtype=torch.uint8).clone()
    if dist.get_backend() == "nccl":
        mine = mine.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    got = [t.cpu() for t in got]
    bad = [r for r, t in enumerate(got) if not torch.equal(t, got[0])]
    if bad:
        raise RuntimeError(f"mobgt: {what} differs between rank 0 and ranks {bad} -- refusing to all-reduce")


def choose_ddp_form(model, batches, steps=20, warmup=5, candidates=None, **ts_kw):
    """Which form of the data-parallel step is the fastest ON THIS JOB'S RANKS (VERDICT r5 next #7b): every candidate
    -- the one-graph step with the exchange captured on the step's stream (host-issued over gloo), the two-phase overlap, the
    layer-wise overlap in three parts -- is built, prepared and timed for `steps` steps behind a barrier (MAX over ranks), the
    model's parameters are put back after each, and all ranks agree on the choice (`assert_same_across_ranks`).
    -> (name, TrainStep constructor arguments incl. the environment to set, {name: ms per step}).  With one rank or no process
    group: the default form, nothing timed.  No multi-GPU node has run this yet: the forms' order there is the measurement this
    function exists to make (the one-rank figures of rounds 4-5 cannot show what an overlap hides)."""
    import time
    forms = candidates or [("one_graph", dict(overlap=False), {}),
                           ("overlap_2", dict(overlap="force"), {"MOBGT_DDP_PARTS": "1"}),
                           ("overlap_3", dict(overlap="force"), {"MOBGT_DDP_PARTS": "3"})]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() <= 1:
        return forms[0][0], forms[0][1], {}
    dev = next(model.parameters()).device
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    saved_env = {k: os.environ.get(k) for _, _, env in forms for k in env}
    timings = {}
    for name, kw, env in forms:
        for k, v in env.items():
            os.environ[k] = v
        try:
            ts = TrainStep(model, batches, **dict(ts_kw, **kw))
            ts.prepare()
            for i in range(warmup):
                ts.step(i)
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            for i in range(steps):
                ts.step(warmup + i)
            torch.cuda.synchronize(dev)
            el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            timings[name] = float(el.item()) / steps * 1e3
            faults = ts.check_faults(on_fault="return")
            if faults:
                timings[name] = float("inf")
        finally:
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        with torch.no_grad():
            model.load_state_dict(sd0)
        del ts
    best = min(timings, key=lambda n: (timings[n], n))
    assert_same_across_ranks(best.encode().ljust(16, b" "), "the chosen data-parallel step form", dev)
    kw, env = next((kw, env) for n, kw, env in forms if n == best)
    return best, dict(kw, _env=env), timings


class TrainStep:
    def __init__(self, model, batches, autocast_dtype=None, use_graph=True, seed=1, overlap=True, batch_fn=None,
                 grad_comm_dtype=None, keep_head_rows=False):
        """`batches`: pre-collated batches resident on the device (their tensors are the static inputs of the captured
        graphs).  `batch_fn`: optional callable applied to a batch INSIDE the step (and so inside its graph) before the model
        sees it -- `EpochLoop` passes raw, un-collated arrays + `DeviceCollator.finish`, so that collating a fresh batch is
        part of the replayed step.  More batches can be added after `prepare()` with `add_batch`."""
        self.model = model
        self.batches = list(batches)
        self.batch_fn = batch_fn
        self.keep_head_rows = bool(keep_head_rows)      # tests: TrainStep.enc_outs[i] = graph-token rows of batch i's last step
        # torch.bfloat16: the gradient exchange moves bf16 (half the bytes over xGMI; RCCL then also SUMS in bf16 -- 8 ranks:
        # relative error ~2^-8 per element on top of the gradient's own bf16-operand noise); the result is widened back
        # into the fp32 flat buffer the optimizer reads.  Default None: fp32 exchange (what the parity tests pin).
        self.grad_comm_dtype = grad_comm_dtype
        self.autocast_dtype = autocast_dtype
        dev = next(model.parameters()).device
        self.device = dev
        self.seed_dev = torch.tensor([seed], dtype=torch.int64, device=dev)
        for m in model.modules():
            if hasattr(m, "seed_dev"):
                m.seed_dev = self.seed_dev
        from . import ops
        ops.set_dropout_state(self.seed_dev, seed)
        # room for every gradient that kernels accumulate into (all but the few huge matrices torch's GEMMs write)
        n_arena = sum(p.numel() for p in model.parameters() if p.numel() <= (1 << 16))
        self.arena = ops.ZeroArena(dev, n=(max(1 << 21, int(n_arena * 1.5) + (1 << 18)) + 3) // 4 * 4)
        model.train()
        # A model that already ran a backward pass on another stream may still hold that autograd graph (the fq model
        # keeps its encoder output in `_enc_out`), and with it AccumulateGrad nodes bound to THAT stream; captured on
        # ours they would put cross-stream waits into the graph (observed: abort in capture_end).  Let go of it first.
        if getattr(model, "_enc_out", None) is not None:
            model._enc_out = None
        model._bias_pack, model._cuts = None, {}
        for l in getattr(model, "layers", []):
            l._mobgt_cut = False
        for p in model.parameters():
            p.grad = None
        import gc
        gc.collect()
        # One side stream for the dry run, the warm-ups and every capture: autograd's AccumulateGrad nodes are
        # created on the stream of the first backward and must match the capture stream later on.
        self.stream = torch.cuda.Stream(device=dev) if use_graph else None
        with self._on_stream():
            used = used_parameters(model, [lambda b=b: self._loss(b) for b in batches])
        used = flat_order(model, used)
        self.flat = FlatGrads(used)
        self.n_head = len(head_parameters(model, used))              # params[:n_head] = the early bucket
        self.n_head_elems = self.flat.offsets[self.n_head] if self.n_head < len(used) else self.flat.flat.numel()
        self.flat_params = FlatParams(model, used)
        # big gradients are written in place (no gather copy) -- except those of parameters that more than one autograd
        # node produces per pass (`used_parameters`' census): they keep torch's accumulate-then-copy path
        single = [(p, v) for p, v in zip(self.flat.params, self.flat.views) if not getattr(p, "_mobgt_multi_use", False)]
        self.sinkless = [p for p in self.flat.params if getattr(p, "_mobgt_multi_use", False)]
        ops.set_grad_sinks([p for p, _ in single], [v for _, v in single])
        # ... and a model that has such parameters runs its backward WITHOUT deferred weight-gradient launches: a deferred
        # producer returns its buffer before the grouped launch fills it, and autograd would sum two still-empty buffers
        self._defer = not self.sinkless
        # The classifier's weight gradient (61 % of the S-FSQ model's gradient bytes) is OVERWRITTEN in full by every backward
        # pass -- by the skinny weight-gradient kernel writing into its sink, or by the gather's copy of a library result --
        # so the per-step zeroing leaves that slice alone (it is zeroed by hand in the one case nothing writes it: a step in
        # which the parameter received no gradient)
        self._skip = None
        w = getattr(getattr(model, "out_proj", None), "weight", None)
        if w is not None:
            for i, q in enumerate(self.flat.params):
                if q is w and w.numel() >= (1 << 20):
                    self._skip = (i, int(self.flat.offsets[i]), int(self.flat.offsets[i]) + w.numel())
                    self.flat.zero_if_missing = (i,)
        self.flat_params.tensor.grad = self.flat.flat
        self.lr_dev = torch.tensor(float(model.peak_lr), dtype=torch.float32, device=dev)
        # AdamW (model_fqandtoyo.py:1599-1616 defaults) as ONE kernel over the flat buffers that also refreshes the bf16
        # shadow weights the layer GEMMs read (csrc/layer.hip: mobgt_adamw_flat); its step count is the per-step device
        # counter `seed_dev` minus a base fixed at the first optimizer call
        n = self.flat.flat.numel()
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.shadow_flat = self._attach_shadows()
        self._step_base = None
        self.betas, self.eps = (0.9, 0.999), 1e-8
        self.sched_state = dict(step_count=1, warmup=model.warmup_updates, tot=model.tot_updates, lr=model.peak_lr,
                                end_lr=model.end_lr, power=1.0)
        # PolynomialDecayLR with power 1 is evaluated inside the optimizer kernel from the device step counter (no
        # per-step host write of the learning rate); `lr_dev` mirrors it for inspection and serves any other schedule
        s = self.sched_state
        # (5th entry: offset between the kernel's step count t and the schedule's step_count; both start at 1)
        self.sched_dev = torch.tensor([float(s["warmup"]), float(s["tot"]), float(s["lr"]), float(s["end_lr"]), 0.0],
                                      dtype=torch.float32, device=dev) if s["power"] == 1.0 and s["warmup"] > 0 else None
        self._set_lr()
        self.use_graph = use_graph
        self.graphs = {}
        self._loss_ref = torch.zeros((), device=dev)          # the loss tensor of the step that ran last
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # MOBGT_FORCE_COMM=1 (tests): a process group of ONE rank takes the data-parallel path -- buckets, exchange on the
        # collective's stream, bf16 exchange buffer -- so that the RCCL ("nccl") branch executes on a one-GPU box
        self.force_comm = (os.environ.get("MOBGT_FORCE_COMM") == "1" and dist.is_available() and dist.is_initialized())
        self.ddp = self.world > 1 or self.force_comm
        # Data parallel, three forms of one step:
        #  * default over RCCL -- `one_graph`: the WHOLE step is one hipGraph per batch: forward, backward, the gradient
        #    all-reduce issued on the step's OWN stream (a synchronous c10d collective runs on the caller's stream; inside a
        #    capture it becomes a node of the graph: no second stream, no cross-stream edge, no host call between replays),
        #    the average and AdamW.  The exchange is serial with the backward -- nothing hides it -- but the step carries no
        #    structure cost: round 4's four-replay form cost +25 % before any wire time (0.577 -> 0.724 ms with one forced
        #    rank), more than the overlap can win back on a 0.6 ms step;
        #  * MOBGT_DDP_OVERLAP=1 (or overlap="force": tests) -- `overlap`: the backward split at the encoder output (and, with
        #    MOBGT_DDP_PARTS > 1, in front of a few layers) into graphs replayed one after the other, every completed slice
        #    of the flat buffer all-reduced asynchronously on RCCL's stream beside the next replay;
        #  * gloo (host collectives are not capturable) or MOBGT_DDP_HOST_EXCHANGE=1: forward + backward graph, all-reduce
        #    issued by the host, optimizer graph.
        want_overlap = overlap == "force" or (bool(overlap) and os.environ.get("MOBGT_DDP_OVERLAP") == "1")
        self.overlap = bool(want_overlap and use_graph and (self.ddp or overlap == "force") and self.n_head > 0
                            and hasattr(model, "_enc_out"))
        self.graphs_b, self._g_enc, self._loss_slots, self.enc_outs = {}, {}, {}, {}
        self._loss_slots_nocomm = {}
        self._plan_buckets()
        self.one_graph = bool(self.ddp and use_graph and not self.overlap and dist.get_backend() == "nccl"
                              and os.environ.get("MOBGT_DDP_HOST_EXCHANGE") != "1")
        self.graphs_nocomm = {}
        self.comm = True        # False: skip the gradient exchange (bench.py measures the exposed all-reduce time that way)
        self._prepared = False
        self.comm_buf = (torch.empty(self.flat.flat.numel(), dtype=grad_comm_dtype, device=dev)
                         if (grad_comm_dtype is not None and self.ddp) else None)
        self.check_layout_across_ranks()

    def _plan_buckets(self):
        """Layer-wise gradient buckets (data parallel; VERDICT r3 next #3a).  `flat_order` lays the buffers out as
        [head | layer L-1 | ... | layer 0 | rest]: the order in which the backward pass completes the gradients.  Phase B -- the
        backward from the encoder output down -- is cut in front of a few layers into PARTS, each its own hipGraph over the one
        autograd graph (torch.autograd.grad from the previous cut's gradient to the next cut's activation + that part's
        parameters); after every part the slice of the flat buffer it completed is all-reduced on RCCL's stream while the next
        part replays.  What is exposed at the end is the LAST bucket only (node features, GCNs, embedding and bias tables: a few
        MB) instead of everything behind the head (39 % of the bytes in round 3).  `self.parts`: [(cut layer or None, first
        parameter, end parameter, first element, end element)], in execution order; [] = the single phase B of round 3.
        MOBGT_DDP_PARTS: number of parts (default 1 = the single phase B; the layer-wise parts have run over RCCL with one rank
        and over gloo with two, never on a multi-GPU node: ADVICE r4)."""
        self.parts = []
        layers = list(getattr(self.model, "layers", []))
        n_parts = int(os.environ.get("MOBGT_DDP_PARTS", "1"))
        if not self.overlap or len(layers) < 2 or n_parts < 2:
            return
        pos = {id(p): i for i, p in enumerate(self.flat.params)}
        runs, at = [], self.n_head
        for li in reversed(range(len(layers))):
            idx = sorted(pos[id(p)] for p in layers[li].parameters() if id(p) in pos)
            if not idx or idx != list(range(at, at + len(idx))):
                return                               # (a layer whose parameters are not one run behind its successor's: keep one phase B)
            at += len(idx)
            runs.append((li, idx[0], at))
        # layer groups of (nearly) equal size, top of the stack first; the last part takes what is left of the model
        n_grp = min(n_parts - 1, len(runs))
        cuts = [runs[(g + 1) * len(runs) // n_grp - 1] for g in range(n_grp)]      # the LOWEST layer of each group
        offs = list(self.flat.offsets) + [self.flat.flat.numel()]
        lo = self.n_head
        for li, _, hi in cuts:
            self.parts.append((li, lo, hi, int(offs[lo]), int(offs[hi])))
            layers[li]._mobgt_cut = True
            lo = hi
        n = len(self.flat.params)
        self.parts.append((None, lo, n, int(offs[lo]), int(offs[n])))

    def layout_digest(self):
        """Hash of what every rank must agree on before one flat buffer can be all-reduced: which parameters are trained,
        in which order, at which offsets (`used_parameters` is computed per rank from that rank's own batches)."""
        import hashlib
        names = {id(p): n for n, p in self.model.named_parameters()}
        h = hashlib.sha256()
        for p, off in zip(self.flat.params, self.flat.offsets):
            h.update(("%s:%d:%d;" % (names.get(id(p), "?"), off, p.numel())).encode())
        h.update(("n=%d;head=%d;parts=%s" % (self.flat.flat.numel(), self.n_head_elems, [p[1:] for p in getattr(self, "parts", [])])).encode())
        return h.digest()[:8]

    def check_layout_across_ranks(self):
        """All-gather the layout digest and abort on a mismatch (a rank whose dry-run batches never reach some parameter
        would otherwise all-reduce a differently laid out buffer: silent garbage)."""
        assert_same_across_ranks(self.layout_digest(), "flat parameter / gradient layout (different sets of trained parameters?)",
                                 self.device)

    def _attach_shadows(self):
        """bf16 copy of the whole flat parameter buffer; the fused layers' shadow weights become views of it, kept
        current by the optimizer kernel instead of a per-step multi-tensor copy (model.refresh_shadows skips them)."""
        layers = [l for l in getattr(self.model, "layers", []) if getattr(l, "fused", False)
                  and getattr(l, "act_dtype", torch.float32) == torch.bfloat16]
        if not layers:
            return None
        flat = self.flat_params.tensor.detach()
        shadow = flat.to(torch.bfloat16)
        base = flat.data_ptr()
        plan = []
        for layer in layers:
            mha = layer.self_attention
            wqkv, bqkv = mha.fuse_qkv_storage()
            masters = (wqkv, bqkv, mha.output_layer.weight, mha.output_layer.bias, layer.ffn.layer1.weight,
                       layer.ffn.layer1.bias, layer.ffn.layer2.weight, layer.ffn.layer2.bias)
            views = []
            for m in masters:
                off = (m.data_ptr() - base) // 4
                if off < 0 or off + m.numel() > flat.numel() or not m.is_contiguous():
                    return None                      # not a slice of the flat buffer: every layer keeps its own copies
                views.append(shadow[off:off + m.numel()].view(m.shape))
            plan.append((layer, tuple(views)))
        for layer, views in plan:                    # all layers validated: only now hand the views over
            layer._shadows = views
            layer._shadow_external = True
        import weakref
        self.model._shadow_sync = weakref.WeakMethod(self.sync_shadows)
        return shadow

    def sync_shadows(self):
        """Re-derive the bf16 shadow weights from the fp32 masters.  The optimizer kernel keeps them current step by
        step; anything ELSE that writes parameters after this TrainStep was built (checkpoint load, broadcast, manual
        edits) must call this (checkpoint.load_lightning_checkpoint and train.broadcast_parameters do)."""
        if getattr(self, "shadow_flat", None) is not None:
            with torch.no_grad():
                self.shadow_flat.copy_(self.flat_params.tensor)

    def _opt_step(self):
        from . import _lib
        from .ops import _p, _stream
        if self._step_base is None:                  # t = 1 on this very call; the counter advances once per step
            self._step_base = int(self.seed_dev.item()) - 1
        n = self.flat.flat.numel()
        _lib.check(_lib.lib().mobgt_adamw_flat(_p(self.flat_params.tensor), _p(self.flat.flat), _p(self.exp_avg),
                                               _p(self.exp_avg_sq), _p(self.shadow_flat), n, _p(self.lr_dev),
                                               _p(self.sched_dev), _p(self.seed_dev), self._step_base, self.betas[0],
                                               self.betas[1], self.eps,
                                               float(self.model.weight_decay), _stream()), "mobgt_adamw_flat")

    def _capturing(self, graph):
        """torch.cuda.graph on the trainer's stream.  With a process group, captures use the THREAD-LOCAL error mode: c10d's
        watchdog thread polls the events of earlier collectives (the layout all-gather, the eager warm-up exchange) with
        hipEventQuery, which the default global mode turns into an error of that thread while THIS one is capturing -- the
        process then dies in the watchdog (seen once in six runs of tests/test_gpu_train.py::test_rccl_branch_executes_on_one_gpu)."""
        if self.ddp:
            self._quiesce_collectives()
        return torch.cuda.graph(graph, stream=self.stream, capture_error_mode="thread_local" if self.ddp else "global")

    def _quiesce_collectives(self):
        """Let c10d's watchdog retire every eager collective before a capture starts (device idle, then two of its 100 ms polling
        rounds): a step graph captures a collective, and a watchdog that still polls events while events are being recorded into
        a capture has been seen to die on `hipErrorCapturedEvent` (one run in six; an event it held was recorded inside the
        capture -- c10d recycles its events through a cache, `TORCH_NCCL_CUDA_EVENT_CACHE=0` in `recommended_env()` switches that
        off).  Costs 0.25 s per captured graph, once."""
        import time
        if not getattr(TrainStep, "_env_warned", False) and dist.get_backend() == "nccl":
            missing = {k: v for k, v in recommended_env().items() if os.environ.get(k) != v}
            if missing:
                # (the pause below is a timing heuristic; what removes the race is the watchdog not recycling events)
                import warnings
                TrainStep._env_warned = True
                warnings.warn(f"mobgt TrainStep: capturing a step graph under a process group without {missing} in the environment "
                              "(train.recommended_env(), to be set BEFORE init_process_group): c10d's watchdog may poll an event that "
                              "is being recorded into the capture")
        torch.cuda.synchronize(self.device)
        time.sleep(0.25)

    def _on_stream(self):
        import contextlib
        if self.stream is None:
            return contextlib.nullcontext()
        self.stream.wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(self.stream)

    def _join(self):
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    # learning rate of lr.py:17-31, kept in a device scalar so captured optimizer graphs see it
    def _set_lr(self):
        s = self.sched_state
        self.lr = polynomial_decay_lr(s["step_count"], s["warmup"], s["tot"], s["lr"], s["end_lr"], s["power"])
        if self.sched_dev is None:
            self.lr_dev.fill_(self.lr)

    def _loss(self, batch):
        if self.batch_fn is not None:
            batch = self.batch_fn(batch)
        if self.autocast_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                return self.model.training_step(batch, 0)
        return self.model.training_step(batch, 0)

    def _prologue(self):
        """Zero the flat gradient buffer (in-place gradient sinks accumulate into it; unused slots stay zero) and the
        zero arena (all small zero-initialised accumulators of the step), advance the step counter (new dropout masks,
        AdamW's t): one launch."""
        from . import _lib
        from .ops import _p, _stream
        self.flat.release()
        self.arena.off = 0
        ops.set_zero_arena(self.arena)             # valid from here to the end of this backward pass only
        lo, hi = (self._skip[1], self._skip[2]) if self._skip is not None else (0, 0)
        _lib.check(_lib.lib().mobgt_step_prologue_skip(_p(self.flat.flat), self.flat.flat.numel(), lo, hi, _p(self.arena.buf),
                                                       self.arena.buf.numel(), _p(self.seed_dev), _stream()),
                   "mobgt_step_prologue_skip")

    def _fwd_bwd(self, batch, slot=None):
        self._prologue()
        loss = self._loss(batch)
        if slot is not None and self.keep_head_rows and getattr(self.model, "_enc_out", None) is not None:
            # (parity tests: a copy of the encoder output's graph-token rows -- what the classifier head reads -- as one more node
            #  of this batch's graph; the encoder output itself is overwritten in place by the backward pass)
            self.enc_outs[slot] = self.model._enc_out[:, 0, :].detach().clone()
        ops.wgrad_deferral(self._defer)          # leaf weight gradients are recorded ...
        try:
            loss.backward(gradient=ops.unit_grad(loss.device))
            ops.flush_deferred_wgrads()          # ... and issued as ONE launch, before anything reads a gradient
        finally:
            ops.wgrad_deferral(False)
        ops.set_zero_arena(None)
        left = ops.step_state_leftovers()
        if left:
            raise RuntimeError(f"mobgt: the step left parked work behind {left} -- its gradients are incomplete (another model's "
                               "forward / backward inside this step?)")
        self.flat.gather()
        self._keep_loss(loss, slot)

    def _keep_loss(self, loss, slot):
        """The loss scalar stays where the loss kernel wrote it (a graph's static output), no copy launch."""
        loss = loss.detach()
        if slot is None:
            self._loss_ref = loss
        else:
            self._loss_slots[slot] = loss

    @property
    def loss_out(self):
        return self._loss_ref

    # ---- the same step in two phases (data parallel): [forward, loss, head backward] | [rest of the backward]
    def _phase_a(self, batch, i):
        self._prologue()
        loss = self._loss(batch)
        enc = self.model._enc_out
        head = self.flat.params[:self.n_head]
        ops.wgrad_deferral(self._defer)
        try:
            grads = torch.autograd.grad(loss, head + [enc], grad_outputs=ops.unit_grad(loss.device), allow_unused=True)   # frees only the nodes it ran
            ops.flush_deferred_wgrads()
        finally:
            ops.wgrad_deferral(False)
        self.flat.gather(0, self.n_head, grads=list(grads[:-1]))
        self._g_enc[i] = (enc, grads[-1])
        self._keep_loss(loss, i)

    def _phase_b(self, i):
        enc, g_enc = self._g_enc[i]
        ops.wgrad_deferral(self._defer)
        try:
            torch.autograd.backward([enc], [g_enc])
            ops.flush_deferred_wgrads()
        finally:
            ops.wgrad_deferral(False)
        ops.set_zero_arena(None)
        self.flat.gather(self.n_head, None)

    def _phase_b_part(self, i, s):
        """Part s of phase B (see _plan_buckets): from the gradient the previous part left at its cut down to this part's cut."""
        li, lo, hi, _, _ = self.parts[s]
        src, g_src = self._g_enc[i]
        last = li is None
        ops.wgrad_deferral(self._defer)
        try:
            if not last:
                cut = self.model._cuts[li]
                grads = torch.autograd.grad([src], self.flat.params[lo:hi] + [cut], grad_outputs=[g_src], allow_unused=True)
                self._g_enc[i] = (cut, grads[-1])
                grads = list(grads[:-1])
            else:
                # (the bias tables' node hangs off every layer through the pack's token, not off the encoder input: name it)
                tok = getattr(getattr(self.model, "_bias_pack", None), "token", None)
                outs, gouts = [src], [g_src]
                if tok is not None and tok.requires_grad:
                    outs.append(tok)
                    gouts.append(torch.zeros_like(tok))
                torch.autograd.backward(outs, gouts)
                grads = None
            ops.flush_deferred_wgrads()
        finally:
            ops.wgrad_deferral(False)
        if last:
            ops.set_zero_arena(None)
        self.flat.gather(lo, hi, grads=grads)

    def _warmup(self, i, check=False):
        """One eager pass of batch i on the capture stream (allocator, lazy initialisations) that does not count as a training
        step: the step counter (dropout stream, AdamW's t) is put back, so that a replayed sequence draws the masks an eager
        sequence draws, however many graphs were captured on the way.  `check` (add_batch): the trained-parameter set -- flat
        layout, sinks, optimizer -- was fixed by the batches given at construction; a batch that reaches a parameter outside it
        would leave that parameter silently untrained.  Looked for in THIS pass (no extra dry run: ADVICE r4).  The
        error is raised on the rank that sees it: ranks meet new shape buckets at different steps, so no collective may run here
        (it would pair with another rank's gradient exchange); under torch.distributed.run a rank that exits takes the job down."""
        batch = self.batches[i]
        known = {id(p) for p in self.flat.params}
        import warnings
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)                   # (torch raises this warning once per process otherwise)
        try:
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                with self._on_stream():
                    saved = self.seed_dev.clone()
                    if check:
                        for p in self.model.parameters():
                            if id(p) not in known:
                                p.grad = None
                    self._fwd_bwd(batch)
                    self.seed_dev.copy_(saved)
        finally:
            torch.set_warn_always(warn_always)
        for w_ in seen:
            if "AccumulateGrad node's stream does not match" in str(w_.message):
                # An autograd graph of an EARLIER forward pass of this model is still alive (a loss tensor kept by the caller, a
                # traceback that holds one): its AccumulateGrad nodes are bound to the stream of that pass, this warm-up ran on
                # the trainer's capture stream, and the capture that follows would record a cross-stream wait on the legacy
                # stream -- hipStreamEndCapture then takes the process down (a core dump, seen in round 6 behind a failed test).
                raise RuntimeError("mobgt TrainStep: an autograd graph of an earlier forward pass of this model is still alive (a loss "
                                   "tensor or an exception traceback holding one?).  Its AccumulateGrad nodes belong to another stream and "
                                   "a hipGraph capture of the backward pass would abort the process.  Delete every reference to earlier "
                                   "losses / outputs of the model (del loss; gc.collect()) before TrainStep.prepare().")
            warnings.warn_explicit(w_.message, w_.category, w_.filename, w_.lineno)
        with self._on_stream():
            extra = [n for n, p in self.model.named_parameters() if id(p) not in known and p.grad is not None] if check else []
        self._join()
        if check and extra:
            raise RuntimeError(f"TrainStep.add_batch: this batch reaches parameters the trainer was not built for: {extra[:5]}")

    def _capture(self, i, warm=True):
        batch = self.batches[i]
        if warm:
            self._warmup(i)
        g = torch.cuda.CUDAGraph()
        # private memory pool per graph: the graphs are replayed in data order, not capture order, and a shared
        # pool is only safe for capture-order replay (measured: NaNs on the second lap with a shared pool)
        if not self.overlap:
            with self._capturing(g):
                self._fwd_bwd(batch, slot=i)
            return g
        # two graphs over ONE autograd graph: phase B is captured right after phase A and replays its kernels on the
        # activations phase A's replay leaves at the same addresses (A's private pool; nothing else writes there)
        with self._capturing(g):
            self._phase_a(batch, i)
        if self.parts:
            gbs = []
            for s in range(len(self.parts)):
                gb = torch.cuda.CUDAGraph()
                with self._capturing(gb):
                    self._phase_b_part(i, s)
                gbs.append(gb)
            self.graphs_b[i] = gbs
            return g
        gb = torch.cuda.CUDAGraph()
        with self._capturing(gb):
            self._phase_b(i)
        self.graphs_b[i] = [gb]
        return g

    def _exchange(self):
        """Sum of the flat gradient buffer over the ranks and the average, issued on the CURRENT stream (eager, or as nodes of the
        step graph being captured): through the bf16 exchange buffer when there is one (half the bytes on the wire; the widening
        pass applies 1 / world), else in place."""
        if self.comm_buf is not None:
            self.comm_buf.copy_(self.flat.flat)
            dist.all_reduce(self.comm_buf, op=dist.ReduceOp.SUM)
            torch.mul(self.comm_buf, 1.0 / self.world, out=self.flat.flat)
        elif self.world > 1:
            self.flat.all_reduce_mean()
        else:                                  # (one forced rank: the collective itself, no division)
            dist.all_reduce(self.flat.flat, op=dist.ReduceOp.SUM)

    def prepare(self):
        """Capture one forward/backward graph per batch and one optimizer graph."""
        if not self.use_graph:
            return
        self.pool = torch.cuda.graph_pool_handle()
        for i in range(len(self.batches)):
            self.graphs[i] = self._capture(i)
        # One eager optimizer call before the capture (first-launch initialisation must not happen inside a capture),
        # made side-effect free: parameters, shadows and both Adam moments are put back, so the first replayed step is
        # AdamW's t = 1 on fresh moments, exactly like torch.optim.AdamW's first step -- on every rank alike.
        with self._on_stream():
            with torch.no_grad():
                saved = self.flat_params.tensor.detach().clone()
                self._step_base = int(self.seed_dev.item()) - 1
                self._opt_step()
                self.flat_params.tensor.copy_(saved)
                self.exp_avg.zero_()
                self.exp_avg_sq.zero_()
                self.flat.flat.zero_()
                self.sync_shadows()
            # every later prologue (one per replayed step) advances the counter by one: t = counter - base = 1, 2, ...
            self._step_base = int(self.seed_dev.item())
        self._join()
        self.opt_graph = torch.cuda.CUDAGraph()
        with self._capturing(self.opt_graph):
            self._opt_step()
        # Single process, no all-reduce between backward and optimizer: the optimizer rides at the end of every batch's
        # graph -- one graph launch per step instead of two (the boundary between two replayed graphs idles the device
        # for ~8.7 us: `tools/prof_gaps.sh`).  Every warm-up has run by now, so this second capture only records.
        self.fused_opt = (not self.overlap and (not self.ddp or self.one_graph))
        if self.fused_opt:
            if self.one_graph:
                with self._on_stream():            # (the collective's first call on these buffers: outside any capture)
                    saved = self.flat.flat.clone()
                    self._exchange()
                    self.flat.flat.copy_(saved)
                self._join()
            for i in range(len(self.batches)):
                self._capture_with_opt(i)
        self._prepared = True

    def _capture_with_opt(self, i, comm=None):
        """The whole step of batch i as ONE graph: forward + backward [+ gradient exchange on this stream: `one_graph`] + AdamW."""
        comm = self.one_graph if comm is None else comm
        g = torch.cuda.CUDAGraph()
        main = comm or not self.one_graph
        kept = self._loss_slots.get(i)
        with self._capturing(g):
            self._fwd_bwd(self.batches[i], slot=i)
            if comm:
                self._exchange()
            self._opt_step()
        if main:
            self.graphs[i] = g
        else:
            # (the no-exchange twin writes its loss into its OWN static tensor -- the main graph's slot keeps pointing at
            #  the main graph's output, whichever of the two was captured last)
            self.graphs_nocomm[i] = g
            self._loss_slots_nocomm[i] = self._loss_slots[i]
            if kept is not None:
                self._loss_slots[i] = kept

    # ---- several steps per graph replay (round 6) -----------------------------------------------------------------------
    # Between two replayed step graphs the device idles (~60 us of a 0.595 ms S-FSQ step in round 5: 0.534 ms of graph wall
    # time): the replay that follows cannot start before the runtime has retired the one in front of it.  Nothing of a step
    # lives on the host -- the dropout stream, AdamW's t and the learning-rate schedule are device counters the step's own
    # kernels advance -- so k consecutive steps on pre-collated batches can be ONE graph: k x (forward, backward [, exchange],
    # AdamW), one replay, no host round trip between them.  The graph holds k steps' activations (S-FSQ: ~30 MB per step).
    def _capture_group(self, i0, k):
        if not (self.use_graph and self._prepared and getattr(self, "fused_opt", False) and self.sched_dev is not None):
            raise RuntimeError("TrainStep.step_group needs prepare(), hipGraphs, the one-graph step form and the device-side schedule")
        nb = len(self.batches)
        g = torch.cuda.CUDAGraph()
        with self._capturing(g):
            for j in range(k):
                self._fwd_bwd(self.batches[(i0 + j) % nb], slot=("group", i0, k, j))
                if self.one_graph:
                    self._exchange()
                self._opt_step()
        self.graphs_group[(i0, k)] = g

    def step_group(self, i, k):
        """Steps i, i + 1, ... i + k - 1 (batches taken cyclically from the pre-collated pool, like k calls of `step`) as ONE
        graph replay; -> the last step's loss tensor.  Same arithmetic, same counters, same order as k single steps."""
        nb = len(self.batches)
        key = (i % nb, int(k))
        if not hasattr(self, "graphs_group"):
            self.graphs_group = {}
        if key not in self.graphs_group:
            self._capture_group(*key)
        self._loss_ref = self._loss_slots[("group", key[0], key[1], key[1] - 1)]
        self.graphs_group[key].replay()
        self.sched_state["step_count"] += key[1]
        self._set_lr()
        return self.loss_out

    # ---- peer waits that gave up (csrc/chain.hip WS_FAULT, head.hip, smallgcn.hip): detection and recovery ---------------
    def recapture(self):
        """Capture every step graph again (after ops.SAFE_FORMS changed which kernels a step launches).  Parameters,
        optimizer state and the step counter are left as they are."""
        if not (self.use_graph and self._prepared):
            return
        self.graphs_nocomm.clear()
        for i in range(len(self.batches)):
            self.graphs[i] = self._capture(i)
            if self.fused_opt:
                self._capture_with_opt(i)

    def check_faults(self, on_fault="raise"):
        """Workgroups that gave up waiting for their cluster partners / at a grid barrier since the last check ({} = none;
        synchronises the device).  The steps since then may have trained on garbage gradients: "raise" -> RuntimeError."""
        f = ops.peer_wait_faults(reset=True)
        if self.ddp:                                    # every rank must take the same decision
            t = torch.tensor([float(sum(f.values()))], device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if float(t.item()) > 0 and not f:
                f = {"other_rank": int(t.item())}
        if f and on_fault == "raise":
            raise RuntimeError(f"mobgt: workgroups gave up waiting for their peers {f} -- the affected steps' gradients are invalid "
                               "(co-residency lost to another stream's kernels?); use TrainStep.guarded_step or MOBGT_SAFE_FORMS=1")
        return f

    def guarded_step(self, i):
        """step(i) with the promise that no peer-wait fault goes unnoticed: state snapshot -> step -> device sync + fault check;
        on a fault the snapshot is restored, every launch switches to its form WITHOUT cross-workgroup waits
        (ops.SAFE_FORMS: one workgroup per row block, the head as three launches, the GCN layer by layer), the graphs are
        captured again and the step is re-run with the same dropout counter.  Costs three buffer copies and a host
        synchronisation per step: meant for runs that share the device with other streams' persistent kernels."""
        with torch.no_grad():
            snap = (self.flat_params.tensor.detach().clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.seed_dev.clone())
        sched = dict(self.sched_state)
        loss = self.step(i)
        if not self.check_faults(on_fault="return"):
            return loss
        with torch.no_grad():
            self.flat_params.tensor.copy_(snap[0])
            self.exp_avg.copy_(snap[1])
            self.exp_avg_sq.copy_(snap[2])
            self.seed_dev.copy_(snap[3])
            self.sync_shadows()
        self.sched_state = sched
        self._set_lr()
        ops.SAFE_FORMS[0] = True
        ops.set_peer_wait_limit(0, 0)
        self.recapture()
        self.faults_recovered = getattr(self, "faults_recovered", 0) + 1
        loss = self.step(i)
        self.check_faults(on_fault="raise")
        return loss

    def add_batch(self, batch):
        """Register one more static batch (a new shape bucket of EpochLoop) and, in graph mode after `prepare()`, capture
        its step graph now (the capture's eager warm-up pass leaves the step counter and the parameters untouched)."""
        self.batches.append(batch)
        i = len(self.batches) - 1
        try:
            self._warmup(i, check=True)                 # (the capture's warm-up pass and the parameter-set check in one)
        except Exception:
            self.batches.pop()
            raise
        if self.use_graph and self._prepared:
            if self.fused_opt:
                self._capture_with_opt(i)               # (the whole step as one graph: no forward / backward-only graph beside it)
            else:
                self.graphs[i] = self._capture(i, warm=False)
        return i

    def step(self, i):
        """One optimizer step on pre-collated batch i (model_fqandtoyo.py:1434-1478 + optimizer + scheduler)."""
        if self.use_graph:
            self._loss_ref = self._loss_slots[i % len(self.batches)]
        if self.one_graph and self.use_graph and getattr(self, "fused_opt", False):
            j = i % len(self.batches)
            if self.comm:
                self.graphs[j].replay()                         # (forward, backward, the captured exchange, AdamW)
            else:
                if j not in self.graphs_nocomm:                 # (bench.py: the same step without its exchange)
                    self._capture_with_opt(j, comm=False)
                self._loss_ref = self._loss_slots_nocomm[j]     # (the loss of the graph that is replayed)
                self.graphs_nocomm[j].replay()
            self.sched_state["step_count"] += 1
            self._set_lr()
            return self.loss_out
        if self.overlap:
            j = i % len(self.batches)
            na = self.n_head_elems
            comm = self.ddp and self.comm
            self.graphs[j].replay()
            cb = self.comm_buf
            # element ranges completed by phase A and by each part of phase B: every one is all-reduced (RCCL's stream) as soon
            # as the replay that completes it has been enqueued, i.e. beside the replay of the next part
            bounds = [(0, na)] + ([(e0, e1) for _, _, _, e0, e1 in self.parts] if self.parts else [(na, self.flat.flat.numel())])
            works = []

            def exchange(e0, e1):
                if comm and e1 > e0:
                    if cb is not None:
                        cb[e0:e1].copy_(self.flat.flat[e0:e1])
                    works.append(dist.all_reduce((cb if cb is not None else self.flat.flat)[e0:e1], op=dist.ReduceOp.SUM, async_op=True))
            exchange(*bounds[0])
            for gb, (e0, e1) in zip(self.graphs_b[j], bounds[1:]):
                gb.replay()
                exchange(e0, e1)
            if comm:
                for w in works:
                    w.wait()
                if cb is not None:
                    torch.mul(cb, 1.0 / self.world, out=self.flat.flat)          # widen + average in one pass
                else:
                    self.flat.flat.div_(self.world)
        else:
            if self.use_graph:
                self.graphs[i % len(self.batches)].replay()
            else:
                self._fwd_bwd(self.batches[i % len(self.batches)])
            if self.ddp and self.comm:
                self._exchange()
        if self.use_graph:
            if not getattr(self, "fused_opt", False):
                self.opt_graph.replay()
        else:
            self._opt_step()
        self.sched_state["step_count"] += 1
        self._set_lr()
        return self.loss_out


class EpochLoop:
    """The fit loop over the training set (Lightning's `trainer.fit` on `train_dataloader`: entry.py:141-161,
    data.py:282-295), one process per GPU: a NEW batch every step, collated on the device.

    * sampler: `data.shard_indices` = torch's DistributedSampler (shuffle by seed + epoch, wrap-around padding, stride by
      rank); consecutive runs of `batch_size` indices form the batches (`drop_last=False`, as the reference's DataLoader);
    * shapes: a batch's padded node count is rounded up to a bucket (`data.BUCKETS`); every (G, bucket) owns ONE static
      collated batch on the device, two staging buffers and ONE step graph = forward + loss + backward + AdamW, captured
      the first time the bucket occurs;
    * a step on the host: pack the NEXT batch's raw trajectories into a pinned staging buffer and, on the copy stream, start
      its host-to-device copy and its device collate (`DeviceCollator.finish_into`: SPD / edge paths / degrees / distance
      bins) while the GPU runs the current step; then -- on the compute stream -- one device-to-device copy of the staged
      raw + derived bytes into the bucket's static batch and one graph replay.  Collators whose finish needs torch ops
      (coordinate bins) keep the collate inside the step graph (`batch_fn = finish`).
    """

    def __init__(self, model, collator, dataset, batch_size=16, seed=1, use_graph=True, overlap=True, buckets=None, rank=None,
                 world=None, shuffle=True, autocast_dtype=None, side_collate=True, balance=None, balance_window=32):
        """`balance`: deal every step's batches by length over the ranks (`data.balanced_batches`: neighbouring shape buckets on
        all ranks in every synchronous step, the epoch's sample set still DistributedSampler's).  Default: on when there is more
        than one rank; with one rank the order is the reference's (DistributedSampler order, consecutive batches).
        `balance_window`: the length sort runs inside windows of that many steps of the permuted epoch (default 32: a
        mega-batch of 32 x world x batch_size i.i.d. samples -- batches stay mixed over the epoch and its long graphs are spread
        over the windows; DEVIATION from the reference's i.i.d. batches, bounded by the window; None = sort the whole epoch, the
        tightest balance; balance=False = the reference's sampler order exactly)."""
        from .data import BUCKETS
        self.model, self.collator, self.dataset = model, collator, dataset
        self.batch_size, self.seed, self.shuffle = int(batch_size), int(seed), shuffle
        self.buckets = tuple(buckets or BUCKETS)
        ddp = dist.is_available() and dist.is_initialized()
        self.rank = rank if rank is not None else (dist.get_rank() if ddp else 0)
        self.world = world if world is not None else (dist.get_world_size() if ddp else 1)
        self.balance = (self.world > 1) if balance is None else bool(balance)
        self.balance_window = balance_window
        self._lengths = None
        self.device = next(model.parameters()).device
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = {}
        self.ts = None
        self._ts_args = dict(autocast_dtype=autocast_dtype, use_graph=use_graph, seed=seed, overlap=overlap)
        self.steps_done = 0
        self.limits = None
        self.side_collate = bool(side_collate)
        # (round 4 measured and round 5 removed: one step graph per STAGING buffer, i.e. no device-to-device copy between two
        #  replays -- 0.734 against 0.700 ms per step on the S-FSQ pool: two graphs per bucket alternate between two sets of
        #  activation buffers, and the copy was not what the loop waited for)

    # ---- data order -----------------------------------------------------------------------------------------------------
    def batches_of_epoch(self, epoch):
        from .data import balanced_batches, shard_indices
        if self.balance:
            if self._lengths is None:
                self._lengths = [len(t["node_name"]) for t in self.dataset]
            steps = balanced_batches(self._lengths, self.world, self.batch_size, epoch=epoch, seed=self.seed, shuffle=self.shuffle,
                                     buckets=self.buckets, window=self.balance_window)
            return [s[self.rank] for s in steps]
        idx = shard_indices(len(self.dataset), self.rank, self.world, epoch=epoch, seed=self.seed, shuffle=self.shuffle)
        B = self.batch_size
        return [idx[i:i + B] for i in range(0, len(idx), B)]

    # ---- buckets --------------------------------------------------------------------------------------------------------
    def _slot(self, G, N):
        from .data import BatchLayout, RawLayout
        key = (G, N)
        s = self.slots.get(key)
        if s is None:
            # `side`: the device collate of the NEXT batch runs on the copy stream, beside the current step (whose kernels
            # leave most of the chip idle); the step graph then reads a collated static batch.  Collators that need torch
            # ops (coordinate bins, S-BIG) keep the collate inside the step graph (batch_fn = finish).
            side = self.side_collate and self.collator.can_finish_into()
            lay = BatchLayout(G, N, self.collator.D) if side else RawLayout(G, N)
            buf = torch.zeros(lay.nbytes, dtype=torch.uint8, device=self.device)
            views = lay.views_torch(buf)
            s = dict(layout=lay, buf=buf, views=views, index=None, stages=[], turn=0, side=side,
                     batch=self.collator.batch_from_views(views) if side else views,
                     copy_bytes=lay.copy_bytes if side else lay.nbytes)
            raw_bytes = lay.raw_bytes if side else lay.nbytes
            for _ in range(2):
                pin = torch.zeros(raw_bytes, dtype=torch.uint8).pin_memory()
                dev = torch.zeros(lay.nbytes, dtype=torch.uint8, device=self.device)
                dv = lay.views_torch(dev) if side else None
                s["stages"].append(dict(pin=pin, np=lay.views_np(pin.numpy()), dev=dev, dev_views=dv,
                                        work=None, ready=torch.cuda.Event(), free=None, used=False, index=None,
                                        batch=None))
            self.slots[key] = s
        return s

    def _check_host(self, h):
        """What nn.Embedding would raise on in the reference (IndexError), checked on the host arrays of a fresh batch --
        the captured step cannot look at values (model.validate_batch does this for pre-collated batches)."""
        m = self.model
        if self.limits is None:
            # (the model says which fields it indexes tables with -- model.index_limits(); a model without the hook is not checked)
            self.limits = m.index_limits() if hasattr(m, "index_limits") else {}
        L = self.limits
        nz = h["counts"] != 0
        bad = None
        if "x" in L and int(h["x"].max()) > L["x"]:
            bad = ("x", int(h["x"].max()), L["x"] + 1)
        elif "user" in L and int(h["user"].max()) > L["user"]:
            bad = ("user", int(h["user"].max()), L["user"] + 1)
        elif "y" in L and int(h["y"].max()) > L["y"]:
            bad = ("y", int(h["y"].max()), L["y"] + 1)
        elif "edge" in L and int(h["counts"].max()) + 3 >= L["edge"]:
            bad = ("edge_input", int(h["counts"].max()) + 3, L["edge"])
        elif "deg" in L and max(int(nz.sum(1).max()), int(nz.sum(2).max())) + 1 >= L["deg"]:
            bad = ("degree", max(int(nz.sum(1).max()), int(nz.sum(2).max())) + 1, L["deg"])
        elif "slots" in L and int(float(h["time_normal"].max()) * 48) >= L["slots"]:
            bad = ("time_normal", float(h["time_normal"].max()), L["slots"])
        if bad:
            raise IndexError(f"batch.{bad[0]} has index {bad[1]}, out of range for a table of {bad[2]} rows")

    def _stage(self, ids):
        """Host half of a step's input: raw trajectories -> the bucket's pinned buffer -> async copy to a device staging
        buffer on the copy stream.  Returns (slot, stage)."""
        from .data import bucket_nodes
        trajs = [self.dataset[i] for i in ids]
        trajs = [t for t in trajs if t is not None and len(t["node_name"]) <= self.collator.max_node]
        G = len(trajs)
        if G == 0:
            return None                                # (every trajectory filtered out: the reference's collator skips such a batch too, collator.py:313)
        N = bucket_nodes(max(len(t["node_name"]) for t in trajs), self.buckets)
        slot = self._slot(G, N)
        st = slot["stages"][slot["turn"]]
        slot["turn"] ^= 1
        if st["free"] is not None:
            st["free"].synchronize()                   # its previous device-to-device copy has been executed
        self.collator.pack_host(trajs, idx0=ids[:G] if len(ids) == G else 0, n_pad=N, out=st["np"])
        self._check_host(st["np"])
        st["used"] = True
        with torch.cuda.stream(self.copy_stream):
            st["dev"][:st["pin"].numel()].copy_(st["pin"], non_blocking=True)
            if slot["side"]:
                st["work"] = self.collator.finish_into(st["dev_views"], st["work"])
            st["ready"].record(self.copy_stream)
        return slot, st

    def _ensure_trainer(self, slot):
        """The first batch builds the TrainStep (dry run for the trained-parameter set, flat buffers, optimizer graph)."""
        if self.ts is None:
            self.ts = TrainStep(self.model, [slot["batch"]], batch_fn=self._batch_fn, **self._ts_args)
            self.ts.prepare()
            slot["index"] = 0
        elif slot["index"] is None:
            slot["index"] = self.ts.add_batch(slot["batch"])

    def _batch_fn(self, b):
        """inside the step: raw views -> collate (in-graph form); an already collated static batch passes through"""
        return self.collator.finish(b) if isinstance(b, dict) else b

    def _launch(self, slot, st):
        cur = torch.cuda.current_stream()                # (graphs replay on the current stream)
        cur.wait_event(st["ready"])
        n = slot["copy_bytes"]
        slot["buf"][:n].copy_(st["dev"][:n], non_blocking=True)
        if st["free"] is None:
            st["free"] = torch.cuda.Event()
        st["free"].record(cur)
        if slot["index"] is None or self.ts is None:
            self._ensure_trainer(slot)                   # (dry run / capture on the data that is now in the static buffer)
        return self.ts.step(slot["index"])

    def run_epoch(self, epoch=0, max_steps=None, on_step=None):
        """One pass over this rank's shard.  Returns dict(steps, graphs, sample_ids); `on_step(step, loss_tensor)` is called
        after each launch (reading the loss there synchronises -- do it sparingly)."""
        batches = self.batches_of_epoch(epoch)
        if max_steps is not None:
            batches = batches[:max_steps]
        seen = []
        if not batches:
            return dict(steps=0, graphs=len(self.slots), sample_ids=seen)
        nxt = self._stage(batches[0])
        steps = 0
        for j, ids in enumerate(batches):
            cur = nxt
            if cur is not None:
                loss = self._launch(*cur)                # asynchronous: the GPU works on step j ...
            nxt = self._stage(batches[j + 1]) if j + 1 < len(batches) else None          # ... while the host packs j + 1
            seen.extend(ids)
            if cur is None:
                continue                                 # (an empty batch: skipped, as the reference's collator does)
            steps += 1
            self.steps_done += 1
            if on_step is not None:
                on_step(self.steps_done, loss)
        # peer waits that gave up during the epoch (the kernels count and carry on, csrc/chain.hip WS_FAULT): raise here, once
        # per epoch, rather than train on silently (one device synchronisation; every rank takes the same decision)
        if self.ts is not None:
            self.ts.check_faults(on_fault="raise")
        return dict(steps=steps, graphs=len(self.slots), sample_ids=seen)

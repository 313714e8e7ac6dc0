"""LONG-BATCH gradient parity against the oracle (VERDICT r5 "weak" #1 / "next" #1a).

Past 4 096 rows an fq `EncoderLayer` (model_fqandtoyo.py:1714-1743; autograd of model.py:393-405, 436-457, 479-489) runs other
kernels than the short batches whose gradients the goldens pin: the 64-row chain kernels (`layer_chain_fwd_big_kernel` /
`layer_chain_bwd_big_kernel`, the latter hosting the tail `dx1 + dqkv Wqkv` of the layer above and `b1`'s column sums), the
long-batch weight-gradient path, the one-pass attention backward (`attn_bwd_one_kernel` + dQ accumulator), one bf16 dBias slice per
layer and the long form of `build_bias_bwd`.  Round 5 compared those with OTHER KERNELS of this repository; here they are
compared, as a composition, with `oracle.encoder_layer_fq` / `oracle.assemble_bias` gradients:

  * a 3-layer fq stack, C 256 (d 32) and C 192 (d 24), G 6 x T 785 = 4 710 rows (73 full 64-row blocks + 38 rows), ragged graphs
    (784 / 784 / 700 / 784 / 512 / 784 nodes: -inf key columns), TRAINING mode with every dropout mask replayed into the oracle;
    the bias comes from the device's bias assembly over the model's tables (bf16 pack), so dBias flows slices -> build_bias_bwd;
    compared: y, dx, dBias [G,H,T,T], every parameter gradient of the three layers, the five bias-table gradients;
  * the same batch through `train.TrainStep` (hipGraph replay, flat gradient buffer + sinks, the parked split-K sums): loss and
    the gradients of the whole model against one oracle training step with the device's masks.

Tolerances are the bf16 gates of tests/test_gpu_train_parity.py (`check_grad`: relative L2 <= 1.5 % -- 3 % on the small cancelling
bias tables --, 99.9 % of the entries within 0.13 rms + 0.05 |ref|, every entry within 4 x that, exact zero pattern); y / dx /
dBias by their largest error against the rms of the reference's non-zero entries, bounds at the assertions.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mobgt_amd import ops, synth, workloads                                        # noqa: E402
from oracle import model_oracle as mo                                               # noqa: E402
from test_gpu_bench_parity import LOSS_SCALE, bad_rows, check_grad, cpu_batch      # noqa: E402
from test_gpu_train_parity import _drop_hook, layer_masks, step_masks              # noqa: E402
from gradcheck import device_head_pattern, replay_head                            # noqa: E402

DEV = "cuda"
EDGE_TABLE_REL_L2 = 1e-1          # relative L2 of the two edge tables (see the assertions: 0.5-0.8 % under an O(1) gradient, 7.3-7.8 % under the real loss)
N_NODES = [784, 784, 700, 784, 512, 784]
BIAS_TABLES = ("rel_pos_encoder.weight", "poi_pos_encoder.weight", "edge_encoder.weight", "edge_dis_encoder.weight",
               "graph_token_virtual_distance.weight")


def _build(hidden, P=2000, seed=1, **over):
    """The S-FSQ model class at S-BIG's layer widths on a P = 2 000 universe the oracle can hold densely; three layers."""
    ov = dict(n_layers=3, hidden_dim=hidden, warmup_updates=4, tot_updates=100, peak_lr=2e-3)
    ov.update(over)
    uni, model, coll = workloads.build("fsq", DEV, seed=seed, P=P, model_overrides=ov)
    trajs = synth.make_batch_of_trajectories(seed=seed + 10, G=len(N_NODES), P=P, n_user=1080, cat_of_poi=uni.cat_of_poi,
                                             n_nodes=N_NODES)
    return uni, model, coll(trajs)


def _graph_slice(cb, g):
    c = SimpleNamespace()
    for f, t in vars(cb).items():
        setattr(c, f, t[g:g + 1])
    return c


def oracle_bias_table_grads(sd_tables, cb, dbias, H=8, D=20, scale=1.0, fp16_roundtrip=True):
    """Gradients of the five bias tables for a given d loss / d bias [G,H,T,T]: `oracle.assemble_bias` graph by graph (the
    [G,N,N,D,H] intermediates of model_fqandtoyo.py:1178-1198 are 400 MB per 784-node graph; the assembly is independent per graph).
    `scale`: the loss scale the backward runs at (the reference's own .half() casts flush per-pair gradients below 6e-8 and overflow
    above 65 504: GradScaler's 65 536 for a real loss, 1 for the O(1) gradients of the stack test).
    `fp16_roundtrip=False`: the same gradients WITHOUT the reference's fp16 casts (exact fp32 arithmetic) -- the yardstick for how
    much of a deviation is the reference's own rounding."""
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in sd_tables.items()}
    orig = mo._edge_term
    if not fp16_roundtrip:
        mo._edge_term = lambda sd_, rp, ei, H_, D_, _f: orig(sd_, rp, ei, H_, D_, False)
    try:
        for g in range(dbias.shape[0]):
            b = mo.assemble_bias(sd, _graph_slice(cb, g), H, D, "fq")
            b.backward(dbias[g:g + 1] * scale)
    finally:
        mo._edge_term = orig
    return {k: v.grad / scale for k, v in sd.items()}


def _close(name, got, want, tol):
    got, want = got.detach().float().cpu().numpy(), want.detach().numpy()
    nz = want != 0
    assert np.all(got[~nz] == 0), name
    scale = float(np.sqrt((want[nz].astype(np.float64) ** 2).mean()))
    err = float(np.abs(got - want).max())
    rel = float(np.sqrt(((got - want).astype(np.float64) ** 2).sum() / (want.astype(np.float64) ** 2).sum()))
    print("%-10s max|err| %.3e  rms %.3e  ratio %.4f  relL2 %.5f" % (name, err, scale, err / scale, rel))
    assert err <= tol * scale, f"{name}: max |err| {err:.3e} vs rms {scale:.3e}"
    return rel


@pytest.mark.parametrize("hidden", [192, 128])
def test_long_batch_three_layer_stack_gradients_vs_oracle(hidden):
    from mobgt_amd.model import refresh_shadows
    uni, model, batch = _build(hidden)
    G, N = batch.x.shape[:2]
    T, H, C, p = N + 1, 8, hidden + 64, 0.1
    assert (G, T) == (6, 785) and G * T > 4096 and (G * T) % 64 != 0
    layers = model.layers
    model.train()
    step = 11
    seed_dev = torch.tensor([step], dtype=torch.int64, device=DEV)
    for l in layers:
        l.self_attention.seed_dev = seed_dev
        for prm in l.parameters():                 # LayerNorm weights / biases away from (1, 0) so that their gradients matter
            if prm.dim() == 1:
                prm.data.add_(0.1 * torch.randn_like(prm))
    rng = np.random.RandomState(5)
    x0 = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))
    # ---- device: the model's own bias assembly (bf16 pack, gradient back into the tables), then the stack as forward() runs it
    for q in model.parameters():
        q.grad = None
    pack = model.assemble_bias(batch)
    assert pack.sliced and pack.needs_grad
    xd = x0.to(DEV).requires_grad_(True)
    refresh_shadows(layers, rows=G * T)
    y = xd
    for li, l in enumerate(layers):
        y = l(y, pack, mask=None, next_layer=layers[li + 1] if li + 1 < len(layers) else None)
        if li + 1 < len(layers):
            assert getattr(y, "_mobgt_qkv", None) is not None            # the 64-row forward chain carried the next QKV projection
    assert y.grad_fn.chain_bwd                                           # ... and the 64-row backward chain is what will run
    y.backward(gy.to(DEV))
    torch.cuda.synchronize()
    assert pack.n_bwd == 3 and pack.dbias.dtype == torch.bfloat16        # one bf16 dBias slice per layer
    dbias_dev = pack.grad_total().cpu()
    # ---- oracle: the bias of the same tables in fp32, the three layers with the device's masks
    cb = cpu_batch(batch)
    sd_all = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    sd_tab = {k: sd_all[k] for k in BIAS_TABLES}
    with torch.no_grad():
        bias = torch.cat([mo.assemble_bias(sd_tab, _graph_slice(cb, g), H, 20, "fq") for g in range(G)])
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_all.items() if k.startswith("layers.")}
    masks = {}
    for li, l in enumerate(layers):
        masks.update(layer_masks(f"layers.{li}", l.self_attention, step, G, T, C, H, p, p))
    xr, br = x0.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = xr
    for li in range(len(layers)):
        ref = mo.encoder_layer_fq(sd, f"layers.{li}", ref, br, H, p, p, True, drop=_drop_hook(masks))
    ref.backward(gy)
    # three post-LN layers deep, bf16 GEMM operands: y rows have rms ~1 (measured: see the printout; the one-layer gate of
    # test_gpu_train_parity is 6e-2 / 8e-2 / 1.5e-1, its two-layer stock stack 8e-2 / 1e-1 / 2e-1)
    _close("y", y, ref, 8e-2)
    _close("dx", xd.grad, xr.grad, 1.2e-1)
    # dBias [G,H,T,T] is heavy-tailed (dS = P (dP - delta): large where a probability is large; max |ref| is ~300 x its rms), and
    # every entry passed a bf16 rounding per layer: relative L2 <= 1.5 % (measured 0.74-0.76 %), 99.9 % of the 29.6 M entries within
    # half of 0.13 rms + 0.05 |ref| (measured 0.24), every entry within 8 x that bound (measured 2.5 / 4.3: the maximum over 3e7
    # entries), exact zeros at the -inf key columns
    report, ok = [], True
    check_grad("dbias [G,H,T,T]", dbias_dev, br.grad, report)
    r = report[0]
    print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
    assert r[2] <= 1.5e-2 and r[3] <= 0.5 and r[4] <= 8.0 and r[5] == 0.0, r
    report = []
    for k, prm in layers.named_parameters():
        want = sd["layers." + k].grad
        if want is None:
            assert prm.grad is None or float(prm.grad.abs().sum()) == 0.0, k
            continue
        if k.endswith("linear_k.bias"):
            continue            # exactly 0 in exact arithmetic (softmax is shift-invariant over keys): round-off on both sides
        ok &= check_grad("layers." + k, prm.grad, want, report)
    # the bias tables: d loss / d bias of the ORACLE pushed through oracle.assemble_bias (the device's went through its own bf16
    # slices and build_bias_bwd's long form)
    tab = oracle_bias_table_grads(sd_tab, cb, br.grad)
    exact = oracle_bias_table_grads(sd_tab, cb, br.grad, fp16_roundtrip=False)
    params = dict(model.named_parameters())
    for k in BIAS_TABLES:
        if k.startswith("edge_"):
            continue
        ok &= check_grad(k, params[k].grad, tab[k], report)
    # the two edge tables: every entry of their gradient is a 1/spd-weighted sum of dBias over (nearly) ALL 3.7 M pairs x 20 hops of
    # the batch -- and sum_j dS_ij = 0 on every row, so the sum cancels almost completely while the bf16 rounding noise of the dS
    # entries (one rounding per layer) does not.  Judged by relative L2 against the reference's gradient (fp16 rounding points of
    # model_fqandtoyo.py:1178-1198 in its backward) AND against the same gradient in exact fp32 arithmetic, printed side by side
    # with the distance between those two -- the reference's own rounding
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
    for k in ("edge_encoder.weight", "edge_dis_encoder.weight"):
        g = params[k].grad.detach().float().cpu()
        e_ref, e_exact, own = rel(g, tab[k]), rel(g, exact[k]), rel(tab[k], exact[k])
        print("%-28s relL2 vs reference %.4f  vs exact fp32 %.4f   reference vs exact %.4f   rms %.3e" % (
            k, e_ref, e_exact, own, float(tab[k].pow(2).mean().sqrt())))
        assert torch.equal(g == 0, tab[k] == 0) or float(g[tab[k] == 0].abs().max()) <= 1e-3 * float(tab[k].abs().max()), k
        assert min(e_ref, e_exact) <= EDGE_TABLE_REL_L2, (k, e_ref, e_exact)
    for r in report:
        print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
    assert ok, bad_rows(report)


def _checkpointed_bias(orig):
    """oracle.assemble_bias graph by graph under activation checkpointing: same values and gradients, ~1/G of the memory (the
    [G,N,N,D,H] intermediates of model_fqandtoyo.py:1178-1198 are recomputed per graph in the backward)."""
    from torch.utils.checkpoint import checkpoint

    def bias(sd, batch, H, D, variant):
        tabs = [sd[k] for k in BIAS_TABLES]

        def one(g, *t):
            return orig(dict(zip(BIAS_TABLES, t)), _graph_slice(batch, g), H, D, variant)
        return torch.cat([checkpoint(one, g, *tabs, use_reentrant=False) for g in range(batch.rel_pos.shape[0])])
    return bias


def test_long_batch_train_step_vs_oracle_with_replayed_masks(monkeypatch):
    """The long-batch step AS THE TRAINER RUNS IT -- `train.TrainStep`: one hipGraph, flat gradient buffer and sinks, the parked
    split-K sums of the weight gradients (`mobgt_partial_sum_multi`), the 64-row chain kernels hosting the upper layer's tail,
    one-pass attention backward, bf16 dBias slices into `build_bias_bwd` riding wherever the step puts it -- against ONE oracle
    training step (model_fqandtoyo.py:1434-1478) with the device's dropout masks and head branch pattern: loss and EVERY
    parameter gradient (three layers, C 256, 4 710 rows, dropout on)."""
    from mobgt_amd.train import TrainStep
    uni, model, batch = _build(192)
    G, N = batch.x.shape[:2]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    consts = mo.fq_constants(uni, "foursquaregraph", diag_inverse=True, num_bins=model.poi_pos_encoder.num_embeddings)
    host_seed = 5
    ts = TrainStep(model, [batch], use_graph=True, seed=host_seed, keep_head_rows=True)
    ts.prepare()
    with torch.no_grad():
        model.load_state_dict(sd0)
        ts.sync_shadows()
    loss = float(ts.step(0))
    assert not ops.step_state_leftovers()
    step = int(ts.seed_dev.item())
    masks = step_masks(model, batch, step, host_seed)
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in sd0.items()}
    cb = cpu_batch(batch)
    pattern, pre_dev = device_head_pattern(model, batch, enc_out=ts.enc_outs[0], state=sd0)
    seen = {}
    monkeypatch.setattr(mo, "assemble_bias", _checkpointed_bias(mo.assemble_bias))
    ref_loss = mo.fq_training_loss(sd, cb, consts, n_layers=3, H=8, D=20, p=0.1, p_in=0.1, p_att=0.1, training=True,
                                   hidden=model.hidden_dim, drop=_drop_hook(masks), act=replay_head(pattern, seen, pre_dev))
    from gradcheck import assert_replay_bounded
    print("replayed head units: %d, largest |pre| %.2e" % assert_replay_bounded(seen))
    (ref_loss * LOSS_SCALE).backward()
    print("loss hip %.7f  oracle %.7f" % (loss, float(ref_loss)))
    np.testing.assert_allclose(loss, float(ref_loss), rtol=3e-3)
    params = dict(model.named_parameters())
    report = []
    for k, v in sd.items():
        if v.grad is None or k not in params or params[k].grad is None or k.endswith("linear_k.bias"):
            continue
        check_grad(k, params[k].grad, v.grad / LOSS_SCALE, report)
    assert len(report) > 60, len(report)
    for r in report:
        print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
    # the gates of test_gpu_train_parity's S-FSQ / S-GOW step: 2 % relative L2 (6 % on the two edge tables, whose gradient passes
    # the reference's own fp16 rounding points and whose few hundred entries are judged by relative L2 and the zero pattern only)
    edge = lambda r: r[0].startswith("edge_")
    bad = [r for r in report if (not edge(r) and (r[2] > 2e-2 or r[3] > 3.0 or r[4] > 6.0)) or r[5] > 1e-3 * r[1]]
    assert not bad, bad
    # the two edge tables at this batch length (see the stack test above: each entry is a cancelling sum over ~3.7 M pairs x 20 hops):
    # against the reference's gradient and against the same step WITHOUT the reference's fp16 casts (exact fp32 arithmetic)
    sd_x = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in sd0.items()}
    orig_edge = mo._edge_term
    monkeypatch.setattr(mo, "_edge_term", lambda sd_, rp, ei, H_, D_, _f: orig_edge(sd_, rp, ei, H_, D_, False))
    loss_x = mo.fq_training_loss(sd_x, cb, consts, n_layers=3, H=8, D=20, p=0.1, p_in=0.1, p_att=0.1, training=True,
                                 hidden=model.hidden_dim, drop=_drop_hook(masks), act=replay_head(pattern, {}, pre_dev))
    loss_x.backward()
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))
    for k in ("edge_encoder.weight", "edge_dis_encoder.weight"):
        g = params[k].grad.detach().float().cpu()
        ref_g, ex_g = sd[k].grad / LOSS_SCALE, sd_x[k].grad
        e_ref, e_exact, own = rel(g, ref_g), rel(g, ex_g), rel(ref_g, ex_g)
        print("%-28s relL2 vs reference %.4f  vs exact fp32 %.4f   reference vs exact %.4f" % (k, e_ref, e_exact, own))
        assert min(e_ref, e_exact) <= EDGE_TABLE_REL_L2, (k, e_ref, e_exact)


def test_long_batch_step_forms_agree():
    """The two round-6 forms of the long-batch step against the forms they replaced, on the same model and batch (dropout off, so
    nothing but the summation order differs): the layer's weight gradients as ONE launch (csrc/wgradbig.hip; fused_layer._WGRAD_BIG off:
    library split-K products + `wgrad_group` + parked partial sums) and the bias tables' backward on the side stream beside the
    tail of the backward pass (ops._bias_bwd_beside; ops._BIAS_BWD_BESIDE off: where autograd reaches it).  One captured
    `TrainStep` each; every parameter gradient of the model."""
    import gc
    from mobgt_amd import fused_layer
    from mobgt_amd.train import TrainStep
    grads, losses = {}, {}
    for mode in ("round6", "before"):
        fused_layer._WGRAD_BIG[0] = ops._BIAS_BWD_BESIDE[0] = mode == "round6"
        try:
            uni, model, batch = _build(192, dropout_rate=0.0, intput_dropout_rate=0.0, attention_dropout_rate=0.0)
            sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
            ts = TrainStep(model, [batch], use_graph=True, seed=5)
            ts.prepare()
            with torch.no_grad():
                model.load_state_dict(sd0)                    # (the warm-up steps of prepare() are real AdamW updates)
                ts.sync_shadows()
            losses[mode] = float(ts.step(0))
            torch.cuda.synchronize()
            assert not ops.step_state_leftovers()
            grads[mode] = {n: q.grad.detach().float().clone() for n, q in model.named_parameters() if q.grad is not None}
            del ts, model, batch, uni
            gc.collect()
        finally:
            fused_layer._WGRAD_BIG[0] = ops._BIAS_BWD_BESIDE[0] = True
    assert losses["round6"] == pytest.approx(losses["before"], rel=1e-5)
    a, b = grads["round6"], grads["before"]
    assert a.keys() == b.keys() and len(a) > 60
    worst = ("", 0.0)
    for n in a:
        if n.endswith("linear_k.bias"):
            continue                                          # exactly zero in exact arithmetic: round-off only
        ref = b[n].double()
        rel = float((a[n].double() - ref).norm() / ref.norm().clamp_min(1e-300))
        worst = max(worst, (n, rel), key=lambda t: t[1])
        # (two runs of ONE form already differ by up to 2.1e-3 -- 3.5 % on the two edge tables, whose sums cancel 30-70 x --: float
        #  atomics in front of bf16 rounding points, profiles/r6_backward_run_to_run.txt; most pairs of runs agree to 9e-5)
        assert rel <= (1e-1 if n.startswith("edge_") else 1e-2), (n, rel)
    print("largest relative L2 between the forms: %s %.2e" % worst)

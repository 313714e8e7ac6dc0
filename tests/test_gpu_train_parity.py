"""TRAINING-mode parity against the oracle with every dropout mask replayed (VERDICT r2 "missing" #6).

The device draws its dropout masks from a counter-based rule (csrc/common.h) that the library also states on the host
(`mobgt_attn_dropout_mask_host`, `mobgt_dropout_mask_host`).  The tests rebuild the masks of every site of
`model_fqandtoyo.py:358, 1347, 1364, 1700, 1735-1741`, `modelGNN.py:72` for the seeds / salts / row numbering the
product uses and hand them to the oracle through its `drop` hook, so that a dropout-ON layer and a dropout-ON train step
are compared element by element instead of through statistics or self-comparisons:

  * one fq EncoderLayer (C 192, ffn 1024, 8 heads), forward + every gradient, in the three forms the product has:
    fp32 separate launches, bf16 separate launches, bf16 chain kernels (fwd + bwd);
  * one S-FSQ train step at the benchmark's configuration (P = 7856, 6 layers, bf16, hipGraph replay through TrainStep):
    loss and elementwise gradients.

Tolerances: fp32 layer: 3e-2 / 5e-2 / 8e-2 of the reference's rms over its non-zero entries (attention MFMA operands are bf16 in every configuration);
bf16 layer: 6e-2 rms on outputs, gradients by `check_grad` of test_gpu_bench_parity (|err| <= 0.12 rms + 0.05 |ref| for
99.9 % of the entries, relative L2 <= 3e-2); train step: loss rtol 3e-3, gradients by the same measure with the allowance
stated at the assertion.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mobgt_amd import ops, workloads                                     # noqa: E402
from oracle import model_oracle as mo                                     # noqa: E402
from gradcheck import assert_replay_bounded, device_head_pattern, n_flipped, replay_head                  # noqa: E402
from test_gpu_bench_parity import GRAD_PARAMS, LOSS_SCALE, bad_rows, check_grad, cpu_batch, oracle_consts   # noqa: E402,F401

DEV = "cuda"
M64 = (1 << 64) - 1


def _inv_keep(p):
    return 1.0 / (1.0 - int(p * 65536 + 0.5) / 65536.0)


def _drop_hook(masks):
    def drop(x, p, site):
        keep = masks[site]
        return x * keep.to(x.dtype) * _inv_keep(p)
    return drop


def layer_masks(prefix, mha, step, G, T, C, H, p, p_att):
    """The masks of one fused encoder layer (fused_layer.py: attention seed = layer seed ^ (salt * 0x9E3779B1), residual
    sites salt + 1 / salt + 2 with rows g*T + t), keyed by the oracle's site names."""
    salt = mha._layer_index * 8
    seed = (mha._seed_salt + step) & M64
    seed_att = ((mha._seed_salt ^ (salt * 0x9E3779B1)) + step) & M64
    return {
        (prefix + ".self_attention", "att"): torch.from_numpy(ops.dropout_keep_mask(seed_att, G, H, T, p_att)),
        (prefix, "res1"): torch.from_numpy(ops.dropout_site_mask(seed, salt + 1, G * T, C, p)).view(G, T, C),
        (prefix, "res2"): torch.from_numpy(ops.dropout_site_mask(seed, salt + 2, G * T, C, p)).view(G, T, C),
    }


@pytest.mark.parametrize("mode", ["f32", "bf16_launches", "bf16_lngemm_bwd", "bf16_chain"])
def test_fq_layer_with_dropout_on_vs_oracle_with_replayed_masks(mode, monkeypatch):
    """(bf16_lngemm_bwd: the opt-in MOBGT_LN_GEMM_BWD=1 form, LayerNorm' / dropout' as the prologue of the backward GEMMs)"""
    from mobgt_amd import fused_layer
    from mobgt_amd.model import refresh_shadows
    from mobgt_amd.model_fqandtoyo import EncoderLayer
    G, H, T, C, F, p, p_att = 4, 8, 53, 192, 1024, 0.1, 0.1
    monkeypatch.setattr(fused_layer, "_CHAIN", [mode == "bf16_chain"])
    monkeypatch.setattr(fused_layer, "_CHAIN_BWD", [mode == "bf16_chain"])
    monkeypatch.setattr(fused_layer, "_LN_GEMM_BWD", [mode == "bf16_lngemm_bwd"])
    torch.manual_seed(0)
    layer = EncoderLayer(C, F, p, p_att, H)
    layer.self_attention.set_layer_index(1)       # (stand-alone layers count process-wide: pin it, so that the masks -- and with
                                                  #  them the max-statistics below -- do not depend on what ran before)
    for prm in layer.parameters():                 # LayerNorm weights / biases away from (1, 0) so that their gradients matter
        if prm.dim() == 1:
            prm.data.add_(0.1 * torch.randn_like(prm))
    sd = {"L." + k: v.detach().clone().requires_grad_(True) for k, v in layer.state_dict().items()}
    rng = np.random.RandomState(1)
    x = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))
    bias = torch.from_numpy((rng.standard_normal((G, H, T, T)) * 0.5).astype(np.float32))
    bias[1, :, :, 40:] = float("-inf")
    bias[3, :, :, 7:] = float("-inf")
    step = 11
    layer = layer.to(DEV).train()
    layer.fused = True
    layer.act_dtype = torch.float32 if mode == "f32" else torch.bfloat16
    seed_dev = torch.tensor([step], dtype=torch.int64, device=DEV)
    layer.self_attention.seed_dev = seed_dev
    masks = layer_masks("L", layer.self_attention, step, G, T, C, H, p, p_att)
    for m in masks.values():
        assert abs(1.0 - float(m.float().mean()) - 0.1) < 0.02
    xr, br = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = mo.encoder_layer_fq(sd, "L", xr, br, H, p, p_att, True, drop=_drop_hook(masks))
    ref.backward(gy)
    xd, bd = x.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    refresh_shadows([layer])                       # bf16 shadows (+ MFMA-order packs for the chain kernels)
    out = layer(xd, bd)
    out.backward(gy.to(DEV))
    torch.cuda.synchronize()
    if mode == "bf16_chain":
        assert layer._packed is not None           # the chain path was eligible ...
    # (dbias: the largest of ~90 k elements against the rms; 0.075-0.123 over different draws of the masks)
    tol_y, tol_dx, tol_db = (3e-2, 5e-2, 1.5e-1) if mode == "f32" else (6e-2, 8e-2, 1.5e-1)

    def close(name, got, want, tol):
        got, want = got.detach().float().cpu().numpy(), want.detach().numpy()
        nz = want != 0                                 # (dbias: the -inf key columns of graphs 1 and 3 are exact zeros)
        assert np.all(got[~nz] == 0), name
        scale = float(np.sqrt((want[nz] ** 2).mean()))
        err = float(np.abs(got - want).max())
        print("%-36s max|err| %.3e  rms %.3e" % (name, err, scale))
        assert err <= tol * scale, f"{name}: max |err| {err:.3e} vs rms {scale:.3e}"
    close("y", out, ref, tol_y)
    close("dx", xd.grad, xr.grad, tol_dx)
    close("dbias", bd.grad, br.grad, tol_db)
    report, ok = [], True
    for k, prm in layer.named_parameters():
        want = sd["L." + k].grad
        if want is None:
            assert prm.grad is None or float(prm.grad.abs().sum()) == 0.0, k
            continue
        if k.endswith("linear_k.bias"):
            continue            # exactly 0 in exact arithmetic (softmax is shift-invariant over keys): round-off on both sides
        if mode == "f32":
            close("d" + k, prm.grad, want, 8e-2)
        else:
            ok &= check_grad(k, prm.grad, want, report)
    for r in report:
        print("%-40s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
    assert ok, bad_rows(report)


@pytest.mark.parametrize("mode", ["bf16_launches", "bf16_chain"])
def test_stock_two_layer_stack_with_dropout_on_vs_oracle_with_replayed_masks(mode, monkeypatch):
    """graphormer/model.py:463-489 -- the pre-LN EncoderLayer north_star names -- twice in a row (C 128, d 16, ffn 1024, 8 heads),
    dropout ON, against oracle.encoder_layer_stock with the device's masks replayed: as separate launches and as the round-4
    chain kernels (layer 0's launch applies layer 1's self_attention_norm and QKV projection; layer 1's norm backward is
    finished in layer 0's backward launch).  Outputs, dx, dbias and EVERY parameter gradient of both layers."""
    from mobgt_amd import fused_layer
    from mobgt_amd.model import EncoderLayer, refresh_shadows
    G, H, T, C, F, p, p_att = 4, 8, 53, 128, 1024, 0.1, 0.1
    monkeypatch.setattr(fused_layer, "_CHAIN", [mode == "bf16_chain"])
    torch.manual_seed(0)
    layers = torch.nn.ModuleList([EncoderLayer(C, F, p, p_att, H) for _ in range(2)])
    for li, layer in enumerate(layers):
        layer.self_attention.set_layer_index(li + 1)
        for prm in layer.parameters():             # LayerNorm weights / biases away from (1, 0) so that their gradients matter
            if prm.dim() == 1:
                prm.data.add_(0.1 * torch.randn_like(prm))
    sd = {"L." + k: v.detach().clone().requires_grad_(True) for k, v in layers.state_dict().items()}
    rng = np.random.RandomState(1)
    x = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))
    gy = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))
    bias = torch.from_numpy((rng.standard_normal((G, H, T, T)) * 0.5).astype(np.float32))
    bias[1, :, :, 40:] = float("-inf")
    bias[3, :, :, 7:] = float("-inf")
    step = 11
    layers = layers.to(DEV).train()
    seed_dev = torch.tensor([step], dtype=torch.int64, device=DEV)
    masks = {}
    for li, layer in enumerate(layers):
        layer.fused, layer.act_dtype = True, torch.bfloat16
        layer.self_attention.seed_dev = seed_dev
        masks.update(layer_masks(f"L.{li}", layer.self_attention, step, G, T, C, H, p, p_att))
    xr, br = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    ref = xr
    for li in range(2):
        ref = mo.encoder_layer_stock(sd, f"L.{li}", ref, br, H, p, p_att, True, drop=_drop_hook(masks))
    ref.backward(gy)
    xd, bd = x.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    refresh_shadows(layers)
    mid = layers[0](xd, bd, next_layer=layers[1])
    assert bool(getattr(mid, "_mobgt_preln", False)) == (mode == "bf16_chain")
    out = layers[1](mid, bd)
    assert bool(getattr(out.grad_fn, "stock_chain", False)) == (mode == "bf16_chain")
    out.backward(gy.to(DEV))
    torch.cuda.synchronize()

    def close(name, got, want, tol):
        got, want = got.detach().float().cpu().numpy(), want.detach().numpy()
        nz = want != 0
        assert np.all(got[~nz] == 0), name
        scale = float(np.sqrt((want[nz] ** 2).mean()))
        err = float(np.abs(got - want).max())
        print("%-36s max|err| %.3e  rms %.3e" % (name, err, scale))
        assert err <= tol * scale, f"{name}: max |err| {err:.3e} vs rms {scale:.3e}"
    close("y", out, ref, 8e-2)                      # (two layers deep; the fq one-layer gate is 6e-2)
    close("dx", xd.grad, xr.grad, 1e-1)
    close("dbias", bd.grad, br.grad, 2e-1)
    report, ok = [], True
    for k, prm in layers.named_parameters():
        want = sd["L." + k].grad
        assert want is not None and prm.grad is not None, k
        if k.endswith("linear_k.bias"):
            continue            # exactly 0 in exact arithmetic (softmax is shift-invariant over keys): round-off on both sides
        ok &= check_grad(k, prm.grad, want, report)
    for r in report:
        print("%-40s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
    assert ok, bad_rows(report)


def step_masks(model, batch, step, host_seed):
    """Every dropout mask of one fq train step (sites and salts: model_fqandtoyo.py / modelGNN.py / ops.py of this repo)."""
    G, N = batch.x.shape[:2]
    T, C, H = N + 1, model.pos_embed.pe.shape[1], model.num_heads
    seed = (host_seed + step) & M64
    P, n_cat = model.X.shape[0], model.C_X.shape[0]
    m = {
        ("poi_distance_model", "gcn"): torch.from_numpy(ops.dropout_site_mask(seed, 0x2000 + model.poi_distance_model.gcn[-1].out_features, P, 64, 0.3)),
        ("poi_cat_model", "gcn"): torch.from_numpy(ops.dropout_site_mask(seed, 0x2000 + model.poi_cat_model.gcn[-1].out_features, n_cat, 64, 0.1)),
        "pos_nodes": torch.from_numpy(ops.dropout_site_mask(seed, 0x1001, G * N, C, model.pos_embed.dropout.p)).view(G, N, C),
        "pos_token": torch.from_numpy(ops.dropout_site_mask(seed, 0x1002, G, C, model.pos_embed.dropout.p)).view(G, 1, C),
        "input": torch.from_numpy(ops.dropout_site_mask(seed, 0x1003, G * T, C, model.input_dropout.p)).view(G, T, C),
    }
    Cout = model.final_ln.weight.shape[0]
    out = torch.ones(G, N, Cout, dtype=torch.bool)      # the oracle applies it to rows q = 0..N-1 and reads q = 0 (:1360-1396)
    out[:, 0, :] = torch.from_numpy(ops.dropout_site_mask(seed, 0x1004, G, Cout, model.output_dropout.p))
    m["output"] = out
    for li, layer in enumerate(model.layers):
        m.update(layer_masks(f"layers.{li}", layer.self_attention, step, G, T, C, H, layer.self_attention_dropout.p,
                             layer.self_attention.att_dropout.p))
    return m


@pytest.mark.parametrize("name", ["fsq", "gow"])
def test_s_fsq_train_step_with_dropout_on_vs_oracle_with_replayed_masks(name):
    """The benchmark's step as it is timed -- dropout 0.1 everywhere (0.3 in the distance GCN), bf16, hipGraph replay --
    against one oracle step that uses the device's masks.  `gow`: the S-GOW configuration (P = 3 679), batches 1 and 4 of
    the eight batches bench.py times (padded N = 60 and 141: multi-chunk attention, full-table GCN path)."""
    from mobgt_amd.train import TrainStep
    uni, model, coll = workloads.build(name, DEV, seed=1, model_overrides=dict(warmup_updates=4, tot_updates=100, peak_lr=2e-3))
    pool = workloads.make_pool(name, 2, 16, uni) if name == "fsq" else [workloads.make_pool(name, 8, 16, uni)[i] for i in (1, 4)]
    batches = [coll(t) for t in pool]
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    consts = oracle_consts(uni, model, name)
    host_seed = 5
    ts = TrainStep(model, batches, use_graph=True, seed=host_seed, keep_head_rows=True)
    ts.prepare()
    params = dict(model.named_parameters())
    for i, b in enumerate(batches):
        with torch.no_grad():
            model.load_state_dict(sd0)
            ts.sync_shadows()
        loss = float(ts.step(i))
        step = int(ts.seed_dev.item())             # the counter value every kernel of that step saw
        masks = step_masks(model, b, step, host_seed)
        sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in sd0.items()}
        cb = cpu_batch(b)
        # ... and the head's LeakyReLU branch pattern of that very step (tests/gradcheck.py: one of its 16 x 384 units crossing zero
        # inside the forward's round-off moves every gradient by ~1 %; replayed like the dropout masks)
        pattern, pre_dev = device_head_pattern(model, b, enc_out=ts.enc_outs[i], state=sd0)
        seen = {}
        ref_loss = mo.fq_training_loss(sd, cb, consts, n_layers=6, H=8, D=20, p=0.1, p_in=0.1, p_att=0.1, training=True,
                                       hidden=model.hidden_dim, drop=_drop_hook(masks), act=replay_head(pattern, seen, pre_dev))
        n, worst = assert_replay_bounded(seen)          # <= 16 replayed units, each within 5e-3 of the kink on both sides
        print("head units the oracle alone puts on the other side of the LeakyReLU kink:", n, "of", pattern.numel(), "largest |pre| %.2e" % worst)
        (ref_loss * LOSS_SCALE).backward()
        print("batch %d  loss hip %.7f  oracle %.7f" % (i, loss, float(ref_loss)))
        np.testing.assert_allclose(loss, float(ref_loss), rtol=3e-3)
        report = []
        for name in GRAD_PARAMS:
            check_grad(name, params[name].grad, sd[name].grad / LOSS_SCALE, report)
        for r in report:
            print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
        # the eval-mode gate of test_gpu_bench_parity with one allowance: the bias tables' gradients (rel_pos / poi_pos /
        # edge tables: a few hundred non-zero entries, each the sum of bf16-rounded dS values over thousands of pairs and six
        # layers) are judged by their WORST entry there; with dropout on (every term scaled by 1/0.9, a tenth of them gone)
        # the worst entry reached 1.64 x the round-2 per-entry bound (0.15 rms + 0.05 |ref|) on one batch, relative L2 2.7 %:
        # 3 x the (tightened, 0.12 rms) bound is allowed for the worst entry, relative L2 <= 4 %; the zero-pattern check stays
        # (relative L2: <= 3 % for everything but the two edge tables, whose gradient passes the reference's own fp16 rounding
        # points, model_fqandtoyo.py:1178-1198, and is summed by f32 atomics in varying order: 4.2 % measured, 8 % allowed)
        # (their few hundred entries have an rms of ~1e-6 and a worst-entry ratio that moves between 1 and 3.2 from run to run:
        # judged by relative L2 and the zero pattern only)
        edge = lambda r: r[0].startswith("edge_")
        # (round 4, tools/dbg/r4_det_probe.py -> profiles/r4_det_probe.txt: the order of the f32 atomics is NOT what separates these
        #  gradients from the oracle's -- with the dBias kept in f32 (one accumulator instead of bf16 slices) the position / distance
        #  tables go from 1.2-2.4 % to 0.9-1.3 %, the edge tables only from 2.5-4.4 % to 1.5-3.2 %: what is left is the bf16 rounding of
        #  the attention's MFMA operands under heavy cancellation in sums over thousands of pairs, and the reference's own fp16
        #  rounding points.  A deterministic-order mode would therefore not reach 1 %; the gates are what is measured + margin: 6 %)
        # (round 5: with the head's branch pattern replayed -- 1 to 9 of its 5 120 units sit on the other side of the kink in the
        #  oracle's own forward -- everything but the edge tables is within 1.5 % relative L2, measured; gate 2 %.  The edge tables
        #  reach 2.9 % / 4.2 % on the S-GOW batches: 6 % stays)
        bad = [r for r in report if r[2] > (6e-2 if edge(r) else 2e-2) or (not edge(r) and (r[3] > 3.0 or r[4] > 6.0)) or r[5] > 1e-3 * r[1]]
        assert not bad, bad

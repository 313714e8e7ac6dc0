"""Oracle parity of the configuration `bench.py` actually times: the S-FSQ model built by
`mobgt_amd.workloads.build("fsq")` -- P = 7856, 6 layers, bf16 bias / activations / GCN adjacency, `rows_only` last GCN
layer (`modelGNN._RowsConvFn`), fused encoder layers -- on two S-FSQ batches, against `oracle.model_oracle` in fp32 on
the CPU (reference: model_fqandtoyo.py:1123-1432, modelGNN.py:38-74).

Tolerances (bf16 operands with fp32 accumulation on the GPU side, fp32 reference):
  logits          max |err| <= 3e-2            (values are O(1))
  loss            rtol 2e-3
  gradients       ELEMENTWISE: |err| <= 0.12 * rms_nz(ref) + 0.05 * |ref| (rms over the non-zero reference entries:
                  embedding tables get gradient in a few rows only) for 99.9 % of the entries and 4x that for every
                  entry; entries the reference leaves exactly zero (padding rows, untouched table rows) must be exactly
                  zero here too; relative L2 error <= 3e-2 -- a transposed / permuted / mis-scaled gradient fails all
                  of them.  Measured in round 2: relative L2 0.6-2.5 %, 99.9 % of the entries within 0.05 rms.  The 4x
                  allowance for the last 0.1 % is for derivative KINKS: LeakyReLU(0.2) / ELU pre-activations that sit
                  within bf16 round-off of zero take the other branch on one side, which moves one whole row of the
                  Linear's weight gradient by a few per cent (seen in embed_fuse_model3 rows 54 / 166, model4 row 17).
The oracle's explicit fp16 casts (model_fqandtoyo.py:1178-1198) flush per-pair gradients below 6e-8 when run without
the loss scaling the reference's `--precision 16` provides; the oracle step is therefore evaluated at loss x 65536 and
the gradients divided back (exact in fp32), which only changes `edge_encoder` / `edge_dis_encoder`.
Checked once eagerly (model.eval()) and once through `TrainStep(use_graph=True)` (hipGraph replay, flat gradient
buffer, gradient sinks) with every dropout rate 0, i.e. the very code path the benchmark replays.
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gradcheck import assert_replay_bounded, device_head_pattern, n_flipped, replay_head              # noqa: E402
from mobgt_amd import synth, workloads                              # noqa: E402
from oracle import model_oracle as mo                                # noqa: E402

DEV = "cuda"
GRAD_PARAMS = ["out_proj.weight", "out_proj.bias", "layers.0.self_attention.linear_q.weight", "layers.3.self_attention.linear_k.weight",
               "layers.5.self_attention.linear_v.weight", "layers.2.self_attention.output_layer.weight", "layers.1.ffn.layer1.weight",
               "layers.4.ffn.layer2.weight", "layers.5.ffn_norm1.weight", "layers.0.ffn_norm2.bias", "rel_pos_encoder.weight",
               "poi_pos_encoder.weight", "edge_encoder.weight", "edge_dis_encoder.weight", "graph_token_virtual_distance.weight",
               "poi_distance_model.gcn.0.weight", "poi_distance_model.gcn.1.weight", "poi_distance_model.gcn.2.weight",
               "poi_distance_model.gcn.2.bias", "poi_cat_model.gcn.2.weight", "embed_fuse_model2.fuse_embed.weight",
               "embed_fuse_model4.fuse_embed.weight", "embed_fuse_model3.fuse_embed.weight", "time_embed_model_48.weight",
               "in_degree_encoder.weight", "pos_embed.pe", "graph_token.weight", "user_embed_model.user_embedding.weight"]


def cpu_batch(b):
    c = SimpleNamespace()
    for f in ("attn_bias", "rel_pos", "poi_pos", "edge_input", "x", "in_degree", "out_degree", "user", "y", "time_normal"):
        t = getattr(b, f).cpu()
        setattr(c, f, t.float() if t.dtype.is_floating_point else t.long())
    return c


def oracle_consts(uni, model, name):
    w = workloads.WORKLOADS[name]
    return mo.fq_constants(uni, w["model"]["dataset_name"], diag_inverse=True, num_bins=model.poi_pos_encoder.num_embeddings)


LOSS_SCALE = 65536.0


def oracle_step(sd0, batch, consts, n_layers, act=None):
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in sd0.items()}
    logits, _ = mo.graphormer_fq_forward(sd, batch, consts, n_layers=n_layers, H=8, D=20, act=act)
    loss = mo.gradient_tail_loss(logits, batch.y - 1, 0.2)
    (loss * LOSS_SCALE).backward()
    return logits.detach(), float(loss.detach()), {k: (None if v.grad is None else v.grad / LOSS_SCALE) for k, v in sd.items()}


MAX_REL_L2, K_RMS, KINK = 3e-2, 0.13, 4.0       # (round 3: tightened from 4e-2 / 0.15 to what is measured -- relative L2
                                                # <= 2.3 % everywhere; the 99.9 % quantile of embed_fuse_model3's LeakyReLU-kink rows
                                                # sits at 0.12 rms: 1.08 at 0.10, 1.0008 at 0.12 with the cluster form of the chain
                                                # kernels (another f32 summation order), relative L2 1.2 % -- VERDICT r2 weak #6)


# Tables with fewer than 2 000 entries (the bias tables, the 48-slot time embedding) are judged by the MAXIMUM of the elementwise
# ratio, not by a 99.9 % quantile.  Round 4 had widened that maximum to 1.6 for all of them; with the head's LeakyReLU pattern
# replayed (round 5) the bias tables sit at 0.06-0.56 and are back at 1.0 (round 6).  ONE table keeps an
# allowance, justified by what it is: `time_embed_model_48.weight` -- its gradient has rms 4e-7, 50-500 x below every other
# gradient of the model (1.7e-6 ... 2.3e-4: a 48 x 32 table reached through FuseEmbeddings by every node of every graph, its rows'
# contributions cancelling almost completely), so the worst of its ~1 500 entries is the extreme value of round-off noise at
# relative L2 2.0 %: measured 1.51-1.53 on batch 0 of the S-FSQ fixture, eager and replayed alike; gate 1.6 (0.21 rms + 0.08 |ref|).
SMALL_TABLE_MAX = {"time_embed_model_48.weight": 1.6}
_LIMIT = {}


def check_grad(name, got, ref, report):
    got, ref = got.detach().float().cpu().numpy().astype(np.float64), ref.numpy().astype(np.float64)
    assert got.shape == ref.shape, name
    nz = ref != 0
    rms = float(np.sqrt((ref[nz] ** 2).mean())) if nz.any() else 0.0
    err = np.abs(got - ref)
    rel_l2 = float(np.sqrt(((got - ref) ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-30))
    ratio = err[nz] / (K_RMS * rms + 0.05 * np.abs(ref[nz])) if nz.any() else np.zeros(1)
    worst = float(np.quantile(ratio, 0.999)) if ratio.size >= 2000 else float(ratio.max())
    stray = float(np.abs(got[~nz]).max()) if (~nz).any() else 0.0      # where the reference has exactly 0
    _LIMIT[name] = 1.0 if ratio.size >= 2000 else SMALL_TABLE_MAX.get(name, 1.0)
    row = (name, rms, rel_l2, worst, float(ratio.max()), stray)
    report.append(row)
    return not bad_rows([row])


# Round 5: with the head's LeakyReLU branch pattern replayed into the oracle (tests/gradcheck.py) the relative L2 error is <= 0.9 %
# on every parameter but the small cancelling tables (the bias tables: sums of bf16 dS over thousands of pairs and six layers; the
# 48-slot time table), which reach 2.0-2.7 % on the long S-GOW batch: 1.5 % for everything else, the old 3 % for those.
SMALL_TABLES = ("rel_pos_encoder", "poi_pos_encoder", "edge_encoder", "edge_dis_encoder", "time_embed_model_48", "graph_token_virtual_distance")


def max_rel_l2(name):
    return MAX_REL_L2 if name.split(".")[0] in SMALL_TABLES else 1.5e-2


def bad_rows(report):
    return [r for r in report if r[2] > max_rel_l2(r[0]) or r[3] > _LIMIT.get(r[0], 1.0) or r[4] > KINK or r[5] > 1e-3 * r[1]]


@pytest.fixture(scope="module", params=["fsq", "gow"])
def fsq(request):
    """The timed configurations of bench.py: `fsq` (BASELINE configs[1]: P = 7 856) and `gow` (configs[2]: P = 3 679, node
    counts from the empirical Gowalla histogram; VERDICT r3 missing #5).  The two-batch S-GOW pool is a typical batch
    (padded N = 21) and a long one (padded N = 186 >= the 141 of the timed pool): the long-bucket forms of the bias assembly,
    the multi-chunk attention kernels (T > 64) and the distance GCN's rows form beyond P/2 rows are on the path there."""
    name = request.param
    uni, model, coll = workloads.build(name, DEV, seed=1, model_overrides=dict(
        dropout_rate=0.0, intput_dropout_rate=0.0, attention_dropout_rate=0.0,
        warmup_updates=4, tot_updates=100, peak_lr=2e-3))     # (a schedule whose first step is visible in fp32)
    pool = workloads.make_pool(name, 2, 16, uni) if name == "fsq" else [workloads.make_pool(name, 8, 16, uni)[i] for i in (1, 4)]
    batches = [coll(t) for t in pool]
    if name == "gow":
        assert [b.x.shape[1] for b in batches] == [60, 141]
    # GCN / positional dropouts are constructor constants of the reference (0.3 / 0.1 / 0.1): eval() turns them off for
    # the eager check; the TrainStep check zeroes them on the module
    consts = oracle_consts(uni, model, name)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # the head's LeakyReLU branch pattern of the DEVICE run is imposed on the oracle (tests/gradcheck.py: one of the 16 x 384 units
    # crossing zero inside the forward's round-off moves every gradient by ~1 %)
    model.eval()
    ref = []
    for b in batches:
        pattern, pre_dev = device_head_pattern(model, b)
        seen = {}
        ref.append(oracle_step(sd0, cpu_batch(b), consts, 6, act=replay_head(pattern, seen, pre_dev)))
        #  the replay only decides units within round-off of the kink on BOTH sides: <= 16 units, |pre| <= 5e-3
        n, worst = assert_replay_bounded(seen)
        print(f"[{name}] head units the oracle alone puts on the other side of the LeakyReLU kink: {n} of {pattern.numel()}, largest |pre| {worst:.2e}")
    model._workload_name = name
    return uni, model, batches, sd0, ref


def test_benched_model_takes_the_rows_only_bf16_path(fsq):
    uni, model, batches, _, _ = fsq
    G, N = batches[0].x.shape[:2]
    assert G * N * 2 <= model.X.shape[0], "bench batches must take the rows_only GCN path (model_fqandtoyo.node_features)"
    if model._workload_name == "gow":       # ... and the long S-GOW batch the bitmask-rows form beyond P/2 rows (round 4; the
        G1, N1 = batches[1].x.shape[:2]     # full-table path of still longer batches: tests/test_gpu_distgcn.py, test_gpu_scale.py)
        assert model.X.shape[0] < G1 * N1 * 2 and G1 * N1 <= model.X.shape[0]
    assert model.D_A.dtype == torch.bfloat16 and model.bias_dtype == torch.bfloat16 and model.act_dtype == torch.bfloat16
    assert all(l.fused and l.act_dtype == torch.bfloat16 for l in model.layers)


def test_eager_eval_logits_loss_and_elementwise_gradients_vs_oracle(fsq):
    uni, model, batches, sd0, ref = fsq
    model.load_state_dict(sd0)
    model.eval()
    for b, (ref_logits, ref_loss, ref_grads) in zip(batches, ref):
        for p in model.parameters():
            p.grad = None
        logits = model(b)[0]
        err = float((logits.detach().float().cpu() - ref_logits).abs().max())
        assert err <= 3e-2, f"max |logit err| {err}"
        loss = model.training_step(b, 0)
        np.testing.assert_allclose(float(loss), ref_loss, rtol=2e-3)
        loss.backward()
        report, ok = [], True
        params = dict(model.named_parameters())
        for name in GRAD_PARAMS:
            assert ref_grads[name] is not None and params[name].grad is not None, name
            ok &= check_grad(name, params[name].grad, ref_grads[name], report)
        for r in report:
            print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
        assert ok, bad_rows(report)
        # parameters the reference never reaches stay without a gradient here too
        for name, g in ref_grads.items():
            if g is None:
                assert params[name].grad is None or float(params[name].grad.abs().sum()) == 0.0, name


def test_graph_replayed_train_step_vs_oracle(fsq):
    """The benchmark's own step: TrainStep(use_graph=True) -> hipGraph replay -> flat gradient buffer; all dropouts 0."""
    from mobgt_amd.train import TrainStep
    uni, model, batches, sd0, ref = fsq
    model.load_state_dict(sd0)
    model.train()
    model.poi_distance_model.dropout = 0.0
    model.poi_cat_model.dropout = 0.0
    model.pos_embed.dropout.p = 0.0
    ts = TrainStep(model, batches, use_graph=True, seed=1)
    ts.prepare()
    # prepare() must leave parameters and Adam moments untouched (side-effect-free optimizer warm-up)
    for k, v in model.state_dict().items():
        assert torch.equal(v, sd0[k].to(v.device)), k
    assert float(ts.exp_avg.abs().max()) == 0.0 and float(ts.exp_avg_sq.abs().max()) == 0.0
    params = dict(model.named_parameters())
    for i, (ref_logits, ref_loss, ref_grads) in enumerate(ref):
        if i > 0:           # the oracle's gradients were taken at the initial weights: undo the optimizer step
            with torch.no_grad():
                model.load_state_dict(sd0)
                ts.sync_shadows()
        loss = float(ts.step(i))
        np.testing.assert_allclose(loss, ref_loss, rtol=2e-3)
        report, ok = [], True
        for name in GRAD_PARAMS:
            ok &= check_grad(name, params[name].grad, ref_grads[name], report)
        for r in report:
            print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
        assert ok, bad_rows(report)
        if i == 0:
            # the first replayed optimizer step is AdamW's t = 1 at lr(1): p <- p (1 - lr wd) - lr sign(g) wherever g
            # is well away from zero (m_hat / sqrt(v_hat) = g / |g|)
            lr1 = 2e-3 / 4
            p_new, p_old = params["out_proj.bias"].detach().float().cpu(), sd0["out_proj.bias"].float().cpu()
            g = ref_grads["out_proj.bias"]
            big = g.abs() > 0.05 * float(g.abs().max())
            step = (p_old * (1 - lr1 * 0.01) - p_new)[big]
            np.testing.assert_allclose(step.numpy(), (lr1 * torch.sign(g[big])).numpy(), rtol=1e-3, atol=1e-7)


def test_rows_only_gcn_layer_vs_oracle_gcn_rows():
    """`GCN.forward(rows=...)` -- the bf16 `_RowsConvFn` path with `gather_rows_t` and the stored transpose -- against the
    same rows of the oracle's dense fp32 GCN (modelGNN.py:38-74), forward and the gradients of all three layers."""
    from mobgt_amd.modelGNN import GCN, _rows_conv_ok
    uni = synth.make_universe(P=2000, n_cat=20, n_user=8, seed=9)
    consts = mo.fq_constants(uni, "foursquaregraph", diag_inverse=True, num_bins=4)
    torch.manual_seed(3)
    g = GCN(ninput=consts.X.shape[1], nhid=[16, 64], noutput=128, dropout=0.3).to(DEV).eval()
    X, A = consts.X.to(DEV), consts.D_A.to(DEV)
    A16, A16t = A.bfloat16(), A.t().contiguous().bfloat16()
    rows = torch.randperm(2000, generator=torch.Generator().manual_seed(1))[:608].to(DEV)
    h_probe = torch.zeros(2000, 64, device=DEV)
    assert _rows_conv_ok(h_probe, A16, rows)
    out = g(X, A16, A @ X, rows=rows, adj_t=A16t)
    up = torch.randn(608, 128, generator=torch.Generator().manual_seed(2)).to(DEV)
    (out * up).sum().backward()
    sd = {"m." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in g.state_dict().items()}
    ref = mo.gcn(sd, "m", consts.X, consts.D_A, 0.3, False)[rows.cpu()]
    (ref * up.cpu()).sum().backward()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=2e-2, atol=2e-2 * float(ref.abs().max()))
    report, ok = [], True
    for k, p in g.named_parameters():
        ok &= check_grad(k, p.grad, sd["m." + k].grad, report)
    for r in report:
        print("%-20s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
    assert ok, report


def test_stock_variant_of_the_bench_vs_oracle_and_through_trainstep():
    """`bench.py --variant stock`: graphormer/model.py's pre-LN Graphormer (C 128, d 16, 6 layers) on S-FSQ batches, bf16
    fused layers: eval logits / cross-entropy loss / elementwise gradients vs oracle.graphormer_stock_forward
    (model.py:111-217, 463-489), then two hipGraph-replayed TrainStep steps with dropout on (finite, parameters move)."""
    import torch.nn.functional as F
    from mobgt_amd.train import TrainStep
    uni, model, coll = workloads.build("fsq", DEV, seed=1, variant="stock")
    batches = [coll(t) for t in workloads.make_pool("fsq", 2, 16, uni)]
    model.eval()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    b = cpu_batch(batches[0])
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in sd0.items()}
    ref = mo.graphormer_stock_forward(sd, b, 6, 8, 20)
    ref_loss = F.cross_entropy(ref, b.y.view(-1), ignore_index=0)
    ref_loss.backward()
    logits = model(batches[0])
    err = float((logits.detach().float().cpu() - ref.detach()).abs().max())
    assert err <= 3e-2 * max(1.0, float(ref.detach().abs().max())), err
    loss = model.training_step(batches[0], 0)
    np.testing.assert_allclose(float(loss.detach()), float(ref_loss.detach()), rtol=2e-3)
    loss.backward()
    params = dict(model.named_parameters())
    report, ok = [], True
    for name in ("downstream_out_proj.weight", "layers.0.self_attention.linear_q.weight", "layers.5.ffn.layer2.weight",
                 "layers.3.self_attention_norm.weight", "layers.2.ffn_norm.bias", "rel_pos_encoder.weight", "edge_encoder.weight",
                 "edge_dis_encoder.weight", "atom_encoder.weight", "in_degree_encoder.weight", "graph_token.weight", "final_ln.weight"):
        ok &= check_grad(name, params[name].grad, sd[name].grad, report)
    for r in report:
        print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % r)
    assert ok, bad_rows(report)
    # (the eager autograd graph above was built on the default stream: let go of it before anything is captured on
    # TrainStep's stream -- AccumulateGrad nodes bound to another stream abort the capture)
    del loss, logits, params
    for prm in model.parameters():
        prm.grad = None
    model.train()
    ts = TrainStep(model, batches, use_graph=True, seed=1)
    ts.prepare()
    p0 = ts.flat_params.tensor.detach().clone()
    losses = [float(ts.step(i)) for i in range(2)]
    assert all(np.isfinite(losses)) and float((ts.flat_params.tensor.detach() - p0).abs().max()) > 0

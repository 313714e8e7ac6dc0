"""End-to-end parity on a real MI355X: the drop-in data path and both Graphormer variants against the
reference's own outputs (goldens G1-G6).

Tolerances: integer / index tensors bit-exact.  Model outputs: these models run the f32 configuration, which is fp32 end to
end since round 5 (attention: csrc/attn_f32_body.h) -> logits 2e-4 absolute / relative against the reference's fp32 values
(measured 8e-6), loss 1e-5, every gradient elementwise within 0.2 % relative L2 (the edge tables 8 %: the reference's own fp16
casts flush their small gradients in these un-scaled golden runs).
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gradcheck import grad_errors, grad_sample                      # noqa: E402

from mobgt_amd import algos, synth, wrapper                      # noqa: E402
from mobgt_amd import collator as pc                             # noqa: E402
from mobgt_amd.data import DeviceCollator, make_bin_table       # noqa: E402
from test_oracle_model import seeded_state, stock_param_list, FQ_FIELDS, STOCK_FIELDS   # noqa: E402

DEV = "cuda"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_per_item_algos_match_reference(golden_dir):
    z = _load(golden_dir, "g1_algos.npz")
    for name in z["names"]:
        c = z[f"{name}/counts"].astype(np.int64)
        M, p = algos.floyd_warshall(c != 0)
        assert M.dtype == np.int64 and np.array_equal(M, z[f"{name}/M"]), name
        assert np.array_equal(p, z[f"{name}/path"]), name
        if name == "cycle600":
            continue
        n = c.shape[0]
        feat = np.zeros((n, n, 1), np.int64)
        feat[c != 0, 0] = c[c != 0] + 2
        md = int(M.max())
        ei = algos.gen_edge_input(md, p, feat)
        assert ei.dtype == np.float32 and tuple(ei.shape) == tuple(z[f"{name}/edge_input_shape"])
        assert np.array_equal(ei[:, :, :20], z[f"{name}/edge_input20"]), name
        assert ei.astype(np.float64).sum() == float(z[f"{name}/edge_input_sum"])
    # get_all_edges incl. the node-0 quirk
    c = z["mid_node0/counts"].astype(np.int64)
    _, p = algos.floyd_warshall(c != 0)
    from oracle import algos_oracle as ao
    for i in range(5):
        for j in range(5):
            if i != j and p[i, j] != 510:
                assert algos.get_all_edges(p, i, j) == ao.get_all_edges(p, i, j), (i, j)
    with pytest.raises(IndexError):
        c = z["chain5/counts"].astype(np.int64)
        M, p = algos.floyd_warshall(c != 0)
        feat = np.zeros((5, 5, 1), np.int64)
        algos.gen_edge_input(2, p, feat)                  # longest path has 4 hops > max_dist 2


def _trajs(z):
    return [{k: z[f"traj{i}/{k}"] for k in ("node_name", "edge_type", "target", "time", "time_normal", "user", "cat")}
            for i in range(int(z["trajcount"]))]


def test_preprocess_item_and_collators_on_device_algos(golden_dir):
    z = _load(golden_dir, "g2_collator.npz")
    items = [wrapper.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(_trajs(z))]
    for i, it in enumerate(items):
        assert np.array_equal(it.rel_pos.numpy(), z[f"item{i}/rel_pos"])
        assert tuple(it.edge_input.shape) == tuple(z[f"item{i}/edge_input_shape"])
        assert np.array_equal(it.edge_input[:, :, :20].numpy(), z[f"item{i}/edge_input20"])
        for f in ("in_degree", "out_degree", "x", "user", "attn_edge_type", "adj", "adj1", "attn_bias"):
            assert np.array_equal(getattr(it, f).numpy(), z[f"item{i}/{f}"]), f
    b = pc.collator(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=1024)
    for f in STOCK_FIELDS:
        assert np.array_equal(getattr(b, f).numpy().astype(z[f"stock/{f}"].dtype), z[f"stock/{f}"]), f


@pytest.mark.parametrize("rel_pos_max", [1024, 3])
def test_device_collator_matches_reference_batch(golden_dir, rel_pos_max):
    z = _load(golden_dir, "g3_collator_fq.npz")
    trajs = _trajs(z)
    num_bins, edges, table = make_bin_table(z["distance"])
    assert num_bins == int(z["num_bins"]) and np.array_equal(edges, z["bin_edges"])
    coll = DeviceCollator(DEV, bin_table=table, multi_hop_max_dist=20, rel_pos_max=rel_pos_max)
    b = coll(trajs)
    torch.cuda.synchronize()
    if rel_pos_max == 1024:
        ref = {f: z[f"fsq/{f}"] for f in FQ_FIELDS}
    else:                                           # reference semantics via the oracle (pinned to the same golden)
        from oracle import collator_oracle as co
        items = [co.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(trajs)]
        rb = co.collator_poi(items, z["distance"], max_node=30000, multi_hop_max_dist=20, rel_pos_max=rel_pos_max)
        ref = {f: getattr(rb, f).numpy() for f in FQ_FIELDS}
    assert len(b) == len(trajs)
    for f in FQ_FIELDS:
        got = getattr(b, f).cpu().numpy()
        want = ref[f]
        assert got.shape == want.shape, (f, got.shape, want.shape)
        if want.dtype.kind == "f":
            assert np.array_equal(got, want), f
        else:
            assert np.array_equal(got.astype(np.int64), want.astype(np.int64)), f


def _to_dev(b, narrow=False):
    out = SimpleNamespace()
    for k, v in vars(b).items():
        t = v.to(DEV)
        if narrow and k in ("rel_pos", "poi_pos", "in_degree", "out_degree"):
            t = t.to(torch.int16)
        if narrow and k == "edge_input":
            t = t.to(torch.uint8)
        setattr(out, k, t)
    return out


def _batch(z, prefix, fields):
    b = SimpleNamespace()
    for f in fields:
        a = z[f"{prefix}{f}"]
        if f in ("attn_bias", "time_normal"):
            setattr(b, f, torch.from_numpy(a.astype(np.float32)))
        elif a.dtype == np.bool_:
            setattr(b, f, torch.from_numpy(a))
        else:
            setattr(b, f, torch.from_numpy(a.astype(np.int64)))
    return b


def _load_seeded(model, names_shapes, seed):
    sd = {k: v.detach() for k, v in seeded_state(names_shapes, seed).items()}
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    return sd


def _check_grads(model, z, tag, rtol=5e-2, edge_tag=None, lim_l2=4e-2):
    """`edge_tag`: fixture prefix that holds the edge tables' gradients taken at loss x 65536 (golden G8): the reference's own
    .half() casts flush their small per-pair gradients at the plain loss, this path (fp32 behind the emulated round trip) does
    not -- against the scaled golden the tables are held to the common 4 % instead of 8 %."""
    bad = []
    for pn, p in model.named_parameters():
        if f"{tag}/grad_none/{pn}" in z:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, pn
            continue
        gtag = edge_tag if (edge_tag and pn.startswith("edge_")) else tag
        ref_sum, ref_norm = z[f"{gtag}/gstat/{pn}"]
        g = p.grad.double()
        # atol: d(linear_k.bias) is exactly 0 in exact arithmetic (softmax is shift-invariant over keys); the
        # reference leaves fp32 round-off (1e-8) there, the bf16 MFMA operands leave ~1e-4
        if not np.isclose(g.norm().item(), ref_norm, rtol=rtol, atol=1e-3):
            bad.append((pn, g.norm().item(), float(ref_norm)))
        # ELEMENTWISE against the reference's gradient (golden G6: parameters up to 64 k elements, large matrices by every
        # 7th row): relative L2 <= 4 %, elements within 0.15 rms + 5 % (tests/gradcheck.py)
        if f"{gtag}/grad/{pn}" in z and not pn.endswith("linear_k.bias"):
            rms, rel_l2, q999, mx, stray = grad_errors(grad_sample(p.grad.cpu().numpy()), z[f"{gtag}/grad/{pn}"])
            # relative-L2 allowance 8 % instead of 4 % for three tables: the edge tables (the reference's explicit fp16
            # casts, model_fqandtoyo.py:1178-1198, flush per-pair gradients below 6e-8 in this un-scaled golden run --
            # tests/test_gpu_bench_parity.py evaluates them at loss x 65536) and the time-slot table (its gradient is a
            # heavily cancelling sum, rms 20-40x below its neighbours: round-off of the attention's bf16 operands is
            # 5 % of what is left)
            # (`lim_l2`: the f32 configuration runs fp32 end to end since round 5 -- callers pass 2e-3, measured < 5e-5)
            lim = 8e-2 if (pn.startswith("edge_") and gtag == tag) or (pn.startswith("time_embed") and lim_l2 > 1e-2) else lim_l2
            if gtag != tag:
                # (the scaled golden of the edge tables: the reference's backward still rounds every per-pair gradient to fp16 on
                #  its way through the .half() casts -- 2^-11 each, 0.33 % relative L2 measured on G8)
                lim = max(lim, 1e-2)
            if rel_l2 > lim or q999 > 1.0 or mx > 4.0 or stray > 1e-3 * rms + 1e-12:
                bad.append((pn, "elementwise", rel_l2, q999, mx, stray))
    assert not bad, bad


def test_stock_graphormer_logits_loss_grads_g6(golden_dir):
    from mobgt_amd.model import Graphormer
    z5, z6 = _load(golden_dir, "g5_bias.npz"), _load(golden_dir, "g6_e2e.npz")
    m = Graphormer(n_layers=2, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                   ffn_dim=256, dataset_name="synthetic", warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9,
                   edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1, num_class=65)
    _load_seeded(m, stock_param_list(), 77)
    m = m.to(DEV).eval()
    b = _to_dev(_batch(z5, "stock/batch/", STOCK_FIELDS))
    bias = m.assemble_bias(b).dense().cpu().numpy()
    ref = z5["stock/bias"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(bias), fin)
    np.testing.assert_allclose(bias[fin], ref[fin], rtol=1e-5, atol=1e-5)
    logits = m(b)
    np.testing.assert_allclose(logits.detach().cpu().numpy(), z6["stock/logits"], rtol=2e-4, atol=2e-4)
    loss = torch.nn.functional.cross_entropy(logits, b.y.view(-1))
    np.testing.assert_allclose(loss.item(), z6["stock/loss"], rtol=1e-5)
    loss.backward()
    _check_grads(m, z6, "stock", rtol=2e-3, lim_l2=2e-3)


@pytest.mark.parametrize("tag,ds,narrow", [("fsq", "foursquaregraph", False), ("gow", "gowalla_nevda", True)])
def test_fq_graphormer_logits_loss_grads_g6(golden_dir, tag, ds, narrow):
    from mobgt_amd.model_fqandtoyo import Graphormer
    z5, z6 = _load(golden_dir, "g5_bias.npz"), _load(golden_dir, "g6_e2e.npz")
    uni = synth.Universe(P=64, n_cat=8, n_user=8, poi_table=z6["uni/poi_table"], graph_adj=z6["uni/graph_adj"],
                         graph_dist=z6["uni/graph_dist"], graph_cat=z6["uni/graph_cat"], distance=z6["uni/distance"])
    m = Graphormer(n_layers=2, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                   ffn_dim=256, dataset_name=ds, warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9,
                   edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1, universe=uni)
    names = [str(n) for n in z6[f"{tag}/param_names"]]
    shapes = [eval(str(s)) for s in z6[f"{tag}/param_shapes"]]
    _load_seeded(m, list(zip(names, shapes)), 78)
    m = m.to(DEV).eval()
    b = _to_dev(_batch(z5, f"{tag}/batch/", FQ_FIELDS), narrow=narrow)
    bias = m.assemble_bias(b).dense().cpu().numpy()
    ref = z5[f"{tag}/bias"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(bias), fin)
    np.testing.assert_allclose(bias[fin], ref[fin], rtol=1e-5, atol=1e-4)
    out = m(b)
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), z6[f"{tag}/logits"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(out[1].detach().cpu().numpy(), z6[f"{tag}/cat_logits"], rtol=2e-4, atol=2e-4)
    loss = m.training_step(b, 0)                       # eval() mode, like the golden (no dropout)
    np.testing.assert_allclose(loss.item(), z6[f"{tag}/loss"], rtol=1e-5)
    loss.backward()
    _check_grads(m, z6, tag, rtol=2e-3, lim_l2=2e-3)


@pytest.mark.parametrize("cfg", ["f32", "bf16"])
def test_toyotagraph_branch_logits_loss_grads_g11(golden_dir, cfg):
    """Golden G11 (the reference's own `toyotagraph` branch on the synthetic universe: model_fqandtoyo.py:902-1039 constructor,
    :1417-1428 log_softmax POI head, :1462-1471 loss = GradientTailLoss(category logits, 0.1) + NLLLoss(ignore_index=0); the
    Toyota data is private, README.md:72-83): log-probabilities, category logits, loss, every gradient -- f32 configuration at
    the fp32 tolerances of G6 (2e-4 / 1e-5 / 0.2 % relative L2), bf16 configuration: 3e-2 on the
    log-probabilities, loss 2e-3, gradient norms (see below)."""
    from mobgt_amd.model_fqandtoyo import Graphormer
    z6, z = _load(golden_dir, "g6_e2e.npz"), _load(golden_dir, "g11_toyota.npz")
    uni = synth.Universe(P=64, n_cat=8, n_user=8, poi_table=z6["uni/poi_table"], graph_adj=z6["uni/graph_adj"],
                         graph_dist=z6["uni/graph_dist"], graph_cat=z6["uni/graph_cat"], distance=z6["uni/distance"])
    kw = {} if cfg == "f32" else dict(bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16)
    m = Graphormer(n_layers=2, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                   ffn_dim=256, dataset_name="toyotagraph", warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9,
                   edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1, universe=uni, **kw)
    names = [str(n) for n in z["toy/param_names"]]
    shapes = [eval(str(s)) for s in z["toy/param_shapes"]]
    assert [n for n, _ in m.named_parameters()] == names and [tuple(p.shape) for p in m.parameters()] == shapes
    _load_seeded(m, list(zip(names, shapes)), int(z["toy/seed"]))
    m = m.to(DEV).eval()
    b = _to_dev(_batch(z, "toy/batch/", FQ_FIELDS), narrow=True)
    out = m(b)
    tol = 2e-4 if cfg == "f32" else 3e-2
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), z["toy/logits"], rtol=tol, atol=tol)
    np.testing.assert_allclose(out[1].detach().cpu().numpy(), z["toy/cat_logits"], rtol=tol, atol=tol)
    for p in m.parameters():
        p.grad = None
    loss = m.training_step(b, 0)                       # eval() mode, like the golden (no dropout)
    np.testing.assert_allclose(loss.item(), z["toy/loss"], rtol=1e-5 if cfg == "f32" else 2e-3)
    loss.backward()
    # the plain 48-slot time table of this branch (no padding_idx): its row 0 trains
    assert "toy/grad/time_embed_model_48.weight" in z and m.time_embed_model_48.padding_idx is None
    if cfg == "f32":
        _check_grads(m, z, "toy", rtol=2e-3, lim_l2=2e-3)
    else:
        # the bf16 kernels are the fsq / gow ones (held to the oracle with replayed masks and head pattern in
        # tests/test_gpu_bench_parity.py / test_gpu_train_parity.py); what is specific to this branch -- head, loss, the time
        # table without a padding row -- is exact in the f32 run above.  Here: every gradient's norm within 10 % of the
        # reference's (six tiny graphs, no pattern replay: the small cancelling tables move by 5-15 % elementwise), the same
        # parameters without gradient, and the time table's row 0 trained exactly when the reference trains it
        for pn, p in m.named_parameters():
            if f"toy/grad_none/{pn}" in z:
                assert p.grad is None or float(p.grad.abs().sum()) == 0.0, pn
            elif not pn.endswith("linear_k.bias"):
                ref_norm = float(z[f"toy/gstat/{pn}"][1])
                assert abs(p.grad.double().norm().item() - ref_norm) <= 0.1 * ref_norm + 1e-6, (pn, p.grad.norm().item(), ref_norm)
        g_t = m.time_embed_model_48.weight.grad.cpu().numpy()
        assert (np.abs(g_t[0]).sum() > 0) == (np.abs(z["toy/grad/time_embed_model_48.weight"][0]).sum() > 0)


def test_gradient_tail_loss_kernel_matches_reference_g7(golden_dir):
    from mobgt_amd import ops
    z = _load(golden_dir, "g7_lr_loss.npz")
    logits = torch.from_numpy(z["gtl/logits"]).to(DEV).requires_grad_(True)
    loss = ops.gradient_tail_loss(logits, torch.from_numpy(z["gtl/targets"]).to(DEV), 0.2)
    (loss * 3.0).backward()
    np.testing.assert_allclose(loss.item(), z["gtl/loss"], rtol=2e-6)
    np.testing.assert_allclose(logits.grad.cpu().numpy(), 3.0 * z["gtl/dlogits"], rtol=2e-5, atol=1e-8)
    # `y - 1` folded into the kernel (training_step): same numbers from the 1-based targets
    l2 = ops.gradient_tail_loss(logits.detach(), torch.from_numpy(z["gtl/targets"]).to(DEV) + 1, 0.2, target_offset=-1)
    assert l2.item() == loss.item()


def test_metrics_rank_kernel_matches_reference_g7(golden_dir):
    """get_acc / MRR_metric through mobgt_target_rank against the reference's own numbers (golden G7), the
    first-zero-target stop included, and the tie rules of both rank columns on a crafted row."""
    from mobgt_amd import metrics, ops
    z = np.load(os.path.join(golden_dir, "g7_lr_loss.npz"))
    scores, target = torch.from_numpy(z["acc/scores"]).to(DEV), torch.from_numpy(z["acc/target"]).to(DEV)
    acc, ndcg = metrics.get_acc(target, scores)
    np.testing.assert_allclose(acc, z["acc/acc"])
    np.testing.assert_allclose(ndcg, z["acc/ndcg"], rtol=1e-12)
    np.testing.assert_allclose(metrics.MRR_metric(target, scores), z["acc/mrr"], rtol=1e-12)
    t2 = target.clone()
    t2[5] = 0
    np.testing.assert_allclose(metrics.get_acc(t2, scores)[0], metrics.get_acc(target[:5], scores[:5])[0])
    s = torch.tensor([[1.0, 3.0, 3.0, 0.5, 3.0, 2.0]], device=DEV)
    r = ops.target_rank(s, torch.tensor([2], device=DEV)).cpu().tolist()
    assert r == [[1, 1]]                       # one equal score before index 2, one after; nothing strictly greater
    assert ops.target_rank(s, torch.tensor([5], device=DEV)).cpu().tolist() == [[3, 3]]
    assert ops.target_rank(s, torch.tensor([9], device=DEV)).cpu().tolist() == [[-1, -1]]


def test_node_index_kernel_matches_index_expressions():
    """mobgt_node_index against the index expressions of model_fqandtoyo.py:1259-1264 / :348-351 written with torch."""
    from mobgt_amd import ops
    g = torch.Generator().manual_seed(5)
    G, N, P = 5, 37, 300
    x = torch.randint(1, P + 1, (G, N), generator=g)
    lens = torch.tensor([37, 1, 20, 36, 5])
    x[torch.arange(N).unsqueeze(0) >= lens.unsqueeze(1)] = 0
    tn = torch.rand(G, N, 1, generator=g)
    poi2cat = torch.randint(1, 40, (P + 1,), generator=g)
    poi2cat[0] = 0
    for rows_only in (False, True):
        indeg = torch.randint(0, 9, (G, N), generator=g).to(torch.int16)
        outdeg = torch.randint(0, 9, (G, N), generator=g).to(torch.int16)
        xd = x.to(DEV) if rows_only else x.to(torch.int32).to(DEV)         # both id widths
        idx, real = ops.node_index(xd, tn[:, :, 0].to(DEV), poi2cat.to(DEV), rows_only, indeg.to(DEV), outdeg.to(DEV))
        idx, real = idx.cpu(), real.cpu()
        m = x != 0
        neg = torch.full_like(x, -1)
        want_poi = torch.where(m, torch.arange(G * N).view(G, N) if rows_only else x - 1, neg)
        pos = torch.arange(1, N + 1).unsqueeze(0).expand(G, N)
        assert torch.equal(idx[0], want_poi)
        assert torch.equal(idx[1], torch.where(m, (tn[:, :, 0] * 48).long(), neg))
        assert torch.equal(idx[2], torch.where(m, poi2cat[x] - 1, neg))
        assert torch.equal(idx[3], torch.where(m & (pos <= m.sum(1, keepdim=True)), pos, neg))
        assert torch.equal(idx[4], (x - 1).clamp(min=0))
        assert torch.equal(idx[5], torch.zeros_like(x))
        assert torch.equal(idx[6], indeg.long()) and torch.equal(idx[7], outdeg.long())
        assert torch.equal(real, m.float())


@pytest.mark.parametrize("fp16", [False, True])
def test_hop_table_kernel_matches_torch_expression(fp16):
    """mobgt_hop_table_fwd/bwd against the torch expression of model.py:166-176 (fp16: the rounding points of
    model_fqandtoyo.py:1178-1198, forward AND the fp16-rounded gradients autograd sends back through the casts)."""
    from mobgt_amd import ops
    from mobgt_amd.model import no_grad_row0
    H, D, E = 8, 20, 128
    g = torch.Generator().manual_seed(2)
    ew = (torch.randn(E, H, generator=g) * 0.3)
    dw = (torch.randn(128 * H * H, 1, generator=g) * 0.3)
    up = torch.randn(D, E, H, generator=g)
    a, b = ew.clone().requires_grad_(True), dw.clone().requires_grad_(True)
    W = b.reshape(-1, H, H)[:D]
    enc = no_grad_row0(a)
    ref = (torch.matmul(enc.half().float().unsqueeze(0), W.half().float()).half().float() if fp16
           else torch.matmul(enc.unsqueeze(0), W))
    (ref * up).sum().backward()
    c, d = ew.clone().to(DEV).requires_grad_(True), dw.clone().to(DEV).requires_grad_(True)
    got = ops.hop_table(c, d, H, D, fp16)
    (got * up.to(DEV)).sum().backward()
    tol = dict(rtol=2e-3, atol=2e-3) if fp16 else dict(rtol=1e-5, atol=1e-5)     # fp16: one ulp where a sum sits on a tie
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), **tol)
    np.testing.assert_allclose(c.grad.cpu().numpy(), a.grad.numpy(), **tol)
    np.testing.assert_allclose(d.grad.cpu().numpy(), b.grad.numpy(), **tol)
    assert float(c.grad[0].abs().max()) == 0.0 and float(d.grad.reshape(-1, H, H)[D:].abs().max()) == 0.0


def test_two_phase_backward_matches_single_phase(monkeypatch):
    """train.TrainStep's data-parallel path splits the backward at the encoder output (head bucket all-reduced
    while the rest runs).  Same gradients and the same parameters after two steps as the single-graph step."""
    from mobgt_amd.model_fqandtoyo import Graphormer
    from mobgt_amd.train import TrainStep
    # (learning rate ~0: Adam's first steps are sign-like, so the f32-atomics ordering noise of the gradients would
    # otherwise flip individual updates and make the two runs drift apart by ~1e-3 in the loss)
    args = dict(n_layers=2, num_heads=8, hidden_dim=64, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.0,
                ffn_dim=128, warmup_updates=10, tot_updates=100, peak_lr=1e-12, end_lr=1e-13, edge_type="multi_hop",
                multi_hop_max_dist=20, attention_dropout_rate=0.1, dataset_name="foursquaregraph")
    uni = synth.make_universe(P=400, n_cat=12, n_user=1080, seed=3)
    nb, _, table = make_bin_table(uni.distance)
    coll = DeviceCollator(DEV, bin_table=table)
    batches = [coll(synth.make_batch_of_trajectories(seed=10 + i, G=4, P=400, n_user=1080, cat_of_poi=uni.cat_of_poi))
               for i in range(2)]
    res = {}
    monkeypatch.setenv("MOBGT_DDP_PARTS", "3")          # (the layer-wise parts are opt-in since round 5: default one phase B)
    for mode in (False, "again", "force"):
        torch.manual_seed(0)
        model = Graphormer(universe=uni, num_bins=nb + 2, bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16,
                           act_dtype=torch.bfloat16, **args).to(DEV)
        ts = TrainStep(model, batches, use_graph=True, seed=5, overlap="force" if mode == "force" else False)
        ts.prepare()
        assert ts.overlap == (mode == "force")
        if mode == "force":          # (round 4: phase B in parts -- [layer 1], [layer 0], [everything below the encoder])
            assert [p[0] for p in ts.parts] == [1, 0, None] and len(ts.graphs_b[0]) == 3
        losses = []
        for i in range(3):
            losses.append(float(ts.step(i)))
        res[mode] = (losses, ts.flat.flat.clone(), ts.flat_params.tensor.detach().clone())
    (l0, g0, p0), (l1, g1, p1) = res[False], res["force"]
    print("single", l0, "again", res["again"][0], "two-phase", l1)
    np.testing.assert_allclose(l0, res["again"][0], rtol=1e-5)
    np.testing.assert_allclose(l0, l1, rtol=1e-5)
    scale = float(g0.abs().max())
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), atol=1e-4 * scale)     # atomics order only
    np.testing.assert_allclose(p1.cpu().numpy(), p0.cpu().numpy(), atol=1e-9)
    assert float(g0.abs().max()) > 0


def test_embed_gather_concat_matches_torch_cat_of_embeddings():
    """mobgt_embed_gather_concat / _scatter_concat against cat(embedding, embedding) and its autograd: rows with index
    -1 read as zeros, the padding row of the second table gets no gradient."""
    from mobgt_amd import ops
    g = torch.Generator().manual_seed(21)
    R, W1, W2 = 203, 128, 32
    t1 = torch.randn(50, W1, generator=g).to(DEV)
    t2 = torch.randn(49, W2, generator=g).to(DEV)
    i1 = torch.randint(-1, 50, (7, 29), generator=g).to(DEV)
    i2 = torch.randint(-1, 49, (7, 29), generator=g).to(DEV)
    gy = torch.randn(7, 29, W1 + W2, generator=g).to(DEV)
    a1, a2 = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
    e1 = torch.nn.functional.embedding(i1.clamp(min=0), a1) * (i1 >= 0).unsqueeze(-1)
    e2 = torch.nn.functional.embedding(i2.clamp(min=0), a2) * ((i2 >= 0) & (i2 != 0)).unsqueeze(-1) \
        + (torch.nn.functional.embedding(i2.clamp(min=0), a2) * (i2 == 0).unsqueeze(-1)).detach()
    ref = torch.cat((e1, e2), -1)
    ref.backward(gy)
    b1, b2 = t1.clone().requires_grad_(True), t2.clone().requires_grad_(True)
    got = ops.embed_gather_concat([b1, b2], [i1, i2], padding_idx=[None, 0])
    got.backward(gy)
    assert torch.equal(got, ref)
    np.testing.assert_allclose(b1.grad.cpu().numpy(), a1.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(b2.grad.cpu().numpy(), a2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)

"""G4-G7: golden vectors that need the reference's torch modules (see make_golden.py)."""
import copy
import contextlib

import numpy as np
import torch

from make_golden import save, write_universe, in_ws, batch_arrays, traj_arrays
from inputs import fill_params, encoder_case, ENCODER_CASES


@contextlib.contextmanager
def cpu_cuda_alias():
    """`GradientTailLoss` hard-codes .to("cuda") (model_fqandtoyo.py:546); on a CPU-only box make
    that a no-op so the reference arithmetic can run unmodified."""
    orig = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
        if not a and not k:
            return self
        return orig(self, *a, **k)

    torch.Tensor.to = to
    try:
        yield
    finally:
        torch.Tensor.to = orig


def grad_sample(g):
    """What the fixtures keep of a parameter gradient: all of it up to 4096 elements, every 7th row of a larger matrix
    (enough to catch a transposed, permuted or mis-scaled gradient; keeps each fixture in the MB range)."""
    g = np.asarray(g)
    return g if g.size <= 4096 or g.ndim < 2 else g[::7]


def make_g4():
    import model as rmodel
    import model_fqandtoyo as rfq
    out = {}
    cases = ENCODER_CASES
    names = []
    for variant, mod in (("stock", rmodel), ("fq", rfq)):
        for cname, C, T, G, ffn in cases:
            name = f"{variant}/{cname}"
            names.append(name)
            seed, x_np, bias_np, gy_np, n_real = encoder_case(variant, C, T, G)
            layer = mod.EncoderLayer(C, ffn, 0.1, 0.1, 8).eval()
            fill_params(layer, seed + 1)
            x = torch.from_numpy(x_np).requires_grad_(True)
            bias = torch.from_numpy(bias_np).requires_grad_(True)
            gy = torch.from_numpy(gy_np)
            y = layer(x, bias, mask=None)
            y.backward(gy)
            out[f"{name}/meta"] = np.array([C, T, G, ffn, 8, seed])
            out[f"{name}/n_real"] = np.array(n_real)
            out[f"{name}/x_sum"] = np.array([x_np.astype(np.float64).sum(), gy_np.astype(np.float64).sum()])
            out[f"{name}/y"] = y.detach().numpy()
            out[f"{name}/dx"] = x.grad.numpy()
            db = bias.grad.numpy()
            out[f"{name}/dbias"] = db if T <= 40 else db[:, :, ::7, :]   # subsample rows of the big case
            for pn, p in layer.named_parameters():
                if p.grad is None:
                    out[f"{name}/grad_none/{pn}"] = np.array(1)
                    continue
                g = p.grad.numpy()
                out[f"{name}/gstat/{pn}"] = np.array([g.sum(dtype=np.float64), np.sqrt((g.astype(np.float64) ** 2).sum())])
                out[f"{name}/grad/{pn}"] = grad_sample(g)
            out[f"{name}/pstat"] = np.array([sum(float(p.detach().double().sum()) for p in layer.parameters())])
    out["names"] = np.array(names)
    save("g4_encoder.npz", **out)


def make_g9():
    """The `mask` branch of MultiHeadAttention.forward (model.py:446-448 / model_fqandtoyo.py:1697-1699) through the
    reference's EncoderLayer, both variants: y, dx, dbias and gradient norms."""
    import model as rmodel
    import model_fqandtoyo as rfq
    from inputs import MASK_CASES, mask_case
    out = {}
    for variant, mod in (("stock", rmodel), ("fq", rfq)):
        for cname, C, T, G, ffn in MASK_CASES:
            name = f"{variant}/{cname}"
            seed, x_np, bias_np, gy_np, _, mask_np = mask_case(variant, C, T, G)
            layer = mod.EncoderLayer(C, ffn, 0.1, 0.1, 8).eval()
            fill_params(layer, seed + 1)
            x = torch.from_numpy(x_np).requires_grad_(True)
            bias = torch.from_numpy(bias_np).requires_grad_(True)
            y = layer(x, bias, mask=torch.from_numpy(mask_np))
            y.backward(torch.from_numpy(gy_np))
            out[f"{name}/y"] = y.detach().numpy()
            out[f"{name}/dx"] = x.grad.numpy()
            out[f"{name}/dbias"] = bias.grad.numpy()
            out[f"{name}/mask_count"] = np.array(int(mask_np.sum()))
            for pn, p in layer.named_parameters():
                if p.grad is not None:
                    g = p.grad.double()
                    out[f"{name}/gstat/{pn}"] = np.array([g.sum().item(), g.norm().item()])
    save("g9_mask.npz", **out)


def build_items(trajs):
    import wrapper
    from mobgt_amd import synth
    return [wrapper.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(trajs)]


STOCK_ARGS = dict(n_layers=2, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1,
                  weight_decay=0.01, ffn_dim=256, dataset_name="synthetic", warmup_updates=10, tot_updates=100,
                  peak_lr=2e-4, end_lr=1e-9, edge_type="multi_hop", multi_hop_max_dist=20,
                  attention_dropout_rate=0.1)


def make_g5_g6():
    import collator as rcoll
    import model as rmodel
    import model_fqandtoyo as rfq
    from mobgt_amd import synth

    uni = synth.make_universe(P=64, n_cat=8, n_user=8, seed=3)
    write_universe(uni)
    trajs = synth.make_batch_of_trajectories(seed=9, G=6, P=64, n_user=8, cat_of_poi=uni.cat_of_poi,
                                             n_nodes=[4, 12, 2, 7, 9, 3])
    out5, out6 = {}, {}
    out6.update(traj_arrays("traj", trajs))
    out6["uni/poi_table"] = uni.poi_table
    out6["uni/graph_adj"] = uni.graph_adj
    out6["uni/graph_dist"] = uni.graph_dist
    out6["uni/graph_cat"] = uni.graph_cat
    out6["uni/distance"] = uni.distance

    # ---------------- stock model.py Graphormer (num_class from the stubbed registry = 65)
    items = build_items(trajs)
    batch = rcoll.collator(copy.deepcopy(items), max_node=512, multi_hop_max_dist=20, rel_pos_max=1024)
    import data as rdata
    old = rdata.get_dataset
    rdata.get_dataset = lambda name: {"num_class": 65, "evaluator": None, "metric": "x",
                                      "loss_fn": torch.nn.NLLLoss(ignore_index=0)}
    rmodel.get_dataset = rdata.get_dataset
    m = rmodel.Graphormer(**STOCK_ARGS).eval()
    fill_params(m, 77)
    captured = {}
    h = m.layers[0].register_forward_pre_hook(lambda mod, args: captured.__setitem__("bias", args[1].detach().clone()))
    logits = m(batch)
    h.remove()
    out5.update(batch_arrays("stock/batch/", batch))
    out5["stock/bias"] = captured["bias"].numpy()
    out5["stock/seed"] = np.array(77)
    out6["stock/logits"] = logits.detach().numpy()
    # one train-mode-free "train step": eval-mode forward (no dropout), NLL of log_softmax, backward
    m.zero_grad()
    logits = m(batch)
    y = batch.y.view(-1)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    out6["stock/loss"] = np.array(loss.item())
    for pn, p in m.named_parameters():
        if p.grad is None:
            out6[f"stock/grad_none/{pn}"] = np.array(1)
        else:
            g = p.grad.double()
            out6[f"stock/gstat/{pn}"] = np.array([g.sum().item(), g.norm().item()])
            if p.grad.numel() <= 65536:
                out6[f"stock/grad/{pn}"] = grad_sample(p.grad.numpy())
    # bias gradient parity inputs: d(graph_attn_bias) -> table grads for a fixed upstream grad
    m.zero_grad()
    rng = np.random.RandomState(5)
    gb = torch.from_numpy(rng.standard_normal(tuple(captured["bias"].shape)).astype(np.float32))
    h = m.layers[0].register_forward_pre_hook(lambda mod, args: captured.__setitem__("bias_t", args[1]))
    m(batch)
    h.remove()
    bt = captured["bias_t"]
    finite = torch.isfinite(bt)
    (torch.where(finite, bt, torch.zeros_like(bt)) * gb).sum().backward()
    out5["stock/gbias"] = gb.numpy()
    for pn in ("rel_pos_encoder.weight", "edge_encoder.weight", "edge_dis_encoder.weight",
               "graph_token_virtual_distance.weight"):
        g = dict(m.named_parameters())[pn].grad
        out5[f"stock/dtable/{pn}"] = g.numpy() if g.numel() <= 4096 else g.numpy()[:4096]
    rdata.get_dataset = old

    # ---------------- fq variants (foursquaregraph + gowalla_nevda)
    for ds, coll in (("foursquaregraph", rcoll.collator_foursquare), ("gowalla_nevda", rcoll.collator_gowalla)):
        items = build_items(trajs)
        with in_ws():
            batch = coll(copy.deepcopy(items), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
            args = dict(STOCK_ARGS)
            args["dataset_name"] = ds
            m = rfq.Graphormer(**args).eval()
        nb = m.poi_pos_encoder.weight.shape[0]
        batch.poi_pos = batch.poi_pos.clamp(max=nb - 1)      # avoid the reference's latent OOB
        fill_params(m, 78)
        captured = {}
        h = m.layers[0].register_forward_pre_hook(lambda mod, a: captured.__setitem__("bias", a[1].detach().clone()))
        outs = m(batch)
        h.remove()
        tag = "fsq" if ds == "foursquaregraph" else "gow"
        out5.update(batch_arrays(f"{tag}/batch/", batch, skip=("feature_matrix",)))
        out5[f"{tag}/bias"] = captured["bias"].numpy()
        out5[f"{tag}/num_bins"] = np.array(nb)
        out6[f"{tag}/logits"] = outs[0].detach().numpy()
        out6[f"{tag}/cat_logits"] = outs[1].detach().numpy()
        out6[f"{tag}/seed"] = np.array(78)
        out6[f"{tag}/param_names"] = np.array([n for n, _ in m.named_parameters()])
        out6[f"{tag}/param_shapes"] = np.array([str(tuple(p.shape)) for _, p in m.named_parameters()])
        m.zero_grad()
        with cpu_cuda_alias():
            loss = m.training_step(batch, 0)
        loss.backward()
        out6[f"{tag}/loss"] = np.array(loss.item())
        for pn, p in m.named_parameters():
            if p.grad is None:
                out6[f"{tag}/grad_none/{pn}"] = np.array(1)
            else:
                g = p.grad.double()
                out6[f"{tag}/gstat/{pn}"] = np.array([g.sum().item(), g.norm().item()])
                if p.grad.numel() <= 65536:
                    out6[f"{tag}/grad/{pn}"] = grad_sample(p.grad.numpy())
    save("g5_bias.npz", **out5)
    save("g6_e2e.npz", **out6)


def make_g11():
    """The `toyotagraph` branch of the fq Graphormer (model_fqandtoyo.py:902-1039 constructor, :1417-1428 log_softmax head,
    :1462-1471 loss = GradientTailLoss(category logits, category of the target, 0.1) + NLLLoss(ignore_index=0)) on the synthetic
    universe of G5 / G6 -- the Toyota data itself is private (README.md:72-83): `collator_toyota`, outputs, loss, gradients."""
    import collator as rcoll
    import model_fqandtoyo as rfq
    from mobgt_amd import synth
    uni = synth.make_universe(P=64, n_cat=8, n_user=8, seed=3)
    write_universe(uni)
    trajs = synth.make_batch_of_trajectories(seed=9, G=6, P=64, n_user=8, cat_of_poi=uni.cat_of_poi, n_nodes=[4, 12, 2, 7, 9, 3])
    out = {}
    items = build_items(trajs)
    with in_ws():
        batch = rcoll.collator_toyota(copy.deepcopy(items), max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
        args = dict(STOCK_ARGS)
        args["dataset_name"] = "toyotagraph"
        m = rfq.Graphormer(**args).eval()
    nb = m.poi_pos_encoder.weight.shape[0]
    batch.poi_pos = batch.poi_pos.clamp(max=nb - 1)      # avoid the reference's latent OOB
    fill_params(m, 79)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                  # (F.log_softmax without dim, :1425)
        outs = m(batch)
        out.update(batch_arrays("toy/batch/", batch, skip=("feature_matrix",)))
        out["toy/num_bins"] = np.array(nb)
        out["toy/logits"] = outs[0].detach().numpy()         # log-probabilities
        out["toy/cat_logits"] = outs[1].detach().numpy()
        out["toy/seed"] = np.array(79)
        out["toy/param_names"] = np.array([n for n, _ in m.named_parameters()])
        out["toy/param_shapes"] = np.array([str(tuple(p.shape)) for _, p in m.named_parameters()])
        m.zero_grad()
        with cpu_cuda_alias():
            loss = m.training_step(batch, 0)
    loss.backward()
    out["toy/loss"] = np.array(loss.item())
    out["toy/cat_target"] = m.cat_target.detach().numpy()
    for pn, p in m.named_parameters():
        if p.grad is None:
            out[f"toy/grad_none/{pn}"] = np.array(1)
        else:
            g = p.grad.double()
            out[f"toy/gstat/{pn}"] = np.array([g.sum().item(), g.norm().item()])
            if p.grad.numel() <= 65536:
                out[f"toy/grad/{pn}"] = grad_sample(p.grad.numpy())
    save("g11_toyota.npz", **out)


def make_g7():
    import lr as rlr
    import model_fqandtoyo as rfq
    out = {}
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=2e-4)
    base_init = torch.optim.lr_scheduler.LRScheduler.__init__
    # torch>=2.7 dropped the `verbose` positional the reference still passes (lr.py:15)
    torch.optim.lr_scheduler.LRScheduler.__init__ = lambda self, o, last_epoch=-1, verbose=False: base_init(self, o, last_epoch)
    sched = rlr.PolynomialDecayLR(opt, warmup_updates=40, tot_updates=400, lr=2e-4, end_lr=1e-9, power=1.0)
    lrs = [opt.param_groups[0]["lr"]]
    for _ in range(450):
        opt.step()
        sched.step()
        lrs.append(opt.param_groups[0]["lr"])
    out["lr/values"] = np.array(lrs, dtype=np.float64)
    out["lr/args"] = np.array([40, 400, 2e-4, 1e-9, 1.0])
    rng = np.random.RandomState(4)
    logits = torch.from_numpy((rng.standard_normal((5, 33)) * 2).astype(np.float32)).requires_grad_(True)
    targets = torch.from_numpy(rng.randint(0, 33, size=5))
    with cpu_cuda_alias():
        loss = rfq.GradientTailLoss(logits, targets, 0.2)
    loss.backward()
    out["gtl/logits"] = logits.detach().numpy()
    out["gtl/targets"] = targets.numpy()
    out["gtl/loss"] = np.array(loss.item())
    out["gtl/dlogits"] = logits.grad.numpy()
    # evaluation metrics: get_acc / MRR_metric
    scores = torch.from_numpy(rng.standard_normal((12, 40)).astype(np.float32))
    target = torch.from_numpy(rng.randint(1, 40, size=12))
    acc, ndcg = rfq.get_acc(target, scores)
    out["acc/scores"] = scores.numpy()
    out["acc/target"] = target.numpy()
    out["acc/acc"] = np.asarray(acc, dtype=np.float64)
    out["acc/ndcg"] = np.asarray(ndcg, dtype=np.float64)
    try:
        out["acc/mrr"] = np.asarray(rfq.MRR_metric(target, scores), dtype=np.float64)
    except Exception as e:  # signature differs; recorded for the record
        print("MRR_metric skipped:", e)
    save("g7_lr_loss.npz", **out)


def run(which):
    torch.manual_seed(0)
    if "g4" in which:
        make_g4()
    if "g5" in which or "g6" in which:
        make_g5_g6()
    if "g7" in which:
        make_g7()
    if "g9" in which:
        make_g9()
    if "g11" in which:
        make_g11()

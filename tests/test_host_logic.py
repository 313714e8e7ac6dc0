"""CPU-side checks of the product's host logic: the C-ABI library loads and exports every symbol of
include/mobgt_hip.h (no compute without a GPU), the drop-in collators reproduce the reference's batches
(goldens G2/G3), LR schedule / loss / state-dict names match the reference (G6/G7)."""
import os
import re

import numpy as np
import pytest
import torch

from mobgt_amd import _lib, synth, collator as pc
from oracle import collator_oracle as co

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    _lib.build()
    hdr = open(os.path.join(ROOT, "include", "mobgt_hip.h")).read()
    declared = set(re.findall(r"\b(mobgt_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    handle = _lib.lib()
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in mobgt_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert handle.mobgt_abi_version() == _lib.ABI_VERSION == 3
    assert b"gfx950" in handle.mobgt_build_info()
    # pure host helper: deterministic keep rule
    a = [handle.mobgt_dropout_keep_host(42, 8, 33, 1, 2, 3, j, 0.1) for j in range(2000)]
    assert a == [handle.mobgt_dropout_keep_host(42, 8, 33, 1, 2, 3, j, 0.1) for j in range(2000)]
    assert 0.05 < 1 - np.mean(a) < 0.15
    assert all(handle.mobgt_dropout_keep_host(42, 8, 33, 1, 2, 3, j, 0.0) for j in range(50))


def test_bulk_dropout_mask_replay_matches_the_per_element_rule_and_the_oracle_hook():
    """mobgt_attn_dropout_mask_host == mobgt_dropout_keep_host elementwise; mobgt_dropout_mask_host: rows are numbered
    from row0 (a slice of a mask equals the mask of the slice), p = 0 keeps everything; the oracle's `drop` hook receives
    every site of a training-mode layer and is inert in eval mode."""
    from mobgt_amd import ops
    from oracle import model_oracle as mo
    handle = _lib.lib()
    G, H, T = 2, 3, 37
    m = ops.dropout_keep_mask(0xABCDEF0123456789, G, H, T, 0.1)
    ref = np.array([[[[handle.mobgt_dropout_keep_host(0xABCDEF0123456789, H, T, g, h, i, j, 0.1) for j in range(T)]
                      for i in range(T)] for h in range(H)] for g in range(G)], dtype=bool)
    assert np.array_equal(m, ref) and 0.05 < 1 - m.mean() < 0.15
    full = ops.dropout_site_mask(77, 0x1003, 40, 192, 0.1)
    assert np.array_equal(full[13:29], ops.dropout_site_mask(77, 0x1003, 16, 192, 0.1, row0=13))
    assert 0.07 < 1 - full.mean() < 0.13 and ops.dropout_site_mask(77, 0x1003, 5, 8, 0.0).all()
    assert not np.array_equal(full, ops.dropout_site_mask(78, 0x1003, 40, 192, 0.1))          # next step: new mask
    assert not np.array_equal(full, ops.dropout_site_mask(77, 0x1004, 40, 192, 0.1))          # other site: other mask
    torch.manual_seed(0)
    C, Hh = 16, 2
    sd = {}
    for name, shape in (("self_attention.linear_q", (C, C)), ("self_attention.linear_k", (C, C)), ("self_attention.linear_v", (C, C)),
                        ("self_attention.output_layer", (C, C)), ("ffn.layer1", (32, C)), ("ffn.layer2", (C, 32))):
        sd[f"L.{name}.weight"], sd[f"L.{name}.bias"] = torch.randn(*shape) * 0.2, torch.zeros(shape[0])
    for n in ("ffn_norm1", "ffn_norm2"):
        sd[f"L.{n}.weight"], sd[f"L.{n}.bias"] = torch.ones(C), torch.zeros(C)
    x = torch.randn(2, 5, C)
    seen = []

    def drop(t, p, site):
        seen.append((site, tuple(t.shape), p))
        return t
    y_eval = mo.encoder_layer_fq(sd, "L", x, None, Hh, 0.1, 0.2, False, drop=drop)
    assert not seen
    y_train = mo.encoder_layer_fq(sd, "L", x, None, Hh, 0.1, 0.2, True, drop=drop)
    assert [s[0] for s in seen] == [("L.self_attention", "att"), ("L", "res1"), ("L", "res2")]
    assert seen[0][1] == (2, Hh, 5, 5) and seen[0][2] == 0.2 and seen[1][1] == (2, 5, C)
    assert torch.equal(y_eval, y_train)                # an all-keep hook is the eval-mode layer


def test_ops_refuse_cpu_tensors():
    from mobgt_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.pack_bias(torch.zeros(1, 8, 4, 4), 1, 8, 4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.attention(torch.zeros(1, 4, 128), torch.zeros(1, 4, 128), torch.zeros(1, 4, 128), None, 0.25)


def _items(z):
    trajs = [{k: z[f"traj{i}/{k}"] for k in ("node_name", "edge_type", "target", "time", "time_normal", "user", "cat")}
             for i in range(int(z["trajcount"]))]
    return [co.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(trajs)]


def _cmp(z, prefix, b, fields):
    for f in fields:
        ref, got = z[f"{prefix}{f}"], getattr(b, f).numpy()
        assert got.shape == ref.shape, (f, got.shape, ref.shape)
        assert np.array_equal(got.astype(ref.dtype) if ref.dtype.kind != "f" else got, ref), f


STOCK = ("idx", "attn_bias", "attn_edge_type", "rel_pos", "in_degree", "out_degree", "x", "edge_input", "y", "adj")


def test_stock_collator_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "g2_collator.npz"))
    items = _items(z)
    b = pc.collator(items, max_node=512, multi_hop_max_dist=20, rel_pos_max=1024)
    assert isinstance(b, pc.Batch) and len(b) == 8
    _cmp(z, "stock/", b, STOCK)
    _cmp(z, "stock_masked/", pc.collator(items, max_node=12, multi_hop_max_dist=5, rel_pos_max=3), STOCK)
    assert b.to("cpu") is b


def test_poi_collators_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "g3_collator_fq.npz"))
    items = _items(z)
    pc.register_poi_distance("tky_distance.pkl", z["distance"])
    pc.register_poi_distance("gowalla_distance.pkl", z["distance"])
    fields = STOCK + ("time", "adj1", "time_normal", "user", "cat", "poi_pos")
    bf = pc.collator_foursquare(items, max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)
    _cmp(z, "fsq/", bf, fields)
    _cmp(z, "gow/", pc.collator_gowalla(items, max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024), fields)
    assert tuple(bf.feature_matrix.shape) == tuple(z["fsq/feature_matrix_shape"])
    assert pc.poi_distance("tky_distance.pkl")["num_bins"] == int(z["num_bins"])


def test_lr_schedule_and_loss_match_reference(golden_dir):
    from mobgt_amd.lr import PolynomialDecayLR
    from mobgt_amd.model_fqandtoyo import GradientTailLoss
    z = np.load(os.path.join(golden_dir, "g7_lr_loss.npz"))
    w, t, lr, end, power = z["lr/args"]
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=lr)
    sched = PolynomialDecayLR(opt, warmup_updates=int(w), tot_updates=int(t), lr=lr, end_lr=end, power=power)
    got = [opt.param_groups[0]["lr"]]
    for _ in range(len(z["lr/values"]) - 1):
        opt.step()
        sched.step()
        got.append(opt.param_groups[0]["lr"])
    np.testing.assert_allclose(got, z["lr/values"], rtol=1e-12)
    logits = torch.from_numpy(z["gtl/logits"]).requires_grad_(True)
    loss = GradientTailLoss(logits, torch.from_numpy(z["gtl/targets"]), 0.2)
    loss.backward()
    np.testing.assert_allclose(loss.item(), z["gtl/loss"], rtol=1e-6)
    np.testing.assert_allclose(logits.grad.numpy(), z["gtl/dlogits"], rtol=1e-5, atol=1e-8)


def test_state_dict_names_match_reference(golden_dir):
    """Reference checkpoints must load: same parameter names and shapes, in the same order."""
    from mobgt_amd.model_fqandtoyo import Graphormer
    z6 = np.load(os.path.join(golden_dir, "g6_e2e.npz"))
    uni = synth.Universe(P=64, n_cat=8, n_user=8, poi_table=z6["uni/poi_table"], graph_adj=z6["uni/graph_adj"],
                         graph_dist=z6["uni/graph_dist"], graph_cat=z6["uni/graph_cat"], distance=z6["uni/distance"])
    args = dict(n_layers=2, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                ffn_dim=256, warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9, edge_type="multi_hop",
                multi_hop_max_dist=20, attention_dropout_rate=0.1)
    for tag, ds in (("fsq", "foursquaregraph"), ("gow", "gowalla_nevda")):
        m = Graphormer(dataset_name=ds, universe=uni, **args)
        names = [n for n, _ in m.named_parameters()]
        shapes = [str(tuple(p.shape)) for _, p in m.named_parameters()]
        assert names == [str(n) for n in z6[f"{tag}/param_names"]]
        assert shapes == [str(s) for s in z6[f"{tag}/param_shapes"]]
    # the toyotagraph branch (golden G11: 996 user rows, no cat_embed_model, a num_cats-way category head)
    z11 = np.load(os.path.join(golden_dir, "g11_toyota.npz"))
    m = Graphormer(dataset_name="toyotagraph", universe=uni, **args)
    assert [n for n, _ in m.named_parameters()] == [str(n) for n in z11["toy/param_names"]]
    assert [str(tuple(p.shape)) for _, p in m.named_parameters()] == [str(s) for s in z11["toy/param_shapes"]]
    assert m.time_embed_model_48.padding_idx is None and m.user_embed_model.user_embedding.num_embeddings == 996


def test_metrics_match_reference_g7(golden_dir):
    from mobgt_amd import metrics
    z = np.load(os.path.join(golden_dir, "g7_lr_loss.npz"))
    scores, target = torch.from_numpy(z["acc/scores"]), torch.from_numpy(z["acc/target"])
    acc, ndcg = metrics.get_acc(target, scores)
    np.testing.assert_allclose(acc, z["acc/acc"])
    np.testing.assert_allclose(ndcg, z["acc/ndcg"], rtol=1e-12)
    np.testing.assert_allclose(metrics.MRR_metric(target, scores), z["acc/mrr"], rtol=1e-12)
    # the reference stops at the first zero target (model_fqandtoyo.py:88-89)
    t2 = target.clone()
    t2[5] = 0
    a2, _ = metrics.get_acc(t2, scores)
    a5, _ = metrics.get_acc(target[:5], scores[:5])
    np.testing.assert_allclose(a2, a5)
    out = metrics.evaluate_outputs([{"y_pred": [scores, None], "y_true": target}])
    assert abs(out["acc@10"] - z["acc/acc"][0, 0] / len(target)) < 1e-12 and 0 <= out["mrr"] <= 1


def test_lightning_checkpoint_round_trip(tmp_path):
    """entry.py:71-93: a Lightning .ckpt keeps the model's parameters under "state_dict" with the reference's
    names; loading copies into the existing (possibly QKV-fused) parameters, strict=False like the reference."""
    from mobgt_amd import checkpoint
    from mobgt_amd.model import EncoderLayer
    torch.manual_seed(0)
    a, b = EncoderLayer(64, 128, 0.1, 0.1, 8), EncoderLayer(64, 128, 0.1, 0.1, 8)
    b.self_attention.fuse_qkv_storage()                       # destination already in the fused layout
    path = tmp_path / "m.ckpt"
    checkpoint.save_lightning_checkpoint(a, path, epoch=3, hyper_parameters={"n_layers": 1})
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"state_dict", "epoch", "hyper_parameters"}
    ck["state_dict"]["not_in_model.weight"] = torch.zeros(2)
    torch.save(ck, path)
    missing, unexpected = checkpoint.load_lightning_checkpoint(b, path)
    assert list(unexpected) == ["not_in_model.weight"] and list(missing) == []
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    wq = b.self_attention.linear_q.weight
    wqkv, _ = b.self_attention.fuse_qkv_storage()
    assert wq.data_ptr() == wqkv.data_ptr() and torch.equal(wqkv[:64], a.self_attention.linear_q.weight)
    ck["state_dict"]["ffn.layer1.weight"] = torch.zeros(3, 3)
    with pytest.raises(ValueError):
        checkpoint.load_lightning_checkpoint(b, ck)

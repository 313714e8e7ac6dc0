"""Pins the oracle to the reference's outputs on REAL Gowalla data (tests/golden/g8_gowalla_real.npz, produced by
tests/golden/make_golden_real.py from /root/reference/gowalla_nevda.7z): eight real trajectories with N = 1, 2, 5, 17,
94, 8, 12, 30 and the 329-node one, the real POI universe (P 3 679, 253 categories, 653 distance bins), the
gowalla_nevda fq Graphormer at BASELINE configs[2] sizes (hidden 128, 6 layers, 8 heads, ffn 1024).
Integer work bit-exact; fp32 model outputs at fp32 tolerances."""
import os

import numpy as np
import pytest
import torch

from inputs import real_universe, real_trajs
from gradcheck import grad_sample
from mobgt_amd import synth
from oracle import algos_oracle as ao
from oracle import collator_oracle as co
from oracle import model_oracle as mo
from test_oracle_model import seeded_state

FIELDS = ("idx", "attn_bias", "rel_pos", "in_degree", "out_degree", "x", "y", "time", "time_normal", "user", "cat", "poi_pos")


@pytest.fixture(scope="module")
def g8(golden_dir):
    z = np.load(os.path.join(golden_dir, "g8_gowalla_real.npz"))
    return z, real_universe(z)


def test_real_graphs_algos_and_preprocess_item_g8(g8):
    z, _ = g8
    for tag in ("a", "b"):
        for i, t in enumerate(real_trajs(z, tag)):
            p = f"{tag}/item{i}/"
            c = t["edge_type"]
            M, path = ao.floyd_warshall(c != 0)
            assert np.array_equal(M, z[p + "M"]) and np.array_equal(path, z[p + "path"]), p
            it = co.preprocess_item(synth.trajectory_to_item(t, idx=i))
            assert np.array_equal(it.rel_pos.numpy(), z[p + "rel_pos"]), p
            assert tuple(it.edge_input.shape) == tuple(z[p + "edge_input_shape"]), p
            ei = it.edge_input[:, :, :20].numpy()
            assert np.array_equal(ei if ei.shape[0] <= 100 else ei[::7], z[p + "edge_input20"]), p
            assert it.edge_input.numpy().astype(np.float64).sum() == float(z[p + "edge_input_sum"]), p
            for f in ("in_degree", "out_degree", "x", "user"):
                assert np.array_equal(getattr(it, f).numpy(), z[p + f]), (p, f)


def _collate(z, uni, tag):
    items = [co.preprocess_item(synth.trajectory_to_item(t, idx=i)) for i, t in enumerate(real_trajs(z, tag))]
    return co.collator_poi(items, uni.distance, max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024)


def check_batch(z, tag, b, to_np=lambda t: t.numpy()):
    """Every field `collator_gowalla` produced for the batch, bit for bit (`b/edge_input`: every 7th query row + checksum)."""
    for f in FIELDS:
        ref, got = z[f"{tag}/batch/{f}"], to_np(getattr(b, f))
        assert got.shape == ref.shape, (f, got.shape, ref.shape)
        assert np.array_equal(got.astype(ref.dtype) if ref.dtype.kind != "f" else got, ref), (tag, f)
    ei = to_np(b.edge_input)
    if tag == "a":
        assert np.array_equal(ei.astype(np.int64), z["a/batch/edge_input"].astype(np.int64))
    else:
        assert np.array_equal(ei[:, ::7].astype(np.int64), z["b/batch/edge_input_rows7"].astype(np.int64))
        assert ei.astype(np.float64).sum() == float(z["b/batch/edge_input_sum"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_real_collator_gowalla_g8(g8, tag):
    z, uni = g8
    dm = np.delete(np.delete(uni.distance, 0, axis=0), 0, axis=1)
    nb, edges = co.freedman_diaconis_bins(dm - dm.min(), True)
    assert nb == int(z["num_bins"]) and np.array_equal(edges, z["bin_edges"])
    check_batch(z, tag, _collate(z, uni, tag))


def test_real_fq_forward_loss_grads_g8(g8):
    z, uni = g8
    consts = mo.fq_constants(uni, "gowalla_nevda", num_bins=int(z["num_bins"]))
    names = [str(n) for n in z["param_names"]]
    shapes = [eval(str(s)) for s in z["param_shapes"]]
    sd = seeded_state(list(zip(names, shapes)), int(z["seed"]))
    kw = dict(n_layers=6, H=8, D=20)
    ba, bb = _collate(z, uni, "a"), _collate(z, uni, "b")
    with torch.no_grad():
        bias = mo.assemble_bias(sd, ba, 8, 20, "fq").numpy()[:, :, ::7]
        ref = z["a/bias_rows7"]
        fin = np.isfinite(ref)
        assert np.array_equal(np.isfinite(bias), fin)
        np.testing.assert_allclose(bias[fin], ref[fin], rtol=2e-5, atol=2e-6)
        for tag, b in (("a", ba), ("b", bb)):
            logits, cat_logits = mo.graphormer_fq_forward(sd, b, consts, **kw)
            np.testing.assert_allclose(logits.numpy(), z[f"{tag}/logits"], rtol=2e-4, atol=2e-5)
            np.testing.assert_allclose(cat_logits.numpy(), z[f"{tag}/cat_logits"], rtol=2e-4, atol=2e-5)
    loss = mo.fq_training_loss(sd, ba, consts, **kw)
    np.testing.assert_allclose(loss.item(), z["a/loss"], rtol=1e-5)
    loss.backward()
    for pn, p in sd.items():
        if f"a/grad_none/{pn}" in z:
            assert p.grad is None, pn
            continue
        g = p.grad.double()
        np.testing.assert_allclose([g.sum().item(), g.norm().item()], z[f"a/gstat/{pn}"], rtol=3e-3, atol=1e-6, err_msg=pn)
        if f"a/grad/{pn}" in z:
            refg = z[f"a/grad/{pn}"]
            np.testing.assert_allclose(grad_sample(p.grad.numpy()), refg, rtol=5e-3, atol=1e-3 * float(np.abs(refg).max()) + 1e-9, err_msg=pn)
    # the edge tables at GradScaler's loss x 65536 (their small per-pair gradients survive the reference's .half() casts there)
    for p in sd.values():
        p.grad = None
    (mo.fq_training_loss(sd, ba, consts, **kw) * 65536.0).backward()
    for pn in ("edge_encoder.weight", "edge_dis_encoder.weight"):
        refg = z[f"a_s65536/grad/{pn}"]
        np.testing.assert_allclose(grad_sample((sd[pn].grad / 65536.0).numpy()), refg, rtol=5e-3, atol=1e-3 * float(np.abs(refg).max()) + 1e-9, err_msg=pn)


# ------------------------------------------------------------------------------------------------ G10: a training trajectory
def g10_batches(z, uni, split, n_batches, batch=16):
    trajs = real_trajs(z, split)
    out = []
    for s in range(n_batches):
        items = [co.preprocess_item(synth.trajectory_to_item(t, idx=s * batch + i)) for i, t in enumerate(trajs[s * batch:(s + 1) * batch])]
        out.append(co.collator_poi(items, uni.distance, max_node=30000, multi_hop_max_dist=20, rel_pos_max=1024))
    return out


def g10_metrics(batches_logits_targets):
    """test_epoch_end's bookkeeping (model_fqandtoyo.py:1546-1597) with the oracle's metric functions."""
    tot, mrr, n = np.zeros(8), 0.0, 0
    for logits, y_true in batches_logits_targets:
        a, d = mo.get_acc(y_true, logits)
        a, d = np.asarray(a).reshape(4), np.asarray(d).reshape(4)
        tot += np.array([a[2], a[1], a[0], d[2], d[1], d[0], a[3], d[3]])
        mrr += float(mo.mrr_metric(y_true, logits))
        n += len(y_true)
    return tot / n, mrr / n


def test_training_trajectory_of_30_updates_g10(g8, golden_dir):
    """Golden G10 (tests/golden/make_golden_traj.py): the REFERENCE trained for 30 AdamW updates (PolynomialDecayLR, warm-up 10,
    peak 1e-3) on 480 real Gowalla trajectories, then evaluated on 256 real test trajectories.  The oracle -- its forward /
    loss, torch's AdamW on its parameter dict, its restatement of the schedule (lr.py:17-31) -- must walk the same trajectory:
    the learning rate of every update exactly, the loss of every update within 2e-4 relative (fp32 round-off through 30
    updates; measured: see the printout), a sample of every parameter after update 30 within 2 % of that parameter's MOVEMENT
    (rms of final - initial; linear_k.bias, whose gradient is exactly zero in exact arithmetic, excepted), the test logits of the
    first 16 test trajectories within 1e-3, and test_epoch_end's metrics."""
    z8, uni = g8
    z = np.load(os.path.join(golden_dir, "g10_traj.npz"))
    steps, batch, n_test = (int(v) for v in z["args/steps_batch_ntest"])
    warm, tot, peak, end, wd = (float(v) for v in z["args/lr"])
    consts = mo.fq_constants(uni, "gowalla_nevda", num_bins=int(z8["num_bins"]))
    names = [str(n) for n in z["param_names"]]
    shapes = [eval(str(s)) for s in z["param_shapes"]]
    sd = seeded_state(list(zip(names, shapes)), int(z["seed"]))
    init = {k: v.detach().clone() for k, v in sd.items()}
    opt = torch.optim.AdamW(list(sd.values()), lr=peak, weight_decay=wd)
    kw = dict(n_layers=6, H=8, D=20)
    worst = 0.0
    for s, b in enumerate(g10_batches(z, uni, "train", steps, batch)):
        lr = mo.polynomial_decay_lr(s + 1, warm, tot, peak, end, 1.0)
        assert lr == float(z["lrs"][s]), (s, lr)
        for g in opt.param_groups:
            g["lr"] = lr
        loss = mo.fq_training_loss(sd, b, consts, **kw)
        opt.zero_grad()
        loss.backward()
        opt.step()
        rel = abs(loss.item() - float(z["losses"][s])) / float(z["losses"][s])
        worst = max(worst, rel)
        assert rel <= 2e-4, (s, loss.item(), float(z["losses"][s]))
    print("largest relative loss deviation over 30 updates: %.2e" % worst)
    from traj_helpers import param_sample
    for pn, p in sd.items():
        if pn.endswith("linear_k.bias"):
            continue
        ref = z[f"final/{pn}"]
        got, ini = param_sample(p.detach().numpy()), param_sample(init[pn].numpy())
        move = float(np.sqrt(((ref - ini) ** 2).mean()))
        assert float(np.abs(got - ref).max()) <= 2e-2 * move + 1e-7, (pn, float(np.abs(got - ref).max()), move)
    with torch.no_grad():
        evals = []
        for s, b in enumerate(g10_batches(z, uni, "test", n_test // batch, batch)):
            logits, _ = mo.graphormer_fq_forward(sd, b, consts, **kw)
            if s == 0:
                np.testing.assert_allclose(logits.numpy(), z["test/logits0"], rtol=1e-3, atol=1e-3)
            evals.append((logits, b.y - 1))
    acc, mrr = g10_metrics(evals)
    np.testing.assert_allclose(acc, z["metrics/acc1_5_10_ndcg1_5_10_acc20_ndcg20"], atol=1.0 / n_test + 1e-12)
    np.testing.assert_allclose(mrr, float(z["metrics/mrr"]), rtol=2e-2)

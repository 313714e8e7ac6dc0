"""Parity on REAL Gowalla data on the MI355X (golden G8, tests/golden/make_golden_real.py): the reference's own outputs for
eight real trajectories (N = 1, 2, 5, 17, 94, 8, 12, 30) and the 329-node one on the real POI universe (P 3 679, 253
categories, 653 distance bins), gowalla_nevda fq Graphormer at BASELINE configs[2] sizes.

Tolerances: index tensors (SPD, paths, edge features, degrees, poi_pos bins) bit-exact.  Model outputs: the f32
configuration is fp32 end to end (attention: csrc/attn_f32_body.h) -> logits at 2e-4 absolute / relative against the
reference's fp32 values (measured 8e-6 on logits up to 5.1), loss 1e-5 (measured: equal to 16 digits), gradients as
tests/test_gpu_model.py::_check_grads: every parameter elementwise, relative L2 <= 0.2 % (measured < 5e-5); the edge tables
against the reference's backward at GradScaler's loss x 65536 (at the plain loss the reference's own fp16 casts flush 8 % of
edge_encoder's gradient: make_golden_real.py).  The bf16 configuration -- what bench.py times -- is held to the oracle with
replayed masks and head branch pattern in tests/test_gpu_bench_parity.py / test_gpu_train_parity.py (S-GOW batches).

These real trajectories are what exposed the un-cancelled delta of rounds 1-4 (csrc/attn.hip header, "consistent softmax"):
every node of a trajectory carries the same user embedding, so K rows share a large common component and dQ / dK live on
sum_j dS_j = 0; with delta = rowsum(dO O) taken from differently rounded dO / O the top layers' q / k weight gradients were
10-50 % off here while every synthetic golden passed."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from inputs import real_universe, real_trajs                       # noqa: E402
from mobgt_amd import algos, synth, wrapper                      # noqa: E402
from mobgt_amd.data import DeviceCollator, make_bin_table       # noqa: E402
from test_oracle_model import seeded_state                        # noqa: E402
from test_oracle_real import check_batch                          # noqa: E402
from test_gpu_model import _check_grads                           # noqa: E402

DEV = "cuda"


@pytest.fixture(scope="module")
def g8(golden_dir):
    z = np.load(os.path.join(golden_dir, "g8_gowalla_real.npz"))
    uni = real_universe(z)
    num_bins, edges, table = make_bin_table(uni.distance)
    assert num_bins == int(z["num_bins"]) and np.array_equal(edges, z["bin_edges"])
    return z, uni, table


def test_real_graphs_device_algos_and_preprocess_item_g8(g8):
    """algos.floyd_warshall / gen_edge_input (csrc/spd.hip) and wrapper.preprocess_item on the real graphs, bit-exact."""
    z = g8[0]
    for tag in ("a", "b"):
        for i, t in enumerate(real_trajs(z, tag)):
            p = f"{tag}/item{i}/"
            M, path = algos.floyd_warshall(t["edge_type"] != 0)
            assert np.array_equal(M, z[p + "M"]) and np.array_equal(path, z[p + "path"]), p
            it = wrapper.preprocess_item(synth.trajectory_to_item(t, idx=i))
            assert np.array_equal(it.rel_pos.numpy(), z[p + "rel_pos"]), p
            assert tuple(it.edge_input.shape) == tuple(z[p + "edge_input_shape"]), p
            ei = it.edge_input[:, :, :20].numpy()
            assert np.array_equal(ei if ei.shape[0] <= 100 else ei[::7], z[p + "edge_input20"]), p
            assert it.edge_input.numpy().astype(np.float64).sum() == float(z[p + "edge_input_sum"]), p
            for f in ("in_degree", "out_degree", "x", "user"):
                assert np.array_equal(getattr(it, f).numpy(), z[p + f]), (p, f)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_device_collator_matches_reference_on_real_trajectories_g8(g8, tag):
    z, _, table = g8
    coll = DeviceCollator(DEV, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
    b = coll(real_trajs(z, tag))
    torch.cuda.synchronize()
    check_batch(z, tag, b, to_np=lambda t: t.cpu().numpy())


@pytest.fixture(scope="module")
def real_model(g8):
    from mobgt_amd.model_fqandtoyo import Graphormer
    z, uni, _ = g8
    m = Graphormer(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                   ffn_dim=1024, dataset_name="gowalla_nevda", warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9,
                   edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1, universe=uni)
    names = [str(n) for n in z["param_names"]]
    shapes = [eval(str(s)) for s in z["param_shapes"]]
    sd = {k: v.detach() for k, v in seeded_state(list(zip(names, shapes)), int(z["seed"])).items()}
    m.load_state_dict(sd, strict=True)                 # the reference's parameter names and shapes on the real universe
    return m.to(DEV).eval()


def test_fq_graphormer_on_real_universe_logits_loss_grads_g8(g8, real_model):
    z, _, table = g8
    m = real_model
    coll = DeviceCollator(DEV, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
    b = coll(real_trajs(z, "a"))
    bias = m.assemble_bias(b).dense().cpu().numpy()[:, :, ::7]
    ref = z["a/bias_rows7"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(bias), fin)
    np.testing.assert_allclose(bias[fin], ref[fin], rtol=1e-5, atol=1e-4)
    out = m(b)
    np.testing.assert_allclose(out[0].detach().cpu().numpy(), z["a/logits"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(out[1].detach().cpu().numpy(), z["a/cat_logits"], rtol=2e-4, atol=2e-4)
    loss = m.training_step(b, 0)                        # eval() mode, like the golden (no dropout)
    np.testing.assert_allclose(loss.item(), z["a/loss"], rtol=1e-5)
    loss.backward()
    _check_grads(m, z, "a", rtol=2e-3, edge_tag="a_s65536", lim_l2=2e-3)


def test_fq_graphormer_329_node_real_trajectory_logits_g8(g8, real_model):
    z, _, table = g8
    coll = DeviceCollator(DEV, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
    b = coll(real_trajs(z, "b"))
    with torch.no_grad():
        bias = real_model.assemble_bias(b).dense()[:, :1, ::31].cpu().numpy()
        out = real_model(b)
    ref = z["b/bias_rows31_h0"]
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(bias), fin)
    np.testing.assert_allclose(bias[fin], ref[fin], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(out[0].cpu().numpy(), z["b/logits"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(out[1].cpu().numpy(), z["b/cat_logits"], rtol=2e-4, atol=2e-4)


def test_benched_bf16_configuration_on_the_real_batch_vs_oracle_g8(g8):
    """the configuration bench.py TIMES (bf16 bias / GCN adjacency product / GEMM-facing activations, bf16 MFMA
    operands) on the real Gowalla batch A of G8, with stated bounds.  The comparison partner is the oracle on the same real batch
    (pinned to G8's logits / loss / gradients at 2e-4 / 1e-5 / 5e-3 by tests/test_oracle_real.py) rather than the fixture itself,
    because the head's LeakyReLU branch pattern must be replayed (tests/gradcheck.py: bounded -- <= 16 units, each within 5e-3 of
    the kink on both sides), which a stored gradient cannot do.

    What bf16 operands cost here (tools/g8_bf16_stages.py, profiles/r6_g8_bf16_stages.txt): every rounding point of a layer adds
    0.15-0.2 % relative L2 (xa 0.17 %, qkv 0.28 %, a 0.25 %, x1 0.30 %, z 0.34 %, u 0.41 %, h 0.47 %, x2 / out 0.49 %), layers add up
    in quadrature (0.50 % after layer 0 ... 0.99 % after layer 5), logits 1.1 % relative L2.  The absolute logit error (0.070) is
    larger than on the synthetic batches (3-7e-3) because G8's seeded weights give logits up to 5.06 (rms 1.33) where the
    default-initialised S-FSQ model's are O(0.1): the RELATIVE error is the same.

    Bounds (measured in round 6 with the head pattern replayed, in brackets): logits relative L2 <= 2 % [1.10 %] and
    max |err| <= 2e-2 x max |logit| [0.070 on 5.06 = 1.4e-2]; loss 1e-3 [1.9e-4]; gradients by relative L2, elementwise bound
    and exact zero pattern:
      * <= 4 % for every parameter that is not listed below [0.5-1.6 % for the 100+ layer / head / GCN / fuse tensors; 2.8 % on
        the distance GCN's second layer and 3.6 % on the two GCNs' FIRST layers, whose input is the raw, un-normalised POI
        feature matrix (check-in counts next to latitudes: model_fqandtoyo.py:650-700)];
      * <= 6 % for linear_q / linear_k (weight and bias) [2.7-5.4 %]: these gradients are 100-1000 x smaller than linear_v's
        (rms 2e-6 ... 5e-5 against 1e-3: every node of a trajectory carries the same user embedding, the scores barely depend on q
        and k) and live on sum_j dS_ij = 0.  A CPU emulation of the attention kernels' rounding points inside the oracle with
        everything else in fp32 (round 5's scratch emulation, re-run in round 6) gives 0.7-2.5 % for them from the attention's bf16
        operands alone; centring K per (graph, head) before its rounding, error-feedback rounding of dS along the key axis and
        unrounded dq / dk / dv were tried there and move nothing (+-0.3 %): the rest is the bf16 rounding of the layers' other
        GEMM operands reaching a heavily cancelled quantity.  VERDICT r5 asked for <= 3 %: NOT met on these;
      * <= 8 % for the small cancelling tables (rel_pos / poi_pos / time slots / virtual distance) [2.9-6.7 %], <= 12 % for the
        two edge tables [3.6 %; 7.3-7.8 % on the long batch of tests/test_gpu_long_parity.py, where the mechanism is spelled
        out]."""
    from gradcheck import assert_replay_bounded, device_head_pattern, replay_head
    from mobgt_amd.model_fqandtoyo import Graphormer
    from oracle import model_oracle as mo
    from test_gpu_bench_parity import LOSS_SCALE, SMALL_TABLES, check_grad
    from test_oracle_real import _collate
    z, uni, table = g8
    m = Graphormer(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
                   ffn_dim=1024, dataset_name="gowalla_nevda", warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9,
                   edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.1, universe=uni,
                   bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16)
    names = [str(n) for n in z["param_names"]]
    shapes = [eval(str(s)) for s in z["param_shapes"]]
    sd0 = {k: v.detach() for k, v in seeded_state(list(zip(names, shapes)), int(z["seed"])).items()}
    m.load_state_dict(sd0, strict=True)
    m = m.to(DEV).eval()
    b = DeviceCollator(DEV, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)(real_trajs(z, "a"))
    pattern, pre_dev = device_head_pattern(m, b)
    # ---- oracle on the same real batch, the device's head pattern imposed
    consts = mo.fq_constants(uni, "gowalla_nevda", num_bins=int(z["num_bins"]))
    sd = {k: v.detach().float().clone().requires_grad_(True) for k, v in sd0.items()}
    cb = _collate(z, uni, "a")
    seen = {}
    ref_logits, _ = mo.graphormer_fq_forward(sd, cb, consts, n_layers=6, H=8, D=20, act=replay_head(pattern, seen, pre_dev))
    print("head pre-activations: rms %.3f" % float(pre_dev.pow(2).mean().sqrt()), {k: v for k, v in seen.items() if v})
    n, worst = assert_replay_bounded(seen, pre_dev=pre_dev)           # (G8's seeded weights: pre-activations of rms ~1)
    print("replayed head units: %d of %d, largest |pre| %.2e" % (n, pattern.numel(), worst))
    ref_loss = mo.gradient_tail_loss(ref_logits, cb.y - 1, 0.2)
    (ref_loss * LOSS_SCALE).backward()
    # (the replay moves the oracle off the golden only by the replayed units' 0.8 |pre| <= 4e-3 each)
    np.testing.assert_allclose(ref_logits.detach().numpy(), z["a/logits"], atol=2e-2)
    # ---- device, benched configuration
    out = m(b)[0].detach().float().cpu()
    r = ref_logits.detach()
    rel = float((out - r).double().norm() / r.double().norm())
    err, top = float((out - r).abs().max()), float(r.abs().max())
    print("logits: relative L2 %.4f  max|err| %.4f  max|ref| %.3f  rms %.3f" % (rel, err, top, float(r.pow(2).mean().sqrt())))
    assert rel <= 2e-2 and err <= 2e-2 * top, (rel, err, top)
    for p in m.parameters():
        p.grad = None
    loss = m.training_step(b, 0)
    print("loss %.7f  oracle %.7f" % (float(loss), float(ref_loss)))
    np.testing.assert_allclose(float(loss), float(ref_loss), rtol=1e-3)
    loss.backward()
    report = []
    for k, p in m.named_parameters():
        if sd[k].grad is None or p.grad is None or k.endswith("linear_k.bias"):
            continue
        check_grad(k, p.grad, sd[k].grad / LOSS_SCALE, report)
    for row in report:
        print("%-48s rms_nz %.3e  relL2 %.4f  q999 %.3f  max %.3f  stray %.1e" % row)
    assert len(report) > 100

    def lim(name):
        if name.startswith("edge_"):
            return 0.12
        if name.split(".")[0] in SMALL_TABLES:
            return 0.08
        return 0.06 if (".linear_q." in name or ".linear_k." in name) else 0.04
    loose = lambda nm: nm.split(".")[0] in SMALL_TABLES or ".linear_q." in nm or ".linear_k." in nm
    bad = [row for row in report if row[2] > lim(row[0]) or row[5] > 1e-3 * row[1]
           or (not loose(row[0]) and (row[3] > 1.5 or row[4] > 6.0))]
    assert not bad, bad


# ------------------------------------------------------------------------------------------------ G10: a training trajectory
@pytest.mark.parametrize("cfg", ["f32", "bf16"])
def test_trainer_walks_the_references_training_trajectory_g10(g8, golden_dir, cfg):
    """Golden G10 (tests/golden/make_golden_traj.py): the REFERENCE trained for 30 AdamW updates (PolynomialDecayLR, warm-up 10,
    peak 1e-3, dropout off) on 480 real Gowalla trajectories in batches of 16, then evaluated on 256 real test trajectories
    (model_fqandtoyo.py:1434-1478, :1546-1616, lr.py:17-31).  `train.TrainStep` -- device collator, hipGraph replay, flat AdamW with
    the device-side schedule -- walks the same trajectory from the same initial state:
      f32 configuration   loss of every update within 1e-4 relative of the reference's, a sample of every
                          parameter after update 30 within 2 % of that parameter's movement, test logits within 2e-3;
      bf16 configuration  (what bench.py times) loss of every update within 3 % relative -- the losses fall from 0.585 to 0.0046
                          over the 30 updates, and a bf16 forward is ~1 % off at any fixed state (test above) -- and parameters
                          within 35 % of their movement (AdamW turns a 1 % gradient error into up to a sign flip of near-zero
                          moment ratios);
    then `metrics.evaluate_outputs` on the 256 test trajectories against the reference's ACC / NDCG @1/5/10/20 (one rank flip
    allowed) and MRR (2 % / 10 %)."""
    from mobgt_amd import metrics
    from mobgt_amd.model_fqandtoyo import Graphormer
    from mobgt_amd.train import TrainStep
    from traj_helpers import param_sample
    z8, uni, table = g8
    z = np.load(os.path.join(golden_dir, "g10_traj.npz"))
    steps, batch, n_test = (int(v) for v in z["args/steps_batch_ntest"])
    warm, tot, peak, end, wd = (float(v) for v in z["args/lr"])
    kw = {} if cfg == "f32" else dict(bias_dtype=torch.bfloat16, gcn_dtype=torch.bfloat16, act_dtype=torch.bfloat16)
    m = Graphormer(n_layers=6, num_heads=8, hidden_dim=128, dropout_rate=0.0, intput_dropout_rate=0.0, weight_decay=wd,
                   ffn_dim=1024, dataset_name="gowalla_nevda", warmup_updates=int(warm), tot_updates=int(tot), peak_lr=peak,
                   end_lr=end, edge_type="multi_hop", multi_hop_max_dist=20, attention_dropout_rate=0.0, universe=uni, **kw)
    names = [str(n) for n in z["param_names"]]
    shapes = [eval(str(s)) for s in z["param_shapes"]]
    sd0 = {k: v.detach() for k, v in seeded_state(list(zip(names, shapes)), int(z["seed"])).items()}
    m.load_state_dict(sd0, strict=True)
    m = m.to(DEV).train()
    m.poi_distance_model.dropout = 0.0                 # (constructor constants of the reference; the golden ran in eval() mode)
    m.poi_cat_model.dropout = 0.0
    m.pos_embed.dropout.p = 0.0
    coll = DeviceCollator(DEV, bin_table=table, multi_hop_max_dist=20, rel_pos_max=1024)
    trajs = real_trajs(z, "train")
    batches = [coll(trajs[s * batch:(s + 1) * batch], idx0=s * batch) for s in range(steps)]
    ts = TrainStep(m, batches, use_graph=True, seed=1)
    ts.prepare()
    worst = 0.0
    for s in range(steps):
        loss = float(ts.step(s))
        rel = abs(loss - float(z["losses"][s])) / float(z["losses"][s])
        worst = max(worst, rel)
        assert rel <= (1e-4 if cfg == "f32" else 3e-2), (s, loss, float(z["losses"][s]))
    print("[%s] largest relative loss deviation over %d updates: %.2e" % (cfg, steps, worst))
    # parameters after update 30, in units of each parameter's own movement (rms of final - initial over the fixture's sample).
    # Excepted: linear_k.bias (zero gradient in exact arithmetic: its AdamW update is round-off over round-off) and the two edge
    # tables -- the reference trains them through its own .half() casts at the PLAIN loss here (no GradScaler in the golden run),
    # which flush part of their per-pair gradients (8 % of edge_encoder's at the first update: make_golden_real.py), while this
    # path keeps them in f32; AdamW normalises the difference into the update.  They are reported, not gated.
    lim = 2e-2 if cfg == "f32" else 0.35           # (measured: 0.8 % on edge_dis_encoder / 27 % on out_degree_encoder)
    rows = []
    for pn, p in m.named_parameters():
        ref, ini = z[f"final/{pn}"], param_sample(sd0[pn].numpy())
        got = param_sample(p.detach().float().cpu().numpy())
        move = float(np.sqrt(((ref - ini) ** 2).mean()))
        dev = float(np.sqrt(((got - ref) ** 2).mean()))
        rows.append((dev / move if move > 0 else (0.0 if dev == 0 else float("inf")), pn, dev, move))
    rows.sort(reverse=True)
    for r in rows[:8]:
        print("[%s] %-52s deviation / movement %.4f  (%.3e / %.3e)" % ((cfg, r[1], r[0]) + r[2:]))
    free = lambda pn: pn.endswith("linear_k.bias") or pn.startswith("edge_")
    bad = [r for r in rows if not free(r[1]) and r[2] > lim * r[3] + 1e-7]
    assert not bad, bad
    m.eval()
    tt = real_trajs(z, "test")
    outs = []
    with torch.no_grad():
        for s in range(n_test // batch):
            b = coll(tt[s * batch:(s + 1) * batch], idx0=s * batch)
            out = m.test_step(b, s)
            if s == 0:
                tol = 2e-3 if cfg == "f32" else 0.15
                np.testing.assert_allclose(out["y_pred"][0].float().cpu().numpy(), z["test/logits0"], rtol=tol, atol=tol)
            outs.append(out)
    r = metrics.evaluate_outputs(outs)
    got = np.array([r["acc@1"], r["acc@5"], r["acc@10"], r["ndcg@1"], r["ndcg@5"], r["ndcg@10"], r["acc@20"], r["ndcg@20"]])
    print("[%s] metrics" % cfg, got, "mrr", r["mrr"], "reference", z["metrics/acc1_5_10_ndcg1_5_10_acc20_ndcg20"], float(z["metrics/mrr"]))
    np.testing.assert_allclose(got, z["metrics/acc1_5_10_ndcg1_5_10_acc20_ndcg20"], atol=1.0 / n_test + 1e-12)
    np.testing.assert_allclose(r["mrr"], float(z["metrics/mrr"]), rtol=2e-2 if cfg == "f32" else 1e-1)

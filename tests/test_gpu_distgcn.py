"""The distance GCN of the fq model in its three-launch form (round 4; `modelGNN._DistGcnFn`, csrc/maskgemm.hip "Round 4").

Reference: graphormer/modelGNN.py:38-44 (GraphConvolution: adj @ (x @ W) + b), :66-72 (GCN.forward: LeakyReLU(0.2) after every
hidden layer, dropout in front of the last layer); model_fqandtoyo.py:481-486 (adj = (D+I)^-1 (A+I) for a 0/1 matrix A),
:1236 / :1264 (the table is read at the batch's POI rows).

  * against a float64 restatement on the host (dense normalised adjacency, the dropout mask replayed through
    `mobgt_dropout_mask_host`): output and all six parameter gradients; operands of the two P-wide products are bf16 in the
    kernels (as in the launch-per-product path), so the gates are relative L2 3e-3 forward, 1e-2 on gradients;
  * against the launch-per-product path (`MOBGT_NO_DIST_GCN_FUSED=1`) with the same masks;
  * a NON-symmetric adjacency (the transposed products must use mask_t), P and R that are not multiples of the tile sizes,
    repeated rows, dropout on and off.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mobgt_amd import ops                                                 # noqa: E402
from mobgt_amd.modelGNN import GCN, MaskAdj                                # noqa: E402

DEV = "cuda"


def _setup(P, R, NO, seed, density=0.03):
    rng = np.random.RandomState(seed)
    a = (rng.rand(P, P) < density).astype(np.float64)
    np.fill_diagonal(a, 0.0)
    x = rng.randn(P, 303) * 0.5
    mask, mask_t, scale = MaskAdj.from_dense01(a)
    ahat = (a + np.eye(P)) / (a.sum(1, keepdims=True) + 1.0)
    ax = ahat @ x
    ax_pad = np.zeros((P, 304), dtype=np.float32)
    ax_pad[:, :303] = ax
    rows = rng.randint(0, P, size=R).astype(np.int64)
    rows[: R // 4] = rows[R // 4: 2 * (R // 4)]                          # repeated rows
    torch.manual_seed(seed)
    gcn = GCN(303, [16, 64], NO, dropout=0.3).to(DEV)
    return dict(ahat=ahat, ax=ax, ax_pad=torch.from_numpy(ax_pad).to(DEV), x=torch.from_numpy(x.astype(np.float32)).to(DEV),
                adj=MaskAdj(mask.to(DEV), mask_t.to(DEV), scale.to(DEV)), dense=torch.from_numpy(ahat).to(DEV).to(torch.bfloat16),
                rows=torch.from_numpy(rows).to(DEV), rows_np=rows, gcn=gcn)


def _run(s, train, gout):
    gcn = s["gcn"]
    gcn.train(train)
    for p in gcn.parameters():
        p.grad = None
    out = gcn(s["x"], s["dense"], s["ax_pad"], rows=s["rows"], mask_adj=s["adj"])
    out.backward(gout)
    torch.cuda.synchronize()
    return out.detach().double().cpu(), [p.grad.detach().double().cpu().clone() for p in gcn.parameters()]


def _reference(s, train, gout, seed_total, NO):
    P = s["ax"].shape[0]
    ps = [p.detach().double().cpu().requires_grad_(True) for p in s["gcn"].parameters()]
    w0, b0, w1, b1, w2, b2 = ps
    ahat = torch.from_numpy(s["ahat"])
    lr = torch.nn.functional.leaky_relu
    y0 = lr(torch.from_numpy(s["ax"]) @ w0 + b0, 0.2)
    y1 = lr(ahat @ y0 @ w1 + b1, 0.2)
    if train:
        thr = int(0.3 * 65536.0 + 0.5)
        keep = torch.from_numpy(ops.dropout_site_mask(seed_total, 0x2000 + NO, P, 64, 0.3))
        y1 = y1 * keep.double() / (1.0 - thr / 65536.0)
    out = ahat[torch.from_numpy(s["rows_np"])] @ y1 @ w2 + b2
    out.backward(gout.double().cpu())
    return out.detach(), [p.grad for p in ps]


def _rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("P,R,NO,train", [(1003, 203, 128, True), (1003, 203, 128, False), (2500, 608, 128, True), (777, 48, 192, True)])
def test_three_launch_distance_gcn_vs_float64_and_vs_the_launch_per_product_path(P, R, NO, train, monkeypatch):
    s = _setup(P, R, NO, seed=P + R)
    gout = torch.randn(R, NO, device=DEV)
    ops.set_dropout_state(torch.tensor([5], dtype=torch.int64, device=DEV), 21)
    try:
        monkeypatch.delenv("MOBGT_NO_DIST_GCN_FUSED", raising=False)
        out_f, g_f = _run(s, train, gout)
        out_f2, _ = _run(s, train, gout)
        assert torch.equal(out_f, out_f2), "the forward pass is bit-reproducible (two-term atomic sums only)"
        monkeypatch.setenv("MOBGT_NO_DIST_GCN_FUSED", "1")
        out_s, g_s = _run(s, train, gout)
    finally:
        ops.set_dropout_state(None, None)
    out_r, g_r = _reference(s, train, gout, 21 + 5, NO)
    names = ["w0", "b0", "w1", "b1", "w2", "b2"]
    print("fused vs f64: out %.2e | %s" % (_rel(out_f, out_r), " ".join("%s %.2e" % (n, _rel(a, b)) for n, a, b in zip(names, g_f, g_r))))
    print("split vs f64: out %.2e | %s" % (_rel(out_s, out_r), " ".join("%s %.2e" % (n, _rel(a, b)) for n, a, b in zip(names, g_s, g_r))))
    assert _rel(out_f, out_r) < 3e-3
    for n, a, b in zip(names, g_f, g_r):
        assert _rel(a, b) < 1e-2, n
    assert _rel(out_f, out_s) < 5e-3
    for n, a, b in zip(names, g_f, g_s):
        assert _rel(a, b) < 1.5e-2, n


def test_full_table_path_vs_float64():
    """rows = None (batches with more positions than the rows form takes: the S-GOW tail): the last layer over all P rows."""
    P, NO = 1003, 128
    s = _setup(P, 64, NO, seed=5)
    gcn = s["gcn"]
    gcn.eval()
    for p in gcn.parameters():
        p.grad = None
    gout = torch.randn(P, NO, device=DEV)
    out = gcn(s["x"], s["dense"], s["ax_pad"], rows=None, adj_t=s["dense"].t().contiguous(), mask_adj=s["adj"])
    out.backward(gout)
    ps = [p.detach().double().cpu().requires_grad_(True) for p in gcn.parameters()]
    w0, b0, w1, b1, w2, b2 = ps
    ahat = torch.from_numpy(s["ahat"])
    lr = torch.nn.functional.leaky_relu
    ref = ahat @ lr(ahat @ lr(torch.from_numpy(s["ax"]) @ w0 + b0, 0.2) @ w1 + b1, 0.2) @ w2 + b2
    ref.backward(gout.double().cpu())
    assert _rel(out.detach().double().cpu(), ref.detach()) < 5e-3
    for n, p, q in zip(["w0", "b0", "w1", "b1", "w2", "b2"], gcn.parameters(), ps):
        assert _rel(p.grad.double().cpu(), q.grad) < 1.5e-2, n


def test_three_launch_form_is_what_the_fq_model_runs():
    """The benched S-FSQ model takes the new path (no row gather of the dense adjacency in its forward)."""
    from mobgt_amd import workloads
    from mobgt_amd import modelGNN
    uni, model, coll = workloads.build("fsq", DEV, seed=1, P=1500, model_overrides=dict(n_layers=1))
    calls = []
    orig = modelGNN._dist_gcn_ok
    modelGNN._dist_gcn_ok = lambda *a: (calls.append(orig(*a)), calls[-1])[1]
    try:
        from mobgt_amd import synth
        batch = coll(synth.make_batch_of_trajectories(seed=21, G=8, P=1500, n_user=1080, cat_of_poi=uni.cat_of_poi,
                                                      n_nodes=[17, 3, 9, 2, 11, 5, 40, 23]))
        model.train()
        model.training_step(batch, 0).backward()
        torch.cuda.synchronize()
    finally:
        modelGNN._dist_gcn_ok = orig
    assert calls and all(calls), "the distance GCN did not take the three-launch form"

"""Larger graphs (BASELINE configs[2]/[4] shapes): the whole device pipeline -- batched SPD / edge paths,
bias assembly, both attention passes, fused layers -- against the oracle at N = 300, and size-independent
properties at the Gowalla maximum N = 814 (T = 815) and the c5 shape (T = 785, C = 256, d = 32)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mobgt_amd import ops, synth                                  # noqa: E402
from mobgt_amd.data import DeviceCollator, make_bin_table          # noqa: E402
from oracle import model_oracle as mo                               # noqa: E402

DEV = "cuda"
ARGS = dict(n_layers=2, num_heads=8, hidden_dim=128, dropout_rate=0.1, intput_dropout_rate=0.1, weight_decay=0.01,
            ffn_dim=256, warmup_updates=10, tot_updates=100, peak_lr=2e-4, end_lr=1e-9, edge_type="multi_hop",
            multi_hop_max_dist=20, attention_dropout_rate=0.1)


def _setup(n_nodes, P=600, seed=3):
    from mobgt_amd.model_fqandtoyo import Graphormer
    uni = synth.make_universe(P=P, n_cat=12, n_user=1080, seed=seed)
    nb, _, table = make_bin_table(uni.distance)
    torch.manual_seed(seed)
    model = Graphormer(dataset_name="gowalla_nevda", universe=uni, num_bins=nb + 2, **ARGS).to(DEV)
    coll = DeviceCollator(DEV, bin_table=table)
    trajs = synth.make_batch_of_trajectories(seed=seed + 1, G=len(n_nodes), P=P, n_user=1080, cat_of_poi=uni.cat_of_poi,
                                             n_nodes=n_nodes)
    return uni, model, coll(trajs)


def _cpu_batch(b):
    c = SimpleNamespace()
    for f in ("attn_bias", "rel_pos", "poi_pos", "edge_input", "x", "in_degree", "out_degree", "user", "y", "time_normal"):
        t = getattr(b, f).cpu()
        setattr(c, f, t.float() if t.dtype.is_floating_point else t.long())
    return c


def test_fq_model_matches_oracle_at_n300():
    uni, model, batch = _setup([300, 120, 7])
    model.eval()
    with torch.no_grad():
        logits = model(batch)[0].cpu()
    consts = mo.fq_constants(uni, "gowalla_nevda")
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref, _ = mo.graphormer_fq_forward(sd, _cpu_batch(batch), consts, n_layers=2, H=8, D=20)
    np.testing.assert_allclose(logits.numpy(), ref.numpy(), rtol=3e-2, atol=3e-2)
    # integer path once more at this size: SPD / edge features from the device pipeline vs the C oracle
    from oracle import algos_oracle as ao
    c = batch._counts[0, :300, :300].cpu().numpy().astype(np.int64)
    M, path = ao.floyd_warshall(c != 0)
    assert np.array_equal(batch.rel_pos[0, :300, :300].cpu().numpy().astype(np.int64), M + 1)


def test_train_step_at_gowalla_max_n814_is_finite_and_masks_padding():
    uni, model, batch = _setup([814, 40], P=1000)
    model.train()
    ops.set_dropout_state(torch.zeros(1, dtype=torch.int64, device=DEV), 7)
    for m in model.modules():
        if hasattr(m, "seed_dev"):
            m.seed_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    loss = model.training_step(batch, 0)
    loss.backward()
    assert torch.isfinite(loss)
    for n, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
    # padded keys of the short graph get exactly zero probability: its logits do not change when the
    # long graph's nodes change (independent samples)
    model.eval()
    with torch.no_grad():
        a = model(batch)[0][1].clone()
        batch.x[0, :800, 0] = torch.roll(batch.x[0, :800, 0], 1)
        b = model(batch)[0][1]
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=1e-5)


def test_attention_backward_properties_at_c5_shape():
    """T = 785, d = 32, C = 256: dV is linear in dO, rows of dBias sum to ~0 (softmax Jacobian), and the
    transposed-bias pass agrees with the row-major pass on dK through the identity sum_k dK = 0-shift test."""
    G, H, T, d = 1, 8, 785, 32
    C = H * d
    g = torch.Generator().manual_seed(1)
    q, k, v, do1, do2 = (torch.randn(G, T, C, generator=g).to(DEV) for _ in range(5))
    bias = torch.randn(G, H, T, T, generator=g).to(DEV).requires_grad_(True)
    pack = ops.pack_bias(bias, G, H, T, dtype=torch.bfloat16)

    def grads(do):
        qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
        pack.n_bwd = 0
        o = ops.attention(qq, kk, vv, pack, d ** -0.5)
        o.backward(do)
        return qq.grad, kk.grad, vv.grad, pack.grad_total().clone()
    dq1, dk1, dv1, db1 = grads(do1)
    dq2, dk2, dv2, db2 = grads(do2)
    dq3, dk3, dv3, db3 = grads(do1 + do2)
    for a, b, c in ((dq1, dq2, dq3), (dk1, dk2, dk3), (dv1, dv2, dv3), (db1, db2, db3)):
        scale = float(c.abs().max())
        np.testing.assert_allclose((a + b).cpu().numpy(), c.cpu().numpy(), atol=2e-2 * scale)
    # dS rows sum to zero: sum_j P_ij (dP_ij - delta_i) = 0
    rs = db3.sum(-1)
    assert float(rs.abs().max()) < 2e-2 * float(db3.abs().max()) * 8


def _rel_l2(got, want):
    got, want = got.double(), want.double()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


def test_gowalla_814_node_graph_bias_and_six_layer_stack_vs_oracle():
    """BASELINE configs[2]'s tail (collator.py:460-608 pads the batch to its longest trajectory: Gowalla's maximum is 814 nodes,
    T = 815) on the TIMED S-GOW model (`workloads.build("gow")`: P = 3 679, C = 192, d = 24, 6 layers, ffn 1024, bf16): the
    assembled bias of the 814-node graph and of a 40-node graph padded to 814 against `oracle.assemble_bias`
    (model_fqandtoyo.py:1143-1216; split Floyd-Warshall, long-bucket bias assembly, finite SPDs far beyond the 20 hops), and
    the six encoder layers on that batch's real token rows and bias against `oracle.encoder_layer_fq`
    (model_fqandtoyo.py:1731-1743), layer by layer -- what tests/test_gpu_c5.py does for S-BIG (VERDICT r3 missing #5b).
    Tolerances as there: bias half a bf16 ulp of its magnitude; stack relative L2 <= 1e-2, max |err| <= 0.06."""
    from mobgt_amd import workloads
    from mobgt_amd.model import refresh_shadows
    uni, model, coll = workloads.build("gow", DEV, seed=1)
    trajs = synth.make_batch_of_trajectories(seed=77, G=2, P=uni.P, n_user=1080,
                                             cat_of_poi=uni.cat_of_poi, n_nodes=[814, 40])
    batch = coll(trajs)
    assert tuple(batch.x.shape[:2]) == (2, 814) and len(model.layers) == 6
    model.eval()
    with torch.no_grad():
        pack = model.assemble_bias(batch)
        T = 815
        got = pack.bias[:, :, :, :T].float().cpu()
        sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()
              if k.split(".")[0] in ("edge_encoder", "edge_dis_encoder", "rel_pos_encoder", "poi_pos_encoder", "graph_token_virtual_distance")}
        ref = mo.assemble_bias(sd, _cpu_batch(batch), 8, 20, "fq")
    assert torch.equal(torch.isinf(got), torch.isinf(ref))
    assert bool(torch.isinf(ref[1, :, :, 41:]).all()) and not bool(torch.isinf(ref[0]).any())       # padding of the short graph
    fin = torch.isfinite(ref)
    err = float((got[fin] - ref[fin]).abs().max())
    print("bias max|err| %.5f  max|ref| %.4f" % (err, float(ref[fin].abs().max())))
    assert err <= 2 ** -8 * max(1.0, float(ref[fin].abs().max()))
    assert torch.equal(pack.bias_t[:, :, :, :T].float().cpu(), got.transpose(2, 3))
    # SPDs of the long graph reach well beyond multi_hop_max_dist (the clamp of model_fqandtoyo.py:1168-1173 is on the path)
    assert int(batch.rel_pos[0].max()) > 40
    with torch.no_grad():
        refresh_shadows(model.layers)
        x0 = model.node_features(batch)
        out, outs = x0, []
        for li, layer in enumerate(model.layers):
            out = layer(out, pack, mask=None, next_layer=model.layers[li + 1] if li + 1 < len(model.layers) else None)
            outs.append(out.float().cpu())
        sdl = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if k.startswith("layers.")}
        r = x0.float().cpu()
        for li in range(len(model.layers)):
            r = mo.encoder_layer_fq(sdl, f"layers.{li}", r, ref, 8)
            # (rows of padded positions of the short graph attend over its 41 real keys like any other row: compared too)
            rel, mx = _rel_l2(outs[li], r), float((outs[li] - r).abs().max())
            print("layer %d  relL2 %.5f  max|err| %.4f" % (li, rel, mx))
            assert rel <= 1e-2 and mx <= 0.06, (li, rel, mx)

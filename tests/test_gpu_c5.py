"""BASELINE configs[4] (c5, the HBM-roofline stress shape) AT FULL SIZE against the oracle (VERDICT r2 "missing" #3):

  * the bias-fused attention forward and all four gradients (dQ, dK, dV, dBias) at T = 785, d = 32, H = 8, bf16 I/O and
    bf16 bias -- the instantiation the S-BIG step runs -- ELEMENTWISE against the fp32 restatement of
    `graphormer/model.py:436-455`, with -inf key columns and with the training-mode dropout mask replayed on the host;
  * one full-size S-BIG batch (`workloads.build("big")`: P = 100 000 as CSR, 16 graphs x 784 nodes, C = 256, 12 layers):
    a training step with finite loss and gradients, the assembled bias of two of its graphs against
    `oracle.assemble_bias` (model_fqandtoyo.py:1143-1216), and the 12-layer encoder stack on that batch's real bias and
    token rows against `oracle.encoder_layer_fq` (model_fqandtoyo.py:1731-1743).

Tolerances: attention operands and the bias are bf16 (fp32 accumulate / softmax); the reference is fp32 on the same
bf16-rounded inputs.  Outputs: |err| <= 8e-3 (values O(1); measured 2e-3); gradients: |err| <= 1.5e-2 * max|ref| and
relative L2 <= 1e-2 (measured: max 1.0e-2 * max|ref| on dK, relative L2 0.17-0.48 %); encoder stack (bf16 GEMM operands, 12
layers, post-LN so rows have rms 1): relative L2 <= 1e-2, max |err| <= 0.06 (measured: 0.10 % after layer 0 rising to 0.38 %
after layer 11, max |err| 0.02).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mobgt_amd import ops, workloads                         # noqa: E402
from oracle import model_oracle as mo                         # noqa: E402
from test_gpu_kernels import bf16r, make_bias, ref_attention  # noqa: E402

DEV = "cuda"


def _rel_l2(got, want):
    got, want = got.double(), want.double()
    return float((got - want).norm() / want.norm().clamp_min(1e-30))


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_attention_all_gradients_at_t785_d32_elementwise(p_drop):
    G, H, T, d = 2, 8, 785, 32
    C = H * d
    rng = np.random.RandomState(785)
    q, k, v, gy = (bf16r(torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))) for _ in range(4))
    bias = bf16r(make_bias(rng, G, H, T, [T, 700]))
    scale = d ** -0.5
    seed = 0x5DEECE66D1234567
    keep, inv_keep = None, 1.0
    if p_drop:
        keep = torch.from_numpy(ops.dropout_keep_mask(seed, G, H, T, p_drop)).float()
        inv_keep = 1.0 / (1.0 - int(p_drop * 65536 + 0.5) / 65536.0)
        assert abs(1.0 - float(keep.mean()) - p_drop) < 5e-3
    qr, kr, vr, br = (t.clone().requires_grad_(True) for t in (q, k, v, bias))
    ref = ref_attention(qr, kr, vr, br, H, scale, keep=keep, inv_keep=inv_keep)
    ref.backward(gy)
    qd, kd, vd = (t.to(DEV).to(torch.bfloat16).requires_grad_(True) for t in (q, k, v))
    bd = bias.to(DEV).requires_grad_(True)
    pack = ops.pack_bias(bd, G, H, T, dtype=torch.bfloat16)
    seed_dev = torch.tensor([3], dtype=torch.int64, device=DEV)
    out = ops.attention(qd, kd, vd, pack, scale, p_drop=p_drop, seed=seed - 3, seed_dev=seed_dev)
    assert out.dtype == torch.bfloat16
    out.backward(gy.to(DEV).to(torch.bfloat16))
    torch.cuda.synchronize()
    o = out.detach().float().cpu()
    print("out   max|err| %.4f  relL2 %.5f" % (float((o - ref.detach()).abs().max()), _rel_l2(o, ref.detach())))
    np.testing.assert_allclose(o.numpy(), ref.detach().numpy(), atol=8e-3, rtol=8e-3)
    for name, got, want in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad), ("dbias", bd.grad, br.grad)):
        g, w = got.float().cpu(), want
        mx = float(w.abs().max())
        err, rel = float((g - w).abs().max()), _rel_l2(g, w)
        print("%-5s max|err| %.4f (max|ref| %.3f)  relL2 %.5f" % (name, err, mx, rel))
        assert err <= 1.5e-2 * max(1.0, mx), (name, err, mx)
        assert rel <= 1e-2, (name, rel)
    # padded key columns of graph 1 (-inf bias): exactly zero gradient for those keys and their bias entries
    assert float(kd.grad[1, 700:].abs().max()) == 0.0 and float(vd.grad[1, 700:].abs().max()) == 0.0
    assert float(bd.grad[1, :, :, 700:].abs().max()) == 0.0


def test_one_pass_backward_leaves_its_dq_accumulator_zero_and_repeats(monkeypatch):
    """csrc/attn.hip, mobgt_attn_bias_bwd_fused_z: the f32 dQ accumulator is ONE persistent buffer per (device, stream, size)
    that a call finds zero and leaves zero (the finishing launch re-zeroes it), rowsum(dO O) is formed inside the pass.  Two
    backward passes in a row (dropout on) give the same dK / dV / dBias bit for bit and the same dQ up to the order of its f32
    atomics; the buffer is all zeros afterwards; a backward on ANOTHER stream gets its own accumulator."""
    G, H, T, d = 2, 8, 300, 32
    C = H * d
    rng = np.random.RandomState(11)
    q, k, v, gy = (bf16r(torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))).to(DEV).to(torch.bfloat16) for _ in range(4))
    bias = bf16r(make_bias(rng, G, H, T, [T, 250])).to(DEV)
    seed_dev = torch.tensor([3], dtype=torch.int64, device=DEV)

    def run():
        qd, kd, vd = (x.clone().requires_grad_(True) for x in (q, k, v))
        bd = bias.clone().requires_grad_(True)
        pack = ops.pack_bias(bd, G, H, T, dtype=torch.bfloat16)
        out = ops.attention(qd, kd, vd, pack, d ** -0.5, p_drop=0.1, seed=77, seed_dev=seed_dev)
        out.backward(gy)
        torch.cuda.synchronize()
        return [x.grad.float().clone() for x in (qd, kd, vd, bd)]
    a = run()
    key = (str(q.device), int(torch.cuda.current_stream().cuda_stream), G * T * C)
    assert key in ops._DQ_ACC and not ops._DQ_ACC[key]["busy"]
    assert float(ops._DQ_ACC[key]["buf"].abs().max()) == 0.0
    b = run()
    assert float(ops._DQ_ACC[key]["buf"].abs().max()) == 0.0
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    assert _rel_l2(b[0], a[0]) < 1e-3
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c = run()
    torch.cuda.current_stream().wait_stream(side)
    key2 = (str(q.device), int(side.cuda_stream), G * T * C)
    assert key2 != key and key2 in ops._DQ_ACC and ops._DQ_ACC[key2]["buf"].data_ptr() != ops._DQ_ACC[key]["buf"].data_ptr()
    assert float(ops._DQ_ACC[key2]["buf"].abs().max()) == 0.0
    assert torch.equal(a[1], c[1]) and torch.equal(a[2], c[2]) and _rel_l2(c[0], a[0]) < 1e-3


# --------------------------------------------------------------------------------------- full-size S-BIG
def _cpu_batch(b, sl=slice(None)):
    c = SimpleNamespace()
    for f in ("attn_bias", "rel_pos", "poi_pos", "edge_input", "x", "in_degree", "out_degree", "user", "y", "time_normal"):
        t = getattr(b, f)[sl].cpu()
        setattr(c, f, t.float() if t.dtype.is_floating_point else t.long())
    return c


@pytest.fixture(scope="module")
def big():
    uni, model, coll = workloads.build("big", DEV, seed=1)
    trajs = workloads.make_pool("big", 1, 16, uni)[0]
    batch = coll(trajs)
    return uni, model, batch


def test_s_big_full_size_training_step_is_finite(big):
    uni, model, batch = big
    assert model.X.shape[0] == 100000 and tuple(batch.x.shape[:2]) == (16, 784) and len(model.layers) == 12
    assert model.layers[0].self_attention.att_size == 32
    model.train()
    sd_ = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.set_dropout_state(sd_, 7)
    for m in model.modules():
        if hasattr(m, "seed_dev"):
            m.seed_dev = sd_
    for p in model.parameters():
        p.grad = None
    loss = model.training_step(batch, 0)
    loss.backward()
    assert torch.isfinite(loss), float(loss)
    n = 0
    for name, p in model.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
            n += 1
    assert n > 100
    for p in model.parameters():
        p.grad = None
    ops.set_dropout_state(None, 0)


def test_s_big_bias_of_two_graphs_vs_oracle(big):
    uni, model, batch = big
    model.eval()
    with torch.no_grad():
        pack = model.assemble_bias(batch)
    T = batch.x.shape[1] + 1
    got = pack.bias[:2, :, :, :T].float().cpu()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()
          if k.split(".")[0] in ("edge_encoder", "edge_dis_encoder", "rel_pos_encoder", "poi_pos_encoder", "graph_token_virtual_distance")}
    with torch.no_grad():
        ref = mo.assemble_bias(sd, _cpu_batch(batch, slice(0, 2)), 8, 20, "fq")
    assert torch.equal(torch.isinf(got), torch.isinf(ref))
    fin = torch.isfinite(ref)
    err = float((got[fin] - ref[fin]).abs().max())
    print("bias max|err| %.5f  max|ref| %.4f" % (err, float(ref[fin].abs().max())))
    # bf16 output: half an ulp of values up to ~0.5 in magnitude (tables are N(0,1)-initialised rows summed: |b| < 8)
    assert err <= 2 ** -8 * max(1.0, float(ref[fin].abs().max()))
    # the transposed copy the dK/dV pass reads
    assert torch.equal(pack.bias_t[:2, :, :, :T].float().cpu(), got.transpose(2, 3))


def test_s_big_encoder_stack_vs_oracle_layers(big):
    uni, model, batch = big
    model.eval()
    from mobgt_amd.model import refresh_shadows
    with torch.no_grad():
        pack = model.assemble_bias(batch)
        refresh_shadows(model.layers)
        x0 = model.node_features(batch)
        out = x0
        outs = []
        for li, layer in enumerate(model.layers):
            out = layer(out, pack, mask=None, next_layer=model.layers[li + 1] if li + 1 < len(model.layers) else None)
            outs.append(out.float().cpu())
    G, T, C = x0.shape
    assert (G, T, C) == (16, 785, 256)
    bias = pack.bias[..., :T].float().cpu()
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if k.startswith("layers.")}
    ref = x0.float().cpu()
    with torch.no_grad():
        for li in range(len(model.layers)):
            ref = mo.encoder_layer_fq(sd, f"layers.{li}", ref, bias, 8)
            rel, mx = _rel_l2(outs[li], ref), float((outs[li] - ref).abs().max())
            print("layer %2d  relL2 %.5f  max|err| %.4f  rms %.3f" % (li, rel, mx, float(ref.pow(2).mean().sqrt())))
            assert rel <= 1e-2 and mx <= 0.06, (li, rel, mx)

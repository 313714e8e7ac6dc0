"""EncoderLayer parity on the GPU against the reference's own outputs (golden G4: y, dx, dbias and every
parameter gradient for a fixed upstream gradient), for both layer variants, the fused single-node path and
the op-by-op path, fp32 and bf16 GEMM-facing activations.

Tolerance: fp32 activations run fp32 end to end since round 5 (the attention's full-f32 instantiation,
csrc/attn_f32_body.h) -> 1e-4 on outputs / input gradients relative to their scale (measured 4e-7 .. 1.5e-6), parameter
gradients within 0.1 % relative L2; bf16 activations (bf16 GEMM and MFMA operands, the benched configuration) -> 4e-2 / 5 %.
d(linear_k.bias) is exactly 0 in exact arithmetic and is compared with an absolute tolerance.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gradcheck import grad_errors, grad_sample                      # noqa: E402

from inputs import ENCODER_CASES, encoder_case              # noqa: E402
from test_oracle_model import encoder_param_list, seeded_state   # noqa: E402

DEV = "cuda"


def build_layer(variant, C, ffn, seed):
    if variant == "stock":
        from mobgt_amd.model import EncoderLayer
    else:
        from mobgt_amd.model_fqandtoyo import EncoderLayer
    layer = EncoderLayer(C, ffn, 0.1, 0.1, 8)
    sd = {k[2:]: v.detach() for k, v in seeded_state([("L." + n, s) for n, s in encoder_param_list(variant, C, ffn)], seed).items()}
    layer.load_state_dict(sd, strict=True)
    return layer.to(DEV).eval()


@pytest.mark.parametrize("variant", ["stock", "fq"])
@pytest.mark.parametrize("case", ENCODER_CASES, ids=[c[0] for c in ENCODER_CASES])
@pytest.mark.parametrize("mode", ["fused_f32", "fused_bf16", "unfused"])
def test_encoder_layer_matches_reference_g4(golden_dir, variant, case, mode):
    z = np.load(os.path.join(golden_dir, "g4_encoder.npz"))
    cname, C, T, G, ffn = case
    name = f"{variant}/{cname}"
    seed, x, bias, gy, _ = encoder_case(variant, C, T, G)
    layer = build_layer(variant, C, ffn, seed + 1)
    layer.fused = mode != "unfused"
    layer.act_dtype = torch.bfloat16 if mode == "fused_bf16" else torch.float32
    xd = torch.from_numpy(x).to(DEV).requires_grad_(True)
    bd = torch.from_numpy(bias).to(DEV).requires_grad_(True)
    y = layer(xd, bd, mask=None)
    y.backward(torch.from_numpy(gy).to(DEV))
    torch.cuda.synchronize()
    # f32 activations: the attention runs its full-f32 instantiation and every GEMM is f32 -> 1e-4 relative to the scale of
    # the reference (measured 4e-7 .. 1.5e-6); bf16 activations: bf16 GEMM and MFMA operands -> 4e-2
    tol = 4e-2 if mode == "fused_bf16" else 1e-4
    ref_y, ref_dx = z[f"{name}/y"], z[f"{name}/dx"]
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref_y, atol=tol * max(1.0, np.abs(ref_y).max()), rtol=tol)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), ref_dx, atol=tol * max(1.0, np.abs(ref_dx).max()), rtol=tol)
    db = bd.grad.cpu().numpy()
    db = db if T <= 40 else db[:, :, ::7, :]
    ref_db = z[f"{name}/dbias"]
    np.testing.assert_allclose(db, ref_db, atol=tol * max(1.0, np.abs(ref_db).max()), rtol=tol)
    bad = []
    gtol = 5e-2 if mode == "fused_bf16" else 1e-3          # (f32 configurations: parameter gradients elementwise within 0.1 % relative L2)
    for pn, p in layer.named_parameters():
        if f"{name}/grad_none/{pn}" in z:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pn
            continue
        rs, rn = z[f"{name}/gstat/{pn}"]
        gn = p.grad.double().norm().item()
        if pn.endswith("linear_k.bias"):
            # exactly 0 in exact arithmetic (softmax is invariant to a per-query shift of all keys); the reference
            # leaves fp32 round-off, bf16 MFMA operands leave ~1e-2 of the neighbouring bias gradients
            vb = z[f"{name}/gstat/{pn.replace('linear_k', 'linear_v')}"][1]
            assert gn <= 2e-2 * vb + 1e-3, (pn, gn, vb)
            continue
        if not np.isclose(gn, rn, rtol=gtol, atol=2e-2):
            bad.append((pn, gn, float(rn)))
        # ELEMENTWISE against the reference's gradient (golden G4 keeps all of it, or every 7th row of a large matrix),
        # tolerance scaled by the RMS of the reference: relative L2 <= gtol, elements within 0.15 rms + 5 %
        ref = z[f"{name}/grad/{pn}"]
        rms, rel_l2, q999, mx, stray = grad_errors(grad_sample(p.grad.cpu().numpy()), ref)
        if rel_l2 > gtol or q999 > 1.0 or mx > 4.0:
            bad.append((pn, "elementwise", rel_l2, q999, mx))
    assert not bad, bad


def test_fused_layer_dropout_statistics():
    """Training mode: dropout masks differ per site and per seed, keep rate ~ 1-p, and the layer stays unbiased
    (mean over many seeds approaches the eval output for a linear probe)."""
    from mobgt_amd.model_fqandtoyo import EncoderLayer
    torch.manual_seed(0)
    C, T, G = 192, 12, 4
    layer = EncoderLayer(C, 256, 0.1, 0.1, 8).to(DEV)
    x = torch.randn(G, T, C, device=DEV)
    bias = torch.zeros(G, 8, T, T, device=DEV)
    layer.eval()
    ref = layer(x, bias)
    layer.train()
    seed_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    layer.self_attention.seed_dev = seed_dev
    outs = []
    for i in range(64):
        seed_dev.fill_(i)
        outs.append(layer(x, bias))
    outs = torch.stack(outs)
    assert not torch.equal(outs[0], outs[1])
    seed_dev.fill_(0)
    assert torch.equal(layer(x, bias), outs[0])               # same seed -> same masks
    err = (outs.mean(0) - ref).abs().mean().item()
    spread = (outs[0] - ref).abs().mean().item()
    assert err < 0.35 * spread


@pytest.mark.parametrize("R,M,N", [(2432, 192, 192), (2432, 576, 192), (37, 6, 10), (1, 64, 64), (12560, 256, 1024),
                                   (100, 130, 66)])
def test_linear_wgrad_matches_fp32_reference(R, M, N):
    """dW = g^T x and db = g.sum(0) of the split-K MFMA kernel (csrc/wgrad.hip) against torch fp64 on the same
    bf16-rounded operands: only fp32 accumulation-order error remains (atol 1e-3 * sqrt(R) * |g||x| scale)."""
    from mobgt_amd import ops
    gen = torch.Generator().manual_seed(R + M + N)
    g = torch.randn(R, M, generator=gen).to(DEV).bfloat16()
    x = torch.randn(R, N, generator=gen).to(DEV).bfloat16()
    dw, db = ops.linear_wgrad(g, x, with_bias=True)
    ref = g.double().t() @ x.double()
    tol = 2e-5 * (R ** 0.5) * 4
    np.testing.assert_allclose(dw.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=tol)
    np.testing.assert_allclose(db.cpu().numpy(), g.double().sum(0).cpu().numpy(), rtol=1e-4, atol=tol)
    # column slices of a wider buffer (the fused QKV layout) and accumulation into an existing gradient
    # f32 operands: rounded to bf16 while loading -- identical to casting first
    g32 = torch.randn(R, M, generator=gen).to(DEV)
    x32 = torch.randn(R, N, generator=gen).to(DEV)
    dw32, db32 = ops.linear_wgrad(g32, x32, with_bias=True)
    gb, xb = g32.bfloat16(), x32.bfloat16()
    np.testing.assert_allclose(dw32.cpu().numpy(), (gb.double().t() @ xb.double()).cpu().numpy(), rtol=1e-4, atol=tol)
    np.testing.assert_allclose(db32.cpu().numpy(), gb.double().sum(0).cpu().numpy(), rtol=1e-4, atol=tol)
    wide = torch.randn(R, 2 * M + 2, generator=gen).to(DEV).bfloat16()
    gs = wide[:, 2:2 + M]
    dw2, _ = ops.linear_wgrad(gs, x)
    np.testing.assert_allclose(dw2.cpu().numpy(), (gs.double().t() @ x.double()).cpu().numpy(), rtol=1e-4, atol=tol)


def test_adamw_flat_matches_torch_adamw():
    """mobgt_adamw_flat (one pass over the flat buffers, device lr / step counter, bf16 shadow) against
    torch.optim.AdamW for five steps with changing learning rates."""
    from mobgt_amd import _lib
    from mobgt_amd.ops import _p, _stream
    gen = torch.Generator().manual_seed(3)
    n = 10007
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=0.01)
    p = p0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sh = torch.zeros(n, dtype=torch.bfloat16, device=DEV)
    lr_dev = torch.zeros((), dtype=torch.float32, device=DEV)
    step_dev = torch.tensor([40], dtype=torch.int64, device=DEV)
    for it in range(5):
        g = torch.randn(n, generator=gen) * (0.1 + it)
        lr = 1e-3 * (it + 1)
        for grp in opt.param_groups:
            grp["lr"] = lr
        ref.grad = g.double()
        opt.step()
        lr_dev.fill_(lr)
        step_dev.add_(1)
        _lib.check(_lib.lib().mobgt_adamw_flat(_p(p), _p(g.to(DEV)), _p(m), _p(v), _p(sh), n, _p(lr_dev), None, _p(step_dev), 40,
                                               0.9, 0.999, 1e-8, 0.01, _stream()), "mobgt_adamw_flat")
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().float().numpy(), rtol=2e-5, atol=2e-6)
    assert torch.equal(sh, p.bfloat16())


def test_adamw_flat_device_schedule_matches_polynomial_decay():
    """The in-kernel learning-rate schedule (sched = warmup, total, peak, end, offset) against torch.optim.AdamW driven
    by the host formula of lr.py:17-31 (power 1): warm-up, decay and the flat tail, with a step offset."""
    from mobgt_amd import _lib
    from mobgt_amd.ops import _p, _stream

    def host_lr(c, warm=3, tot=7, peak=2e-3, end=1e-4):
        if c <= warm:
            return c / float(warm) * peak
        if c >= tot:
            return end
        return (peak - end) * (1 - (c - warm) / (tot - warm)) + end
    gen = torch.Generator().manual_seed(4)
    n = 4099
    p0 = torch.randn(n, generator=gen)
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=0.01)
    p = p0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    sched = torch.tensor([3.0, 7.0, 2e-3, 1e-4, 1.0], device=DEV)      # offset +1: optimizer call t runs at lr(t + 1)
    step_dev = torch.tensor([10], dtype=torch.int64, device=DEV)
    for it in range(9):
        g = torch.randn(n, generator=gen)
        for grp in opt.param_groups:
            grp["lr"] = host_lr(it + 2)
        ref.grad = g.double()
        opt.step()
        step_dev.add_(1)
        _lib.check(_lib.lib().mobgt_adamw_flat(_p(p), _p(g.to(DEV)), _p(m), _p(v), None, n, None, _p(sched), _p(step_dev), 10,
                                               0.9, 0.999, 1e-8, 0.01, _stream()), "mobgt_adamw_flat")
        np.testing.assert_allclose(p.cpu().numpy(), ref.detach().float().numpy(), rtol=2e-5, atol=2e-6)


def test_step_prologue_zeroes_both_buffers_and_bumps_the_counter():
    from mobgt_amd import _lib
    from mobgt_amd.ops import _p, _stream
    a = torch.randn(100003 * 4, device=DEV)
    b = torch.randn(52, device=DEV)
    guard = a[-4:].clone()
    c = torch.tensor([41], dtype=torch.int64, device=DEV)
    _lib.check(_lib.lib().mobgt_step_prologue(_p(a), a.numel() - 4, _p(b), b.numel(), _p(c), _stream()), "mobgt_step_prologue")
    assert int(c.item()) == 42
    assert not a[:-4].any() and not b.any() and torch.equal(a[-4:], guard)
    _lib.check(_lib.lib().mobgt_step_prologue(None, 0, _p(guard), 4, None, _stream()), "mobgt_step_prologue")
    assert not guard.any() and int(c.item()) == 42


@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
def test_head_input_matches_torch_chain(idt):
    """mobgt_head_input_fwd/bwd against cat(output[:, 0, :], embedding(user - 1)) and its autograd."""
    from mobgt_amd import ops
    gen = torch.Generator().manual_seed(12)
    G, T, C, U, NU = 16, 9, 192, 128, 50
    enc = torch.randn(G, T, C, generator=gen).to(DEV)
    table = torch.randn(NU, U, generator=gen).to(DEV)
    user = torch.randint(1, NU + 1, (G, 1), generator=gen)
    user[3] = user[7]                                             # a repeated user: its row receives two gradient rows
    gy = torch.randn(G, C + U, generator=gen).to(DEV)
    ea, ta = enc.clone().requires_grad_(True), table.clone().requires_grad_(True)
    ref = torch.cat((ea[:, 0, :], torch.nn.functional.embedding(user.to(DEV).long().view(-1) - 1, ta)), 1)
    ref.backward(gy)
    eb, tb = enc.clone().requires_grad_(True), table.clone().requires_grad_(True)
    got = ops.head_input(eb, tb, user.to(idt).to(DEV), -1)
    got.backward(gy)
    assert torch.equal(got, ref)
    assert torch.equal(eb.grad, ea.grad)
    np.testing.assert_allclose(tb.grad.cpu().numpy(), ta.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_head_act_matches_torch_chain(p_drop):
    """mobgt_head_act_fwd/bwd against LeakyReLU -> LayerNorm -> ELU -> (the library's own) dropout built from torch ops."""
    from mobgt_amd import ops
    gen = torch.Generator().manual_seed(9)
    R, C = 16, 448
    u = torch.randn(R, C, generator=gen).to(DEV)
    w = (1 + 0.1 * torch.randn(C, generator=gen)).to(DEV)
    b = (0.1 * torch.randn(C, generator=gen)).to(DEV)
    gy = torch.randn(R, C, generator=gen).to(DEV)
    ops.set_dropout_state(torch.tensor([3], dtype=torch.int64, device=DEV), 11)
    ua, wa, ba = (t.clone().requires_grad_(True) for t in (u, w, b))
    ref = torch.nn.functional.elu(torch.nn.functional.layer_norm(torch.nn.functional.leaky_relu(ua, 0.2), (C,), wa, ba, 1e-5))
    ref = ops.dropout(ref, p_drop, True, 0x1004)
    ref.backward(gy)
    ub, wb, bb = (t.clone().requires_grad_(True) for t in (u, w, b))
    got = ops.head_act(ub, wb, bb, 1e-5, 0.2, p_drop, True, 0x1004)
    got.backward(gy)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ub.grad.cpu().numpy(), ua.grad.cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(wb.grad.cpu().numpy(), wa.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bb.grad.cpu().numpy(), ba.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    ops.set_dropout_state(None, None)


@pytest.mark.parametrize("C", [16, 64, 192])          # 4 / 16 lanes per row (a wave walks several rows), and the wide layout
@pytest.mark.parametrize("p_drop,has_bias", [(0.0, True), (0.3, True), (0.3, False)])
def test_bias_act_matches_torch_chain(p_drop, has_bias, C):
    from mobgt_amd import ops
    gen = torch.Generator().manual_seed(12)
    R = 777
    x = torch.randn(R, C, generator=gen).to(DEV)
    b = torch.randn(C, generator=gen).to(DEV) if has_bias else None
    gy = torch.randn(R, C, generator=gen).to(DEV)
    ops.set_dropout_state(torch.tensor([5], dtype=torch.int64, device=DEV), 21)
    xa = x.clone().requires_grad_(True)
    ba = b.clone().requires_grad_(True) if has_bias else None
    ref = ops.dropout(torch.nn.functional.leaky_relu(xa + ba if has_bias else xa, 0.2), p_drop, True, 0x2040)
    ref.backward(gy)
    xb = x.clone().requires_grad_(True)
    bb = b.clone().requires_grad_(True) if has_bias else None
    got = ops.bias_act(xb, bb, 0.2, p_drop, True, 0x2040)
    got.backward(gy)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xb.grad.cpu().numpy(), xa.grad.cpu().numpy(), rtol=1e-6, atol=1e-6)
    if has_bias:
        np.testing.assert_allclose(bb.grad.cpu().numpy(), ba.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    ops.set_dropout_state(None, None)


@pytest.mark.parametrize("M,N,K", [(608, 576, 192), (608, 192, 1024), (37, 1024, 192), (608, 192, 576), (16, 8, 32)])
def test_layer_gemm_epilogues_match_torch(M, N, K):
    """mobgt_layer_gemm, both weight orientations and all four epilogues, against f64 products of the same bf16 operands
    (tolerance: bf16 rounding of the result, 2^-8 relative)."""
    from mobgt_amd import ops
    gen = torch.Generator().manual_seed(M + N + K)
    a = (torch.randn(M + 3, K, generator=gen) / K ** 0.5).to(DEV).bfloat16()[:M]
    w_nk = torch.randn(N, K, generator=gen).to(DEV).bfloat16()
    w_kn = torch.randn(K, N, generator=gen).to(DEV).bfloat16()
    bias = torch.randn(N, generator=gen).to(DEV).bfloat16()
    tol = dict(rtol=1e-2, atol=1e-2)
    for kn, w in ((False, w_nk), (True, w_kn)):
        assert ops.layer_gemm_ok(a, w, kn)
        acc = a.double() @ (w.double() if kn else w.double().t())
        ref = acc + bias.double()
        c = ops.layer_gemm(a, w, bias, kn)
        np.testing.assert_allclose(c.float().cpu().numpy(), ref.float().cpu().numpy(), **tol)
        c0 = ops.layer_gemm(a, w, None, kn)
        np.testing.assert_allclose(c0.float().cpu().numpy(), acc.float().cpu().numpy(), **tol)
        u, h = ops.layer_gemm(a, w, bias, kn, ops.GEMM_GELU)
        assert torch.equal(u, c)
        assert torch.equal(h, torch.nn.functional.gelu(u.float()).bfloat16()) or \
            (h.float() - torch.nn.functional.gelu(u.float())).abs().max() < 2e-2
        du = ops.layer_gemm(a, w, None, kn, ops.GEMM_GELU_BWD, aux_in=u)
        ud = u.double().requires_grad_(True)
        torch.nn.functional.gelu(ud).backward(acc)
        np.testing.assert_allclose(du.float().cpu().numpy(), ud.grad.float().cpu().numpy(), **tol)
        addend = torch.randn(M, N, generator=gen).to(DEV)
        want = addend.double() + acc
        got = ops.layer_gemm(a, w, None, kn, ops.GEMM_ADD, aux_in=addend)
        assert got.data_ptr() == addend.data_ptr()
        np.testing.assert_allclose(got.cpu().numpy(), want.float().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(7856, 64, 16), (300, 32, 64), (301, 17, 303), (16, 320, 320), (5, 3, 2)])
def test_small_gemm_f32_matches_torch(M, N, K):
    """mobgt_small_gemm_f32 (both B orientations, bias, strided A, ragged M / N / K) against f64 products: f32 accuracy."""
    from mobgt_amd import ops
    gen = torch.Generator().manual_seed(M * 7 + N)
    wide = torch.randn(M, K + 5, generator=gen).to(DEV)
    a = wide[:, 2:2 + K]                                       # row-strided, unaligned view
    a2 = torch.randn(M, K, generator=gen).to(DEV)
    bias = torch.randn(N, generator=gen).to(DEV)
    for nk in (False, True):
        b = torch.randn((N, K) if nk else (K, N), generator=gen).to(DEV)
        bd = b.double().t() if nk else b.double()
        for aa in (a, a2):
            want = (aa.double() @ bd + bias.double()).float()
            got = ops.small_gemm(aa, b, bias, nk)
            np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-5 * K ** 0.5)
        np.testing.assert_allclose(ops.small_gemm(a2, b, None, nk).cpu().numpy(), (a2.double() @ bd).float().cpu().numpy(),
                                   rtol=2e-5, atol=2e-5 * K ** 0.5)
        # bf16 result = the f32 result rounded
        assert torch.equal(ops.small_gemm(a2, b, bias, nk, out_dtype=torch.bfloat16), ops.small_gemm(a2, b, bias, nk).bfloat16())


def test_layer_backward_tail_matches_separate_launches():
    """mobgt_layer_backward_tail (weight gradients + the dx GEMM in one launch) against mobgt_linear_wgrad_group followed
    by mobgt_layer_gemm(MOBGT_GEMM_ADD): the same dx bit for bit, the same weight gradients up to atomic ordering."""
    from mobgt_amd import ops
    from mobgt_amd.fused_layer import _WgradBatch
    gen = torch.Generator().manual_seed(44)
    R, C = 599, 192
    dqkv = torch.randn(R, 3 * C, generator=gen).to(DEV).bfloat16()
    xa = torch.randn(R, C, generator=gen).to(DEV).bfloat16()
    dy = torch.randn(R, C, generator=gen).to(DEV).bfloat16()
    a = torch.randn(R, C, generator=gen).to(DEV).bfloat16()
    wqkv = torch.randn(3 * C, C, generator=gen).to(DEV).bfloat16()
    dx1 = torch.randn(R, C, generator=gen).to(DEV)
    res = []
    for with_tail in (False, True):
        wb = _WgradBatch()
        db = torch.zeros(3 * C, device=DEV)
        dw1 = wb.add(dqkv, xa, db=db)
        dw2 = wb.add(dy, a)
        c = dx1.clone()
        rode = wb.flush(tail=(dqkv, wqkv, c) if with_tail else None)
        assert rode == with_tail
        if not with_tail:
            ops.layer_gemm(dqkv, wqkv, None, True, ops.GEMM_ADD, aux_in=c)
        res.append((dw1.clone(), dw2.clone(), db.clone(), c))
    for u, v in zip(res[0][:3], res[1][:3]):          # split-R partial tiles meet through f32 atomics: order-dependent rounding
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-5, atol=1e-3)
    assert torch.equal(res[0][3], res[1][3])          # the GEMM itself is deterministic
    want = dx1.double() + dqkv.double() @ wqkv.double()
    np.testing.assert_allclose(res[1][3].cpu().numpy(), want.float().cpu().numpy(), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("variant", ["fq", "stock"])
def test_ln_prologue_gemms_match_the_two_launch_form(variant):
    """csrc/lngemm.hip: dropout_add_ln as the prologue of the GEMM that consumes it (FFN layer 1 forward, the GELU'
    GEMM and the output-projection data gradient backward) against the separate launches -- same arithmetic and the
    same dropout masks, so outputs and every gradient agree to bf16 round-off (the row reductions run in another order)."""
    from mobgt_amd import fused_layer
    from mobgt_amd.model import EncoderLayer as StockLayer
    from mobgt_amd.model_fqandtoyo import EncoderLayer as FqLayer
    torch.manual_seed(3)
    G, T, C, H = 5, 37, 192, 8
    layer = (FqLayer if variant == "fq" else StockLayer)(C, 256, 0.1, 0.1, H).to(DEV)
    layer.act_dtype = torch.bfloat16
    layer.train()
    seed_dev = torch.tensor([11], dtype=torch.int64, device=DEV)
    layer.self_attention.seed_dev = seed_dev             # fixed device seed: both runs draw the same masks
    x0 = torch.randn(G, T, C, device=DEV)
    bias = torch.randn(G, H, T, T, device=DEV) * 0.3
    gy = torch.randn(G, T, C, device=DEV)
    res = {}
    for on in (True, False):
        fused_layer._LN_GEMM[0] = on
        fused_layer._LN_GEMM_BWD[0] = on
        try:
            for p in layer.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            y = layer(x, bias)
            y.backward(gy)
            torch.cuda.synchronize()
            res[on] = (y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters() if p.grad is not None})
        finally:
            fused_layer._LN_GEMM[0] = True
            fused_layer._LN_GEMM_BWD[0] = False
    (ya, dxa, ga), (yb, dxb, gb) = res[True], res[False]

    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        assert err <= 1.5e-2 * scale, (name, err, scale)
    close(ya, yb, "y")
    close(dxa, dxb, "dx")
    assert ga.keys() == gb.keys()
    for n in ga:
        if n.endswith("linear_k.bias"):
            continue                                      # exactly zero in exact arithmetic: round-off only
        close(ga[n], gb[n], n)


def test_torch_library_attention_op_autograd_autocast_no_grad():
    """the attention core as a dispatcher-visible op (`torch.ops.mobgt.attention_forward`) -- values and
    gradients equal the reference expression of model.py:436-455; under torch.autocast it runs the bf16-I/O kernels; it
    works under torch.no_grad; the schema / fake implementation pass torch.library.opcheck."""
    from mobgt_amd import torch_ops
    g = torch.Generator().manual_seed(9)
    G, H, T, d = 3, 8, 29, 24
    C = H * d
    q, k, v, gy = (torch.randn(G, T, C, generator=g).to(DEV) for _ in range(4))
    bias = (torch.randn(G, H, T, T, generator=g) * 0.5).to(DEV)
    bias[1, :, :, 20:] = float("-inf")

    def ref(q, k, v, b):
        r = lambda t: t                                    # f32 I/O: the full-f32 instantiation (csrc/attn_f32_body.h)
        s = (r(q * d ** -0.5).view(G, T, H, d).transpose(1, 2) @ r(k).view(G, T, H, d).transpose(1, 2).transpose(2, 3)) + b
        return (torch.softmax(s, 3) @ r(v).view(G, T, H, d).transpose(1, 2)).transpose(1, 2).reshape(G, T, C)
    a = [t.clone().requires_grad_(True) for t in (q, k, v, bias)]
    ref(*a).backward(gy)
    b = [t.clone().requires_grad_(True) for t in (q, k, v, bias)]
    out = torch_ops.attention(b[0], b[1], b[2], b[3], H)
    out.backward(gy)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref(q, k, v, bias).cpu().numpy(), atol=1e-5, rtol=1e-4)
    for x, y, n in zip(b, a, "qkvb"):
        scale = float(y.grad.abs().max())
        np.testing.assert_allclose(x.grad.cpu().numpy(), y.grad.cpu().numpy(), atol=1e-4 * scale, rtol=1e-4, err_msg=n)
    with torch.no_grad():
        o2 = torch_ops.attention(q, k, v, bias, H)
    assert not o2.requires_grad and torch.equal(o2, out.detach())
    with torch.autocast("cuda", dtype=torch.bfloat16):
        o3 = torch_ops.attention(q, k, v, bias, H)
    assert o3.dtype == torch.bfloat16
    np.testing.assert_allclose(o3.float().cpu().numpy(), out.detach().cpu().numpy(), atol=3e-2, rtol=3e-2)
    torch.library.opcheck(torch.ops.mobgt.attention_forward.default, (q, k, v, bias, H, d ** -0.5, 0.0, 0),
                          test_utils=("test_schema", "test_faketensor"))


def test_fused_layer_under_autocast_takes_the_bf16_configuration():
    """AMP (README.md:62 --precision 16): a layer configured for fp32 activations, run under torch.autocast, gives what the
    same layer configured with act_dtype = bf16 gives, forward and backward; without autocast it stays fp32."""
    from mobgt_amd.model_fqandtoyo import EncoderLayer
    torch.manual_seed(5)
    G, T, C, H = 4, 33, 192, 8
    layer = EncoderLayer(C, 256, 0.1, 0.1, H).to(DEV).eval()
    x0 = torch.randn(G, T, C, device=DEV)
    bias = torch.randn(G, H, T, T, device=DEV) * 0.3
    gy = torch.randn(G, T, C, device=DEV)
    res = {}
    for mode in ("fp32", "amp", "bf16"):
        layer.act_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
        for p in layer.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        if mode == "amp":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = layer(x, bias)
        else:
            y = layer(x, bias)
        assert y.dtype == torch.float32                       # the residual stream stays fp32
        y.backward(gy)
        res[mode] = (y.detach().clone(), x.grad.clone(), layer.ffn.layer1.weight.grad.clone())
    for a, b in zip(res["amp"], res["bf16"]):
        assert torch.equal(a, b)
    assert not torch.equal(res["amp"][0], res["fp32"][0])
    np.testing.assert_allclose(res["amp"][0].cpu().numpy(), res["fp32"][0].cpu().numpy(), atol=5e-2, rtol=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("C,G,T,p", [(192, 5, 37, 0.1), (192, 3, 11, 0.0), (256, 2, 70, 0.1)])
def test_layer_chain_kernel_matches_the_separate_launches(C, G, T, p):
    """csrc/chain.hip: out-proj -> LN -> FFN -> LN -> the NEXT layer's QKV projection as one launch per layer, against the
    separate launches (MOBGT_NO_CHAIN path) on a 3-layer fq stack: same rounding points, same dropout masks, so outputs
    and every gradient agree to bf16 round-off (full-K accumulation here, split-K there)."""
    from mobgt_amd import fused_layer
    from mobgt_amd.model import refresh_shadows
    from mobgt_amd.model_fqandtoyo import EncoderLayer as FqLayer
    torch.manual_seed(5)
    H = 8
    layers = torch.nn.ModuleList([FqLayer(C, 1024, p, p, H) for _ in range(3)]).to(DEV)
    for li, l in enumerate(layers):
        l.act_dtype = torch.bfloat16
        l.self_attention.set_layer_index(li + 1)
        l.self_attention.seed_dev = torch.tensor([11], dtype=torch.int64, device=DEV)     # both runs draw the same masks
    layers.train()
    x0 = torch.randn(G, T, C, device=DEV)
    bias = torch.randn(G, H, T, T, device=DEV) * 0.3
    gy = torch.randn(G, T, C, device=DEV)
    res = {}
    for mode in ("both", "forward", "off"):                # chain kernels forward + backward / forward only / neither
        fused_layer._CHAIN[0] = mode != "off"
        fused_layer._CHAIN_BWD[0] = mode == "both"
        try:
            for q in layers.parameters():
                q.grad = None
            x = x0.clone().requires_grad_(True)
            refresh_shadows(layers)
            y = x
            rode = []
            for li, l in enumerate(layers):
                y = l(y, bias, next_layer=layers[li + 1] if li + 1 < len(layers) else None)
                rode.append(getattr(y, "_mobgt_qkv", None) is not None)
            assert rode == ([True, True, False] if mode != "off" else [False, False, False])     # the QKV projections did ride along
            assert y.grad_fn.chain_bwd == (mode == "both")
            y.backward(gy)
            torch.cuda.synchronize()
            res[mode] = (y.detach().clone(), x.grad.clone(), {n: q.grad.clone() for n, q in layers.named_parameters() if q.grad is not None})
        finally:
            fused_layer._CHAIN[0] = fused_layer._CHAIN_BWD[0] = True

    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        assert err <= 2e-2 * scale, (name, err, scale)
    yb, dxb, gb = res["off"]
    for mode in ("both", "forward"):
        ya, dxa, ga = res[mode]
        close(ya, yb, mode + " y")
        close(dxa, dxb, mode + " dx")
        assert ga.keys() == gb.keys()
        for n in ga:
            if n.endswith("linear_k.bias"):
                continue                                      # exactly zero in exact arithmetic: round-off only
            close(ga[n], gb[n], mode + " " + n)


@pytest.mark.gpu
@pytest.mark.parametrize("C,G,T,p", [(256, 7, 640, 0.1), (192, 9, 500, 0.0), (256, 6, 785, 0.1), (128, 11, 401, 0.1)])
def test_long_batch_forward_chain_matches_the_library_launches(C, G, T, p):
    """csrc/chain.hip, layer_chain_fwd_big_kernel / layer_chain_bwd_big_kernel (round 5): past 4 096 rows a layer's row-local part
    is ONE launch each way with 64 rows per workgroup (FFN in chunks of 384 hidden columns, split-K partial sums in the f32
    tile).  Against MOBGT_NO_CHAIN_BIG=1 (library GEMMs + csrc/layer.hip glue) on a 3-layer fq stack: same
    rounding points, same dropout masks -- outputs, input gradient and every parameter gradient to bf16 round-off.  The
    last row block is full in the first case (4 480 = 70 x 64) and ragged in the others (4 500, 4 710, 4 411 rows)."""
    from mobgt_amd import fused_layer
    from mobgt_amd.model import refresh_shadows
    from mobgt_amd.model_fqandtoyo import EncoderLayer as FqLayer
    torch.manual_seed(6)
    H = 8
    layers = torch.nn.ModuleList([FqLayer(C, 1024, p, p, H) for _ in range(3)]).to(DEV)
    for li, l in enumerate(layers):
        l.act_dtype = torch.bfloat16
        l.self_attention.set_layer_index(li + 1)
        l.self_attention.seed_dev = torch.tensor([13], dtype=torch.int64, device=DEV)
    layers.train()
    x0 = torch.randn(G, T, C, device=DEV)
    bias = torch.randn(G, H, T, T, device=DEV) * 0.3
    gy = torch.randn(G, T, C, device=DEV)
    res = {}
    for mode in ("big", "off"):
        fused_layer._CHAIN_BIG[0] = mode == "big"
        try:
            for q in layers.parameters():
                q.grad = None
            x = x0.clone().requires_grad_(True)
            refresh_shadows(layers, rows=G * T)
            y = x
            rode = []
            for li, l in enumerate(layers):
                y = l(y, bias, next_layer=layers[li + 1] if li + 1 < len(layers) else None)
                rode.append(getattr(y, "_mobgt_qkv", None) is not None)
            assert rode == ([True, True, False] if mode == "big" else [False, False, False])
            assert y.grad_fn.chain_bwd == (mode == "big")
            y.backward(gy)
            torch.cuda.synchronize()
            res[mode] = (y.detach().clone(), x.grad.clone(), {n: q.grad.clone() for n, q in layers.named_parameters() if q.grad is not None})
        finally:
            fused_layer._CHAIN_BIG[0] = True

    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        assert err <= 2e-2 * scale, (name, err, scale)
    ya, dxa, ga = res["big"]
    yb, dxb, gb = res["off"]
    close(ya, yb, "y")
    close(dxa, dxb, "dx")
    assert ga.keys() == gb.keys()
    for n in ga:
        if n.endswith("linear_k.bias"):
            continue                                          # exactly zero in exact arithmetic: round-off only
        close(ga[n], gb[n], n)


@pytest.mark.gpu
@pytest.mark.parametrize("R,K,N", [(608, 160, 160), (37, 192, 192), (16, 448, 224)])
def test_linear_with_leaky_epilogue_matches_torch(R, K, N):
    """ops.linear_splitk(..., slope): FuseEmbeddings' Linear + LeakyReLU (model_fqandtoyo.py:452-455) with the activation in
    the GEMM's epilogue and its derivative applied to the gradient inside the two backward products, against torch fp32
    (weight gradient: operands rounded to bf16 while loading, hence the tolerance)."""
    from mobgt_amd import ops
    torch.manual_seed(R)
    x = torch.randn(R, K, device=DEV, requires_grad=True)
    lin = torch.nn.Linear(K, N).to(DEV)
    gy = torch.randn(R, N, device=DEV)
    y = ops.linear_splitk(x, lin.weight, lin.bias, True, slope=0.2)
    y.backward(gy)
    got = (y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad()
    yr = torch.nn.functional.leaky_relu(lin(x), 0.2)
    yr.backward(gy)
    torch.testing.assert_close(got[0], yr.detach(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(got[1], x.grad, rtol=1e-4, atol=1e-4)
    for u, v in ((got[2], lin.weight.grad), (got[3], lin.bias.grad)):
        assert float((u - v).abs().max()) <= 1e-2 * float(v.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("R", [600, 5000])
def test_layernorm_backward_and_column_sums_short_and_long_batches(R):
    """mobgt_dropout_add_ln_bwd / mobgt_colsum / mobgt_gelu_bwd_colsum vs torch autograd: R = 600 runs the four-wave
    workgroups, R = 5000 (>= 4096) the sixteen-wave long-batch forms of csrc/layer.hip (same results, fewer
    same-address atomics)."""
    from mobgt_amd import _lib
    from mobgt_amd.fused_layer import _k1_bwd
    from mobgt_amd.ops import _p, _stream
    C, F = 256, 1024
    g = torch.Generator().manual_seed(R)
    x1 = torch.randn(R, C, generator=g).to(DEV).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    dz = torch.randn(R, C, generator=g).to(DEV).bfloat16()
    dres = torch.randn(R, C, generator=g).to(DEV)
    z = torch.nn.functional.layer_norm(x1, (C,), w, b, 1e-5)
    (z * dz.float()).sum().backward()
    mean = x1.detach().mean(1)
    rstd = (x1.detach().var(1, unbiased=False) + 1e-5).rsqrt()
    dx1 = torch.empty(R, C, device=DEV)
    dy = torch.empty(R, C, device=DEV, dtype=torch.bfloat16)
    dgamma, dbeta, dbias = (torch.zeros(C, device=DEV) for _ in range(3))
    _k1_bwd(dz, None, dres, x1.detach(), mean, rstd, w.detach(), dx1, dy, dgamma, dbeta, dbias, R, C, 0.0, 0, None, 0, _lib.BF16)
    want_dx = x1.grad + dres
    np.testing.assert_allclose(dx1.cpu().numpy(), want_dx.cpu().numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(dy.float().cpu().numpy(), want_dx.cpu().numpy(), rtol=8e-3, atol=1e-3)      # bf16 copy, p = 0
    s = float(R) ** 0.5
    np.testing.assert_allclose(dgamma.cpu().numpy(), w.grad.cpu().numpy(), rtol=1e-4, atol=2e-5 * s)
    np.testing.assert_allclose(dbeta.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-4, atol=2e-5 * s)
    np.testing.assert_allclose(dbias.cpu().numpy(), want_dx.sum(0).cpu().numpy(), rtol=1e-4, atol=2e-5 * s)      # f32 sums
    # column sums of a bf16 [R, F] gradient, plain and through gelu'
    dh = torch.randn(R, F, generator=g).to(DEV).bfloat16()
    u = torch.randn(R, F, generator=g).to(DEV).bfloat16()
    out = torch.zeros(F, device=DEV)
    _lib.check(_lib.lib().mobgt_colsum(_p(dh), _p(out), R, F, _lib.BF16, _stream()), "mobgt_colsum")
    np.testing.assert_allclose(out.cpu().numpy(), dh.float().sum(0).cpu().numpy(), rtol=1e-4, atol=2e-5 * s)
    du = torch.empty_like(dh)
    db1 = torch.zeros(F, device=DEV)
    _lib.check(_lib.lib().mobgt_gelu_bwd_colsum(_p(dh), _p(u), _p(du), _p(db1), R, F, _lib.BF16, _stream()), "mobgt_gelu_bwd_colsum")
    uf = u.float().requires_grad_(True)
    (torch.nn.functional.gelu(uf) * dh.float()).sum().backward()
    np.testing.assert_allclose(du.float().cpu().numpy(), uf.grad.cpu().numpy(), rtol=8e-3, atol=2e-3)
    np.testing.assert_allclose(db1.cpu().numpy(), uf.grad.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-4 * s)


@pytest.mark.gpu
@pytest.mark.parametrize("C,R,p", [(192, 185, 0.1), (192, 1040, 0.1), (256, 140, 0.0), (256, 1500, 0.1), (192, 16, 0.1)])
def test_chain_cluster_form_equals_the_one_workgroup_form(C, R, p):
    """csrc/chain.hip, cluster form (a 16-row block shared by 4 / 2 workgroups that split the weights and meet through the
    workspace) against the one-workgroup form (ws = null) through the C ABI, forward and backward (with the upper layer's tail
    product): everything in front of the first split-K sum is BIT-identical (x1, z, u, h / df, du); behind it the f32 sums
    are added in 4 / 2 parts before the same bf16 rounding point, so values agree to one bf16 ulp of the rounded
    intermediate.  Runs every launch twice (the hand-over counters must carry over) and at R = 1040 / 1500 the 2-member
    form."""
    import ctypes
    from mobgt_amd import _lib
    from mobgt_amd.fused_layer import chain_workspace
    from mobgt_amd.ops import _p, _stream
    lib = _lib.lib()
    F = 1024
    g = torch.Generator().manual_seed(C + R)
    bf = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(DEV).bfloat16()
    f32 = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(DEV)

    def pack(w, transposed=False):
        out = torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        N, K = (w.shape[1], w.shape[0]) if transposed else w.shape
        _lib.check(lib.mobgt_pack_mfma_b(1, (vp * 1)(w.data_ptr()), (vp * 1)(out.data_ptr()), (ci * 1)(N), (ci * 1)(K),
                                         (ci * 1)(1 if transposed else 0), _stream()), "mobgt_pack_mfma_b")
        return out
    wo, w1, w2, wq = bf(C, C, k=C ** -0.5), bf(F, C, k=C ** -0.5), bf(C, F, k=F ** -0.5), bf(3 * C, C, k=C ** -0.5)
    bo, b1, b2, bq = bf(C, k=0.1), bf(F, k=0.1), bf(C, k=0.1), bf(3 * C, k=0.1)
    n1w, n1b, nxw, nxb = 1 + f32(C, k=0.1), f32(C, k=0.1), 1 + f32(C, k=0.1), f32(C, k=0.1)
    a, x = bf(R, C), f32(R, C)
    pk = {k: pack(w) for k, w in (("wo", wo), ("w1", w1), ("w2", w2), ("wq", wq))}
    pkt = {k: pack(w, True) for k, w in (("wo", wo), ("w1", w1), ("w2", w2), ("wq", wq))}
    ws = chain_workspace(a.device, C, R)
    assert ws is not None and ws.numel() == lib.mobgt_chain_ws_bytes()
    seed_dev = torch.tensor([5], dtype=torch.int64, device=DEV)

    def forward(w):
        o = dict(x1=torch.empty(R, C, device=DEV), x2=torch.empty(R, C, device=DEV), out=torch.empty(R, C, device=DEV),
                 z=bf(R, C), out_a=bf(R, C), u=bf(R, F), h=bf(R, F), qkv=bf(R, 3 * C), st=torch.empty(4, R, device=DEV))
        for _ in range(2):
            _lib.check(lib.mobgt_layer_chain_fwd(_p(a), _p(x), _p(pk["wo"]), _p(bo), _p(n1w), _p(n1b), _p(pk["w1"]), _p(b1), _p(pk["w2"]),
                                                 _p(b2), _p(nxw), _p(nxb), _p(pk["wq"]), _p(bq), _p(o["x1"]), _p(o["z"]), _p(o["u"]),
                                                 _p(o["h"]), _p(o["x2"]), _p(o["out"]), _p(o["out_a"]), _p(o["qkv"]), _p(o["st"][0]),
                                                 _p(o["st"][1]), _p(o["st"][2]), _p(o["st"][3]), R, C, F, p, 77, _p(seed_dev), 9, 10,
                                                 _p(w), _stream()), "mobgt_layer_chain_fwd")
        torch.cuda.synchronize()
        return o
    one, cl = forward(None), forward(ws)
    for k in ("x1", "z", "u", "h"):
        assert torch.equal(one[k], cl[k]), k
    assert torch.equal(one["st"][:2], cl["st"][:2])
    for k, tol in (("x2", 2 ** -7), ("out", 2 ** -6), ("qkv", 2 ** -5)):
        d = float((one[k].float() - cl[k].float()).abs().max())
        frac = float((one[k].float() != cl[k].float()).float().mean())
        assert d <= tol * max(1.0, float(one[k].float().abs().max())), (k, d)
        assert frac < 0.2, (k, frac)                       # (most sums round to the same bf16 value)

    # ---- backward, with the upper layer's tail (dout holds its dx1; dout + dqkv Wqkv is finished inside)
    dout, dqkv = f32(R, C), bf(R, 3 * C, k=0.3)
    fw = one

    def backward(w):
        o = dict(df=bf(R, C), du=bf(R, F), dy=bf(R, C), da=bf(R, C), dx1=torch.empty(R, C, device=DEV),
                 sums=torch.zeros(6, C, device=DEV))
        for it in range(2):
            o["sums"].zero_()
            s = o["sums"]
            _lib.check(lib.mobgt_layer_chain_bwd(_p(dout), _p(fw["x2"]), _p(fw["x1"]), _p(fw["u"]), _p(fw["st"][0]), _p(fw["st"][1]),
                                                 _p(fw["st"][2]), _p(fw["st"][3]), _p(n1w), _p(nxw), _p(pkt["w2"]), _p(pkt["w1"]),
                                                 _p(pkt["wo"]), _p(o["df"]), _p(o["du"]), _p(o["dy"]), _p(o["da"]), _p(o["dx1"]),
                                                 _p(s[0]), _p(s[1]), _p(s[2]), _p(s[3]), _p(s[4]), _p(s[5]), R, C, F, p, 77, _p(seed_dev),
                                                 9, 10, _p(dqkv), _p(pkt["wq"]), 0, None, None, None, None, None, None, None, None,
                                                 None, _p(w), _stream()), "mobgt_layer_chain_bwd")
        torch.cuda.synchronize()
        return o
    one, cl = backward(None), backward(ws)
    for k in ("df", "du"):
        assert torch.equal(one[k], cl[k]), k
    for k in ("dy", "da", "dx1"):
        sc = max(1.0, float(one[k].float().abs().max()))
        d = float((one[k].float() - cl[k].float()).abs().max())
        assert d <= 2 ** -6 * sc, (k, d, sc)
    sc = one["sums"].abs().amax(1, keepdim=True).clamp_min(1e-6)
    assert float(((one["sums"] - cl["sums"]).abs() / sc).max()) <= 1e-2       # (f32 atomics in another order + the above)


@pytest.mark.gpu
@pytest.mark.parametrize("C,R,p", [(192, 185, 0.1), (256, 140, 0.0), (256, 333, 0.1), (128, 70, 0.1)])
def test_chain_64_row_backward_entry_equals_the_16_row_form(C, R, p):
    """csrc/chain.hip through the C ABI: mobgt_layer_chain_bwd_big (64-row workgroups, FFN in three chunks, norms with four
    columns per lane, b1's gradient summed inside) against mobgt_layer_chain_bwd in its one-workgroup 16-row form on the same
    saved tensors of a 16-row forward, at ragged R (one to six row blocks): df is bit-identical up to the norm's summation
    order, everything behind it to bf16 round-off; db1 against the column sums of du.  And the dispatch rule: past 4 096 rows
    mobgt_layer_chain_bwd takes the 64-row form, which has no guests -- a hosted tail is refused (MOBGT_EBADDIM)."""
    import ctypes
    from mobgt_amd import _lib
    from mobgt_amd.ops import _p, _stream
    lib = _lib.lib()
    F = 1024
    g = torch.Generator().manual_seed(3 * C + R)
    bf = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(DEV).bfloat16()
    f32 = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).to(DEV)

    def pack(w, transposed=False):
        out = torch.empty(w.numel(), dtype=torch.bfloat16, device=DEV)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        N, K = (w.shape[1], w.shape[0]) if transposed else w.shape
        _lib.check(lib.mobgt_pack_mfma_b(1, (vp * 1)(w.data_ptr()), (vp * 1)(out.data_ptr()), (ci * 1)(N), (ci * 1)(K),
                                         (ci * 1)(1 if transposed else 0), _stream()), "mobgt_pack_mfma_b")
        return out
    wo, w1, w2 = bf(C, C, k=C ** -0.5), bf(F, C, k=C ** -0.5), bf(C, F, k=F ** -0.5)
    bo, b1, b2 = bf(C, k=0.1), bf(F, k=0.1), bf(C, k=0.1)
    n1w, n1b, nxw, nxb = 1 + f32(C, k=0.1), f32(C, k=0.1), 1 + f32(C, k=0.1), f32(C, k=0.1)
    a, x = bf(R, C), f32(R, C)
    pk = {k: pack(w) for k, w in (("wo", wo), ("w1", w1), ("w2", w2))}
    pkt = {k: pack(w, True) for k, w in (("wo", wo), ("w1", w1), ("w2", w2))}
    seed_dev = torch.tensor([5], dtype=torch.int64, device=DEV)
    fw = dict(x1=torch.empty(R, C, device=DEV), x2=torch.empty(R, C, device=DEV), out=torch.empty(R, C, device=DEV),
              z=bf(R, C), out_a=bf(R, C), u=bf(R, F), h=bf(R, F), st=torch.empty(4, R, device=DEV))
    _lib.check(lib.mobgt_layer_chain_fwd(_p(a), _p(x), _p(pk["wo"]), _p(bo), _p(n1w), _p(n1b), _p(pk["w1"]), _p(b1), _p(pk["w2"]), _p(b2),
                                         _p(nxw), _p(nxb), None, None, _p(fw["x1"]), _p(fw["z"]), _p(fw["u"]), _p(fw["h"]), _p(fw["x2"]),
                                         _p(fw["out"]), _p(fw["out_a"]), None, _p(fw["st"][0]), _p(fw["st"][1]), _p(fw["st"][2]),
                                         _p(fw["st"][3]), R, C, F, p, 77, _p(seed_dev), 9, 10, None, _stream()), "mobgt_layer_chain_fwd")
    dout = f32(R, C)

    def outs():
        return dict(df=bf(R, C), du=bf(R, F), dy=bf(R, C), da=bf(R, C), dx1=torch.empty(R, C, device=DEV),
                    sums=torch.zeros(6, C, device=DEV), db1=torch.zeros(F, device=DEV))
    small, big = outs(), outs()
    s = small["sums"]
    _lib.check(lib.mobgt_layer_chain_bwd(_p(dout), _p(fw["x2"]), _p(fw["x1"]), _p(fw["u"]), _p(fw["st"][0]), _p(fw["st"][1]), _p(fw["st"][2]),
                                         _p(fw["st"][3]), _p(n1w), _p(nxw), _p(pkt["w2"]), _p(pkt["w1"]), _p(pkt["wo"]), _p(small["df"]),
                                         _p(small["du"]), _p(small["dy"]), _p(small["da"]), _p(small["dx1"]), _p(s[0]), _p(s[1]), _p(s[2]),
                                         _p(s[3]), _p(s[4]), _p(s[5]), R, C, F, p, 77, _p(seed_dev), 9, 10, None, None, 0, None, None,
                                         None, None, None, None, None, None, None, None, _stream()), "mobgt_layer_chain_bwd")
    s = big["sums"]
    _lib.check(lib.mobgt_layer_chain_bwd_big(_p(dout), _p(fw["x2"]), _p(fw["x1"]), _p(fw["u"]), _p(fw["st"][0]), _p(fw["st"][1]),
                                             _p(fw["st"][2]), _p(fw["st"][3]), _p(n1w), _p(nxw), _p(pkt["w2"]), _p(pkt["w1"]), _p(pkt["wo"]),
                                             _p(big["df"]), _p(big["du"]), _p(big["dy"]), _p(big["da"]), _p(big["dx1"]), _p(s[0]), _p(s[1]),
                                             _p(s[2]), _p(s[3]), _p(s[4]), _p(s[5]), _p(big["db1"]), R, C, F, p, 77, _p(seed_dev), 9, 10,
                                             None, None, _stream()), "mobgt_layer_chain_bwd_big")
    torch.cuda.synchronize()
    for k in ("df", "du", "dy", "da", "dx1"):
        sc = max(1.0, float(small[k].float().abs().max()))
        d = float((small[k].float() - big[k].float()).abs().max())
        assert d <= 2 ** -6 * sc, (k, d, sc)
        if small[k].dtype == torch.bfloat16:                                               # (most values round to the same bf16)
            assert float((small[k].float() != big[k].float()).float().mean()) < 0.2, k
    sc = small["sums"].abs().amax(1, keepdim=True).clamp_min(1e-6)
    assert float(((small["sums"] - big["sums"]).abs() / sc).max()) <= 1e-2
    ref_db1 = big["du"].float().sum(0)
    assert float((big["db1"] - ref_db1).abs().max()) <= 1e-3 * max(1.0, float(ref_db1.abs().max()))
    # ... and with the upper layer's tail hosted by both (dout = its dx1; dout + dqkv Wqkv in front of the first norm)
    dqkv = bf(R, 3 * C, k=0.3)
    wq = bf(3 * C, C, k=C ** -0.5)
    wqt = pack(wq, True)
    small2, big2 = outs(), outs()
    s = small2["sums"]
    _lib.check(lib.mobgt_layer_chain_bwd(_p(dout), _p(fw["x2"]), _p(fw["x1"]), _p(fw["u"]), _p(fw["st"][0]), _p(fw["st"][1]), _p(fw["st"][2]),
                                         _p(fw["st"][3]), _p(n1w), _p(nxw), _p(pkt["w2"]), _p(pkt["w1"]), _p(pkt["wo"]), _p(small2["df"]),
                                         _p(small2["du"]), _p(small2["dy"]), _p(small2["da"]), _p(small2["dx1"]), _p(s[0]), _p(s[1]), _p(s[2]),
                                         _p(s[3]), _p(s[4]), _p(s[5]), R, C, F, p, 77, _p(seed_dev), 9, 10, _p(dqkv), _p(wqt), 0, None, None,
                                         None, None, None, None, None, None, None, None, _stream()), "mobgt_layer_chain_bwd")
    s = big2["sums"]
    _lib.check(lib.mobgt_layer_chain_bwd_big(_p(dout), _p(fw["x2"]), _p(fw["x1"]), _p(fw["u"]), _p(fw["st"][0]), _p(fw["st"][1]),
                                             _p(fw["st"][2]), _p(fw["st"][3]), _p(n1w), _p(nxw), _p(pkt["w2"]), _p(pkt["w1"]), _p(pkt["wo"]),
                                             _p(big2["df"]), _p(big2["du"]), _p(big2["dy"]), _p(big2["da"]), _p(big2["dx1"]), _p(s[0]), _p(s[1]),
                                             _p(s[2]), _p(s[3]), _p(s[4]), _p(s[5]), _p(big2["db1"]), R, C, F, p, 77, _p(seed_dev), 9, 10,
                                             _p(dqkv), _p(wqt), _stream()), "mobgt_layer_chain_bwd_big")
    torch.cuda.synchronize()
    assert not torch.equal(small2["df"], small["df"])                       # (the tail did change the gradient)
    for k in ("df", "du", "dy", "da", "dx1"):
        sc = max(1.0, float(small2[k].float().abs().max()))
        d = float((small2[k].float() - big2[k].float()).abs().max())
        assert d <= 2 ** -6 * sc, ("tail", k, d, sc)
    # the dispatch rule of the 16-row entry point: a hosted tail past 4 096 rows is refused
    Rb = 4100
    z16 = torch.zeros(Rb, 3 * C, dtype=torch.bfloat16, device=DEV)
    zf = torch.zeros(Rb, C, device=DEV)
    zu = torch.zeros(Rb, F, dtype=torch.bfloat16, device=DEV)
    zs = torch.ones(Rb, device=DEV)
    wqt0 = torch.zeros(3 * C * C, dtype=torch.bfloat16, device=DEV)
    rc = lib.mobgt_layer_chain_bwd(_p(zf), _p(zf), _p(zf), _p(zu), _p(zs), _p(zs), _p(zs), _p(zs), _p(n1w), _p(nxw), _p(pkt["w2"]),
                                   _p(pkt["w1"]), _p(pkt["wo"]), _p(z16[:, :C].contiguous()), _p(zu.clone()), _p(z16[:, :C].contiguous()),
                                   _p(z16[:, :C].contiguous()), _p(zf.clone()), _p(s[0]), _p(s[1]), _p(s[2]), _p(s[3]), _p(s[4]), _p(s[5]),
                                   Rb, C, F, p, 77, _p(seed_dev), 9, 10, _p(z16), _p(wqt0), 0, None, None, None, None, None, None, None,
                                   None, None, None, _stream())
    assert rc == -1, rc                                     # MOBGT_EBADDIM


@pytest.mark.gpu
@pytest.mark.parametrize("G,T,C,p", [(16, 38, 192, 0.1), (5, 9, 192, 0.0), (16, 20, 256, 0.1), (33, 12, 192, 0.1)])
def test_head_chain_equals_the_three_launches(G, T, C, p):
    """csrc/head.hip (token row | user row -> FuseEmbeddings' Linear -> LeakyReLU -> LayerNorm -> ELU -> dropout, one launch
    each way) against ops.head_input + linear_splitk + head_act: same full-f32 products (another summation order), the
    same dropout mask; d(enc) must be exactly zero off the token rows."""
    from mobgt_amd import ops
    U = 128
    W = C + U
    gen = torch.Generator().manual_seed(G * T)
    enc0 = torch.randn(G, T, C, generator=gen).to(DEV)
    table = torch.randn(50, U, generator=gen).to(DEV).requires_grad_(True)
    user = torch.randint(0, 52, (G,), generator=gen).to(DEV)              # (user - 1 in [-1, 50]: -1 and 50 read as zero rows)
    lin = torch.nn.Linear(W, W).to(DEV)
    ln = torch.nn.LayerNorm(W).to(DEV)
    with torch.no_grad():
        ln.weight.add_(0.1 * torch.randn(W, device=DEV))
        ln.bias.add_(0.1 * torch.randn(W, device=DEV))
    gy = torch.randn(G, W, generator=gen).to(DEV)
    sd = torch.tensor([3], dtype=torch.int64, device=DEV)
    ops.set_dropout_state(sd, 1234)
    try:
        res = []
        for fused in (True, False):
            for q in (table, lin.weight, lin.bias, ln.weight, ln.bias):
                q.grad = None
            enc = enc0.clone().requires_grad_(True)
            if fused:
                assert ops.head_chain_ok(enc, table, user, lin.weight)
                tok = ops.head_chain(enc, table, user, -1, lin.weight, lin.bias, ln.weight, ln.bias, ln.eps, 0.2, p, True, 0x1004,
                                     bf16_wgrad=False)
            else:
                x3 = ops.head_input(enc, table, user, -1)
                u3 = ops.linear_splitk(x3, lin.weight, lin.bias, False)
                tok = ops.head_act(u3, ln.weight, ln.bias, ln.eps, 0.2, p, True, 0x1004)
            tok.backward(gy)
            torch.cuda.synchronize()
            res.append([tok.detach().clone(), enc.grad.clone()] + [q.grad.clone() for q in (table, lin.weight, lin.bias, ln.weight, ln.bias)])
    finally:
        ops.set_dropout_state(None, 0)
    a, b = res
    assert torch.equal(a[0] == 0, b[0] == 0)                              # the same mask
    assert float(a[1][:, 1:].abs().max()) == 0.0 and float(b[1][:, 1:].abs().max()) == 0.0
    for name, x, y in zip(("tok", "denc", "dtable", "dW3", "db3", "dgamma", "dbeta"), a, b):
        sc = float(y.abs().max()) + 1e-12
        err = float((x - y).abs().max())
        assert err <= 2e-5 * max(sc, 1.0) + 1e-5 * sc, (name, err, sc)


@pytest.mark.gpu
def test_weight_pack_as_passenger_of_the_category_gcn_launch_equals_the_pack_launch():
    """mobgt_small_gcn_fwd_pack (csrc/smallgcn.hip + csrc/pack_body.h, front_body.h): the step's MFMA-order weight pack, the node
    features' index derivation and the hop table's forward carried by the category GCN's forward launch as passenger workgroups
    -- packs bit-identical to mobgt_pack_mfma_b's, indices to mobgt_node_index's, the hop table to mobgt_hop_table_fwd's, the
    network's outputs bit-identical to the launch without passengers."""
    import ctypes
    from mobgt_amd import _lib
    from mobgt_amd.ops import _p, _stream
    lib = _lib.lib()
    g = torch.Generator().manual_seed(3)
    n, K0, H1, H2, H3 = 300, 300, 16, 64, 32
    A = (torch.rand(n, n, generator=g) / n).to(DEV)
    AX = torch.randn(n, K0, generator=g).to(DEV)
    ws = [torch.randn(K0, H1, generator=g).to(DEV) * 0.1, torch.zeros(H1, device=DEV), torch.randn(H1, H2, generator=g).to(DEV) * 0.1,
          torch.zeros(H2, device=DEV), torch.randn(H2, H3, generator=g).to(DEV) * 0.1, torch.zeros(H3, device=DEV)]
    C, F = 192, 1024
    srcs = [torch.randn(s, generator=g).to(DEV).bfloat16() for _ in range(6) for s in ((3 * C, C), (C, C), (F, C), (C, F))]
    jobs = [(w, 0) for w in srcs] + [(w, 1) for w in srcs[:12]]                # 24 forward + 12 transposed packs

    def arrays(dsts):
        nj = len(jobs)
        vp, ci = ctypes.c_void_p, ctypes.c_int
        N = [w.shape[1] if t else w.shape[0] for w, t in jobs]
        K = [w.shape[0] if t else w.shape[1] for w, t in jobs]
        return nj, (vp * nj)(*[w.data_ptr() for w, _ in jobs]), (vp * nj)(*[d.data_ptr() for d in dsts]), (ci * nj)(*N), (ci * nj)(*K), \
            (ci * nj)(*[t for _, t in jobs])

    # node_index: 16 trajectories of 20 positions (ragged: pads = 0) over 5000 POIs; hop table: 20 hops x 40 edge ids x 8 heads
    G, N, P = 16, 20, 5000
    x = torch.randint(1, P + 1, (G, N), generator=g)
    x[torch.arange(N)[None, :] >= torch.randint(3, N + 1, (G, 1), generator=g)] = 0
    x = x.to(DEV)
    tn = torch.rand(G, N, generator=g).to(DEV)
    poi2cat = torch.randint(0, 300, (P,), generator=g).to(DEV)
    indeg, outdeg = (torch.randint(0, 64, (G, N), generator=g).to(DEV) for _ in range(2))
    D, E, H = 20, 40, 8
    ew, dw = torch.randn(E, H, generator=g).to(DEV), torch.randn(D * H * H, 1, generator=g).to(DEV)

    def front():
        idx = torch.full((8, G, N), -7, dtype=torch.int64, device=DEV)
        real = torch.full((G, N), -7.0, device=DEV)
        tab = torch.full((D, E, H), -7.0, device=DEV)
        ni = [_p(x), _lib.I64, x.stride(0), x.stride(1), _p(tn), tn.stride(0), tn.stride(1), _p(poi2cat), _p(indeg), _p(outdeg), _lib.I64,
              _p(idx), _p(real), G, N, 1]
        hop = [_p(ew), _p(dw), _p(tab), D, E, H, 1]
        return (idx, real, tab), ni, hop

    def run(with_pack):
        outs = [torch.empty(n, w, device=DEV) for w in (H1, H1, H2, H2, H3)]
        fr, ni, hop = front()
        dsts = [torch.zeros(w.numel(), dtype=torch.bfloat16, device=DEV) for w, _ in jobs]
        counter = torch.zeros(4, dtype=torch.int32, device=DEV)
        nj, src, dst, N, K, T = arrays(dsts)
        if with_pack:
            _lib.check(lib.mobgt_small_gcn_fwd_pack(_p(AX), _p(A), *[_p(w) for w in ws], *[_p(o) for o in outs], _p(counter), n, K0, H1, H2,
                                                    H3, 0.2, 0.3, 5, None, 7, nj, src, dst, N, K, T, 1, *ni, 1, *hop, _stream()), "fwd_pack")
        else:
            _lib.check(lib.mobgt_pack_mfma_b(nj, src, dst, N, K, T, _stream()), "pack")
            _lib.check(lib.mobgt_small_gcn_fwd(_p(AX), _p(A), *[_p(w) for w in ws], *[_p(o) for o in outs], _p(counter), n, K0, H1, H2, H3,
                                               0.2, 0.3, 5, None, 7, _stream()), "fwd")
            _lib.check(lib.mobgt_node_index(*ni, _stream()), "node_index")
            _lib.check(lib.mobgt_hop_table_fwd(*hop, _stream()), "hop_table_fwd")
        torch.cuda.synchronize()
        return outs + list(fr), dsts
    (o1, d1), (o2, d2) = run(True), run(False)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    assert int((o1[5] == -7).sum()) == 0 and float(o1[7].abs().sum()) > 0
    for k, (a, b) in enumerate(zip(d1, d2)):
        assert torch.equal(a, b), k
    assert float(d1[0].float().abs().sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("G,K,V", [(16, 320, 7856), (5, 448, 1500), (16, 64, 1024), (16, 384, 100001)])
def test_classifier_and_loss_in_one_launch(G, K, V):
    """mobgt_skinny_linear_gtl (csrc/skinny.hip): out_proj + GradientTailLoss(alpha = 0.2) on y - 1 (model_fqandtoyo.py:1394,
    :1446-1460, :545-550) in one launch against (a) the torch restatement of the loss on F.linear in float64 and (b) the two
    launches it replaces -- loss, and the gradients of the tokens, the weight and the bias; V not a multiple of 32, G < 16."""
    import types
    from mobgt_amd import ops
    from mobgt_amd.model_fqandtoyo import GradientTailLoss
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(G, K, generator=g).to(DEV)
    lin = torch.nn.Linear(K, V).to(DEV)
    with torch.no_grad():
        lin.weight.mul_(3.0)                                  # (logits of a few units: both branches of the loss matter)
    y = torch.randint(1, V + 1, (G, 1), generator=g).to(DEV)
    assert ops.skinny_linear_gtl_ok(x0, lin.weight)
    res = []
    for mode in ("fused", "two", "ref"):
        lin.weight.grad = lin.bias.grad = None
        x = x0.clone().requires_grad_(True)
        keep = types.SimpleNamespace(t=None)
        if mode == "fused":
            loss = ops.skinny_linear_gtl(x, lin.weight, lin.bias, y, 0.2, target_offset=-1, logits_out=keep)
        elif mode == "two":
            keep.t = ops.skinny_linear(x, lin.weight, lin.bias)
            loss = ops.gradient_tail_loss(keep.t, y, 0.2, target_offset=-1)
        else:
            keep.t = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
            loss = GradientTailLoss(keep.t, y.view(-1) - 1, alpha=0.2)
        loss.backward()
        torch.cuda.synchronize()
        res.append([loss.detach().double(), keep.t.detach().double(), x.grad.double(), lin.weight.grad.double().clone(), lin.bias.grad.double().clone()])
    fused, two, ref = res
    for name, a, b, c in zip(("loss", "logits", "dx", "dW", "db"), fused, two, ref):
        sc = float(c.abs().max()) + 1e-30
        assert float((a - c).abs().max()) <= 2e-5 * sc, (name, "vs float64", float((a - c).abs().max()), sc)
        assert float((a - b).abs().max()) <= 2e-5 * sc, (name, "vs two launches", float((a - b).abs().max()), sc)
    # without the logits: nothing but the loss and the gradients leave the kernel
    x = x0.clone().requires_grad_(True)
    lin.weight.grad = None
    loss = ops.skinny_linear_gtl(x, lin.weight, lin.bias, y, 0.2, target_offset=-1)
    loss.backward()
    assert torch.equal(loss.detach().double(), fused[0]) and torch.equal(lin.weight.grad.double(), fused[3])


@pytest.mark.gpu
@pytest.mark.parametrize("G,V", [(16, 7857), (3, 1000), (40, 10240)])
def test_cross_entropy_in_one_launch(G, V):
    """mobgt_cross_entropy (csrc/layer.hip) -- the stock variant's loss, model.py:218-285 with ignore_index = 0 (data.py:76 / :98)
    -- against F.cross_entropy in float64: value and gradient, with ignored rows, and with every row ignored (nan, zero
    gradient -- as torch)."""
    from mobgt_amd import ops
    g = torch.Generator().manual_seed(G)
    z0 = (torch.randn(G, V, generator=g) * 3).to(DEV)
    y = torch.randint(1, V, (G,), generator=g).to(DEV)
    y[::3] = 0                                                            # ignored rows
    for tgt in (y, torch.zeros_like(y)):
        z = z0.clone().requires_grad_(True)
        loss = ops.cross_entropy(z, tgt, ignore_index=0)
        loss.backward()
        zr = z0.double().clone().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(zr, tgt, ignore_index=0)
        ref.backward()
        if bool((tgt != 0).any()):
            assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-6, (float(loss), float(ref))
            assert float((z.grad.double() - zr.grad).abs().max()) <= 2e-6 * float(zr.grad.abs().max()) + 1e-9
        else:
            assert torch.isnan(loss) and torch.isnan(ref) and float(z.grad.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("G,T,C", [(16, 42, 128), (3, 7, 190), (5, 1, 1024)])
def test_token_layer_norm(G, T, C):
    """ops.token_layer_norm (csrc/layer.hip: final_ln on the graph-token rows, model.py:211-217) against torch's LayerNorm on
    enc[:, 0, :] in float64: value, d(enc) (zero outside the token rows) and the affine gradients."""
    from mobgt_amd import ops
    g = torch.Generator().manual_seed(C)
    enc0 = torch.randn(G, T, C, generator=g).to(DEV)
    ln = torch.nn.LayerNorm(C).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g).to(DEV)); ln.bias.copy_(torch.randn(C, generator=g).to(DEV))
    gy = torch.randn(G, C, generator=g).to(DEV)
    enc = enc0.clone().requires_grad_(True)
    y = ops.token_layer_norm(enc, ln.weight, ln.bias, ln.eps)
    y.backward(gy)
    got = [y.detach().double(), enc.grad.double(), ln.weight.grad.double().clone(), ln.bias.grad.double().clone()]
    ln.weight.grad = ln.bias.grad = None
    ref_ln = torch.nn.LayerNorm(C).to(DEV).double()
    with torch.no_grad():
        ref_ln.weight.copy_(ln.weight.double()); ref_ln.bias.copy_(ln.bias.double())
    encr = enc0.double().clone().requires_grad_(True)
    yr = ref_ln(encr[:, 0, :])
    yr.backward(gy.double())
    want = [yr.detach(), encr.grad, ref_ln.weight.grad, ref_ln.bias.grad]
    for name, a, b in zip(("y", "denc", "dgamma", "dbeta"), got, want):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-12), (name, float((a - b).abs().max()))
    if T > 1:
        assert float(got[1][:, 1:].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_stock_encoder_input_in_one_launch(idt, monkeypatch):
    """ops.stock_tokens (csrc/layer.hip; model.py:193-205) against the three ops it replaces -- embed_gather_sum, cat with the
    graph token, ops.dropout at the same site -- with the dropout on: bit-identical output (same mask), table / token gradients
    to f32 summation order; padding rows (index 0) receive no gradient; and against plain torch without dropout."""
    from mobgt_amd import ops
    g = torch.Generator().manual_seed(4)
    G, N, C, P = 16, 41, 128, 900
    x = torch.randint(0, P, (G, N), generator=g).to(idt).to(DEV)
    deg = torch.randint(0, 40, (G, N), generator=g).to(idt).to(DEV)
    tabs = [torch.nn.Parameter(torch.randn(n, C, generator=g).to(DEV)) for n in (P, 512, 512)]
    gtok = torch.nn.Parameter(torch.randn(1, C, generator=g).to(DEV))
    gy = torch.randn(G, N + 1, C, generator=g).to(DEV)
    seed_dev = torch.tensor([7], dtype=torch.int64, device=DEV)
    ops.set_dropout_state(seed_dev, 1234)
    try:
        res = []
        for fused in (True, False):
            for q in tabs + [gtok]:
                q.grad = None
            if fused:
                assert ops.stock_tokens_ok(x, *tabs, gtok)
                y = ops.stock_tokens(x, deg, deg, *tabs, gtok, 0.1, True, 0x1003)
            else:
                nf = ops.embed_gather_sum(list(tabs), [x, deg, deg], padding_idx=[0, 0, 0])
                y = ops.dropout(torch.cat([gtok.unsqueeze(0).expand(G, -1, -1), nf], dim=1), 0.1, True, 0x1003)
            y.backward(gy)
            torch.cuda.synchronize()
            res.append([y.detach().clone()] + [q.grad.clone() for q in tabs + [gtok]])
        a, b = res
        assert torch.equal(a[0], b[0]) and int((a[0] == 0).sum()) > 0
        for k, (u, v) in enumerate(zip(a[1:], b[1:])):
            assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 1e-7, k
            if k < 3:
                assert float(u[0].abs().max()) == 0.0
        # the degrees in a narrower dtype of their own (the device collator's int16): no cast, same result
        y16 = ops.stock_tokens(x, deg.to(torch.int16), deg.to(torch.int16), *[t.detach() for t in tabs], gtok.detach(), 0.1, True, 0x1003)
        assert torch.equal(y16, a[0])
        y0 = ops.stock_tokens(x, deg, deg, *[t.detach() for t in tabs], gtok.detach(), 0.1, False, 0x1003)
        ref = torch.cat([gtok.detach().unsqueeze(0).expand(G, -1, -1), tabs[0].detach()[x.long()] + tabs[1].detach()[deg.long()]
                         + tabs[2].detach()[deg.long()]], dim=1)
        assert torch.equal(y0, ref)
    finally:
        ops.set_dropout_state(None, 0)


@pytest.mark.gpu
def test_stock_front_as_one_grid_is_bit_identical_to_its_three_launches(monkeypatch):
    """Round 4: the stock model's encoder input, hop table forward and weight pack as one grid (csrc/layer.hip
    stock_front_kernel via mobgt_stock_front_fwd) against their own launches (MOBGT_NO_STOCK_FRONT=1): the three jobs are
    independent and each block runs the same body, so the packed weights, the eval logits (through the hop table and the
    encoder input) and the train-mode encoder input (device-seeded mask) are bit-identical."""
    from mobgt_amd import ops, workloads, synth
    uni, model, coll = workloads.build("fsq", DEV, seed=1, P=1500, variant="stock", model_overrides=dict(n_layers=2))
    batch = coll(synth.make_batch_of_trajectories(seed=21, G=8, P=1500, n_user=1080, cat_of_poi=uni.cat_of_poi,
                                                  n_nodes=[17, 3, 9, 2, 11, 5, 40, 23]))
    seen = []
    real = ops._lib.lib
    class _Spy:
        def __init__(self, lib): self._lib = lib
        def __getattr__(self, name):
            if name in ("mobgt_stock_front_fwd", "mobgt_stock_tokens_fwd", "mobgt_hop_table_fwd", "mobgt_pack_mfma_b"):
                seen.append(name)
            return getattr(self._lib, name)
    spy = _Spy(real())
    monkeypatch.setattr(ops._lib, "lib", lambda: spy)
    rows, tokens = [], ops.stock_tokens
    monkeypatch.setattr(ops, "stock_tokens", lambda *a, **k: (lambda y: (rows.append(y.detach().clone()), y)[1])(tokens(*a, **k)))
    ops.set_dropout_state(torch.tensor([3], dtype=torch.int64, device=DEV), 99)
    try:
        out = {}
        for off in ("", "1"):
            if off:
                monkeypatch.setenv("MOBGT_NO_STOCK_FRONT", "1")
            else:
                monkeypatch.delenv("MOBGT_NO_STOCK_FRONT", raising=False)
            del seen[:]
            model.eval()
            with torch.no_grad():
                logits = model(batch).clone()
            packs = [t.clone() for layer in model.layers for t in (getattr(layer, "_packed", None) or ())]
            model.train()
            n_eval = list(seen)
            del seen[:]
            del rows[:]
            model.training_step(batch, 0).backward()
            torch.cuda.synchronize()
            out[off] = (logits, packs, rows[0], n_eval, list(seen))
    finally:
        ops.set_dropout_state(None, 0)
    a, b = out[""], out["1"]
    assert a[3].count("mobgt_stock_front_fwd") == 1 and "mobgt_hop_table_fwd" not in a[3] and "mobgt_stock_tokens_fwd" not in a[3]
    assert a[4].count("mobgt_stock_front_fwd") == 1 and "mobgt_hop_table_fwd" not in a[4] and "mobgt_pack_mfma_b" not in a[4]
    assert "mobgt_stock_front_fwd" not in b[3] + b[4] and "mobgt_stock_tokens_fwd" in b[4] and "mobgt_hop_table_fwd" in b[4]
    assert torch.equal(a[0], b[0])
    assert len(a[1]) == len(b[1]) and all(torch.equal(u, v) for u, v in zip(a[1], b[1]))
    assert torch.equal(a[2], b[2]) and int((a[2] == 0).sum()) > 0          # (train mode: the input dropout's mask is the same)


@pytest.mark.gpu
def test_stock_first_layer_norm_inside_its_qkv_projection(monkeypatch):
    """Round 4: the first pre-LN layer's self_attention_norm (model.py:480) as the prologue of its QKV projection
    (mobgt_ln_gemm_fwd with no residual input) against the two launches (MOBGT_NO_STOCK_LN_QKV=1): the norm's arithmetic is
    the same code, the product is another kernel (other summation order in front of a bf16 rounding): eval logits to 5e-3 on
    logits of magnitude ~1, every gradient to 1e-2 relative L2; the launch is taken once per forward."""
    from mobgt_amd import ops, workloads, synth, fused_layer
    uni, model, coll = workloads.build("fsq", DEV, seed=1, P=1500, variant="stock", model_overrides=dict(n_layers=2))
    batch = coll(synth.make_batch_of_trajectories(seed=21, G=8, P=1500, n_user=1080, cat_of_poi=uni.cat_of_poi,
                                                  n_nodes=[17, 3, 9, 2, 11, 5, 40, 23]))
    model.eval()
    calls = []
    real = fused_layer._ln_gemm_fwd
    monkeypatch.setattr(fused_layer, "_ln_gemm_fwd", lambda *a, **k: (calls.append(a[1] is None), real(*a, **k))[1])
    res = {}
    for off in ("", "1"):
        if off:
            monkeypatch.setenv("MOBGT_NO_STOCK_LN_QKV", "1")
        else:
            monkeypatch.delenv("MOBGT_NO_STOCK_LN_QKV", raising=False)
        del calls[:]
        for p in model.parameters():
            p.grad = None
        logits = model(batch)
        n_calls = list(calls)
        model.training_step(batch, 0).backward()
        torch.cuda.synchronize()
        res[off] = (logits.detach().float().clone(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters()
                                                      if p.grad is not None}, n_calls)
    (la, ga, ca), (lb, gb, cb) = res[""], res["1"]
    assert ca == [True] and cb == [], (ca, cb)
    assert float((la - lb).abs().max()) < 5e-3 and float(lb.abs().max()) > 0.1
    assert ga.keys() == gb.keys()
    for n in ga:
        if n.endswith("linear_k.bias"):
            continue                # exactly 0 in exact arithmetic (softmax is shift-invariant over keys): round-off on both sides
        d = float((ga[n] - gb[n]).norm() / (gb[n].norm() + 1e-30))
        assert d < 1e-2, (n, d)


@pytest.mark.gpu
@pytest.mark.parametrize("C,G,T,p", [(128, 16, 38, 0.1), (128, 3, 130, 0.1), (128, 1, 17, 0.0), (256, 9, 130, 0.1)])
def test_preln_layer_chain_kernel_matches_the_separate_launches(C, G, T, p):
    """Round 4: the chain kernels for graphormer/model.py's PRE-LN EncoderLayer (:463-489, the layer BASELINE.json's north_star
    names; C = 128, d = 16): out-projection -> +residual -> ffn_norm -> FFN -> +residual -> the NEXT layer's
    self_attention_norm -> its QKV projection as one launch per layer, the backward as one launch too (the next layer's
    norm backward is finished in THIS layer's chain launch: dx2 = dout + norm'(dqkv Wqkv)), against the separate launches
    (MOBGT_NO_CHAIN path: 7 + 7 launches per layer) on a 3-layer stock stack with dropout on: same rounding points, same
    masks, so outputs and every gradient agree to bf16 round-off.  Both cluster sizes and the one-workgroup form (R = 608 ->
    4 members, 390 -> 4, 17 -> 4, 1170 -> 2)."""
    from mobgt_amd import fused_layer
    from mobgt_amd.model import EncoderLayer as StockLayer, refresh_shadows
    torch.manual_seed(5)
    H = 8
    layers = torch.nn.ModuleList([StockLayer(C, 1024, p, p, H) for _ in range(3)]).to(DEV)
    for li, l in enumerate(layers):
        l.act_dtype = torch.bfloat16
        l.self_attention.set_layer_index(li + 1)
        l.self_attention.seed_dev = torch.tensor([11], dtype=torch.int64, device=DEV)     # both runs draw the same masks
        with torch.no_grad():                           # (norm weights away from their 1 / 0 initialisation)
            for n in (l.self_attention_norm, l.ffn_norm):
                n.weight.add_(0.2 * torch.randn_like(n.weight))
                n.bias.add_(0.2 * torch.randn_like(n.bias))
    layers.train()
    x0 = torch.randn(G, T, C, device=DEV)
    bias = torch.randn(G, H, T, T, device=DEV) * 0.3
    gy = torch.randn(G, T, C, device=DEV)
    res = {}
    for mode in ("chain", "off"):
        fused_layer._CHAIN[0] = mode != "off"
        try:
            for q in layers.parameters():
                q.grad = None
            x = x0.clone().requires_grad_(True)
            refresh_shadows(layers)
            y = x
            rode = []
            for li, l in enumerate(layers):
                y = l(y, bias, next_layer=layers[li + 1] if li + 1 < len(layers) else None)
                rode.append(bool(getattr(y, "_mobgt_preln", False)))
            assert rode == ([True, True, False] if mode == "chain" else [False, False, False])
            assert bool(getattr(y.grad_fn, "stock_chain", False)) == (mode == "chain")
            y.backward(gy)
            torch.cuda.synchronize()
            res[mode] = (y.detach().clone(), x.grad.clone(), {n: q.grad.clone() for n, q in layers.named_parameters() if q.grad is not None})
        finally:
            fused_layer._CHAIN[0] = True

    def close(a, b, name):
        scale = float(b.abs().max()) + 1e-12
        err = float((a - b).abs().max())
        print("%-44s max|err| %.3e of max %.3e" % (name, err, scale))
        assert err <= 2e-2 * scale, (name, err, scale)
    (ya, dxa, ga), (yb, dxb, gb) = res["chain"], res["off"]
    close(ya, yb, "y")
    close(dxa, dxb, "dx")
    assert ga.keys() == gb.keys() and len(ga) == 3 * 16
    for n in ga:
        if n.endswith("linear_k.bias"):
            continue                                      # exactly zero in exact arithmetic: round-off only
        close(ga[n], gb[n], n)


@pytest.mark.gpu
@pytest.mark.parametrize("R,C,F", [(12560, 256, 1024), (4710, 256, 1024), (13040, 192, 1024), (4411, 128, 1024), (4200, 256, 1024)])
def test_long_batch_weight_gradients_in_one_launch_match_fp32(R, C, F):
    """csrc/wgradbig.hip, mobgt_layer_wgrad_big (round 6): the four weight gradients of an encoder layer past 4 096 rows --
    dWqkv = dqkv^T xa (with dbqkv = column sums of dqkv), dWo = dy^T a, dW1 = du^T z, dW2 = df^T h (autograd of F.linear,
    model.py:393-405, 436-455) -- as ONE launch of 128 x 256 output tiles over S row ranges, against fp32 torch on the same
    bf16 operands.  The products of bf16 values are exact in f32; the sums over up to 13 040 rows differ by summation order only:
    |err| <= 2e-5 x the largest |entry| (measured ~2e-6).  Shapes: S-BIG (all tiles full), 4 710 rows (ragged last chunk: 38
    rows), C = 192 (S-GOW's tail batch: ragged M = 576 / N = 192 tiles), C = 128, and a strided G (dq | dk | dv as one [R, 3C]
    row-major block, which is what the layer hands over)."""
    from mobgt_amd import ops
    torch.manual_seed(R + C)
    bf = dict(dtype=torch.bfloat16, device=DEV)
    dqkv, xa = torch.randn(R, 3 * C, **bf), torch.randn(R, C, **bf)
    dy, a = torch.randn(R, C, **bf), torch.randn(R, C, **bf)
    du, z = torch.randn(R, F, **bf) * 0.5, torch.randn(R, C, **bf)
    df, h = torch.randn(R, C, **bf), torch.randn(R, F, **bf).abs()
    dbqkv = torch.zeros(3 * C, device=DEV)
    items = [(df, h, None, None), (du, z, None, None), (dy, a, None, None), (dqkv, xa, dbqkv, None)]
    assert ops.layer_wgrad_big_ok(items)
    outs = ops.layer_wgrad_big(items, R)
    torch.cuda.synchronize()
    for (g, x, db, _), got in zip(items, outs):
        want = g.float().t() @ x.float()
        assert got.shape == want.shape and got.dtype == torch.float32
        err, top = float((got - want).abs().max()), float(want.abs().max())
        assert err <= 2e-5 * top, (tuple(want.shape), err, top)
    want_db = dqkv.float().sum(0)
    assert float((dbqkv - want_db).abs().max()) <= 2e-5 * float(want_db.abs().max()) + 1e-3
    # column views of a wider row-major block as G (row stride 3C): the q third alone
    part = ops.layer_wgrad_big([(dqkv[:, C:2 * C], xa, None, None)], R)[0]
    want = dqkv[:, C:2 * C].float().t() @ xa.float()
    assert float((part - want).abs().max()) <= 2e-5 * float(want.abs().max())

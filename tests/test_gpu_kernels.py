"""Parity of the HIP kernels (through the C ABI) against the oracle, on a real MI355X.

Tolerances (stated per SURVEY/BASELINE "fp32 tolerance; bit-exact for integer SPD indexing"):
  * integer outputs (spd, path, rel_pos, edge_input, degrees): bit-exact;
  * attention: operands are rounded to bf16 for the MFMA (fp32 accumulate / softmax).  Against an fp32
    oracle fed the SAME bf16-rounded operands: |err| <= 4e-3 * scale-of-output; against the untouched
    fp32 oracle: <= 2e-2 (bf16 has 8 significand bits);
  * bias assembly (fp32 tables, fp32 output): rtol 1e-5 / atol 1e-5.
"""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mobgt_amd import ops, synth            # noqa: E402
from oracle import algos_oracle as ao       # noqa: E402
from oracle import model_oracle as mo       # noqa: E402

DEV = "cuda"


def bf16r(t):
    return t.to(torch.bfloat16).float()


def make_bias(rng, G, H, T, n_real, scale=0.7):
    b = (rng.standard_normal((G, H, T, T)) * scale).astype(np.float32)
    for g in range(G):
        b[g, :, :, n_real[g]:] = -np.inf
    return torch.from_numpy(b)


def ref_attention(q, k, v, bias, H, scale, keep=None, inv_keep=1.0, round_ops=True):
    """model.py:436-455 on [G,T,C] inputs, fp32 CPU; optional bf16 rounding of the MFMA operands."""
    G, T, C = q.shape
    d = C // H
    qs = q * scale
    if round_ops:
        qs, k, v = bf16r(qs), bf16r(k), bf16r(v)
    qh = qs.view(G, T, H, d).transpose(1, 2)
    kh = k.view(G, T, H, d).transpose(1, 2)
    vh = v.view(G, T, H, d).transpose(1, 2)
    s = qh @ kh.transpose(2, 3) + bias
    p = torch.softmax(s, dim=3)
    if keep is not None:
        p = p * keep * inv_keep
    o = p @ vh
    return o.transpose(1, 2).reshape(G, T, C)


CASES = [(2, 8, 1, 16), (2, 8, 5, 16), (3, 8, 33, 16), (2, 8, 64, 24), (2, 8, 65, 24), (1, 8, 130, 32),
         (2, 4, 97, 32), (1, 8, 200, 16)]


@pytest.mark.parametrize("G,H,T,d", CASES)
@pytest.mark.parametrize("bias_dtype", [torch.float32, torch.bfloat16])
def test_attention_fwd_bwd_f32(G, H, T, d, bias_dtype):
    rng = np.random.RandomState(1000 * T + d)
    C = H * d
    n_real = [T] + [max(1, T - 1 - 3 * g) for g in range(1, G)]
    q, k, v, gy = (torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32)) for _ in range(4))
    bias = make_bias(rng, G, H, T, n_real)
    scale = d ** -0.5
    if bias_dtype == torch.bfloat16:
        bias = bf16r(bias)
    # oracle: plain fp32, NO operand rounding -- f32 I/O runs the full-f32 instantiation (csrc/attn_f32_body.h: f32 matrix
    # instruction, f32 softmax) since round 5, so the device is held to fp32 tolerances against model.py:436-455 itself
    qr, kr, vr, br = (t.clone().requires_grad_(True) for t in (q, k, v, bias))
    ref = ref_attention(qr, kr, vr, br, H, scale, round_ops=False)
    ref.backward(gy)
    # device
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    bd = bias.to(DEV).requires_grad_(True)
    pack = ops.pack_bias(bd, G, H, T, dtype=bias_dtype)
    out = ops.attention(qd, kd, vd, pack, scale)
    out.backward(gy.to(DEV))
    torch.cuda.synchronize()
    # packed layouts
    dense = pack.bias[..., :T].float().cpu()
    assert torch.equal(torch.isinf(dense), torch.isinf(bias))
    assert torch.equal(dense[torch.isfinite(bias)], bias[torch.isfinite(bias)])
    assert torch.equal(pack.bias_t[..., :T].float().cpu(), dense.transpose(2, 3))
    assert bool(torch.isinf(pack.bias[..., T:]).all())
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-5, rtol=1e-4)
    gt = 1e-4                                     # (measured 1e-6: summation order)
    for name, got, want in (("dq", qd.grad, qr.grad), ("dk", kd.grad, kr.grad), ("dv", vd.grad, vr.grad)):
        w = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), w, atol=gt * max(1.0, np.abs(w).max()), rtol=gt, err_msg=name)
    db = bd.grad.cpu().numpy()
    if bias_dtype == torch.bfloat16:              # (a bf16 bias gets its gradient as a bf16 slice: dS rounded once, 2^-9)
        np.testing.assert_allclose(db, br.grad.numpy(), atol=1e-4, rtol=8e-3)
    else:
        np.testing.assert_allclose(db, br.grad.numpy(), atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("G,H,T,d", [(2, 8, 33, 16), (2, 8, 70, 24), (1, 8, 130, 32)])
def test_attention_bf16_io(G, H, T, d):
    rng = np.random.RandomState(7 * T + d)
    C = H * d
    q, k, v, gy = (bf16r(torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))) for _ in range(4))
    bias = bf16r(make_bias(rng, G, H, T, [T] * G))
    scale = d ** -0.5
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = ref_attention(qr, kr, vr, bias, H, scale)
    ref.backward(gy)
    qd, kd, vd = (t.to(DEV).to(torch.bfloat16).requires_grad_(True) for t in (q, k, v))
    pack = ops.pack_bias(bias.to(DEV), G, H, T, dtype=torch.bfloat16)
    out = ops.attention(qd, kd, vd, pack, scale)
    assert out.dtype == torch.bfloat16
    out.backward(gy.to(DEV).to(torch.bfloat16))
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref.detach().numpy(), atol=1.2e-2, rtol=1.2e-2)
    for got, want in ((qd.grad, qr.grad), (kd.grad, kr.grad), (vd.grad, vr.grad)):
        w = want.numpy()
        np.testing.assert_allclose(got.float().cpu().numpy(), w, atol=2.5e-2 * max(1.0, np.abs(w).max()), rtol=2.5e-2)


def test_attention_fused_qkv_matches_separate():
    G, H, T, d = 2, 8, 45, 16
    C = H * d
    rng = np.random.RandomState(3)
    qkv = torch.from_numpy(rng.standard_normal((G, T, 3 * C)).astype(np.float32)).to(DEV)
    bias = make_bias(rng, G, H, T, [T, T - 7]).to(DEV)
    gy = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32)).to(DEV)
    pack = ops.pack_bias(bias, G, H, T)
    a = qkv.clone().requires_grad_(True)
    o1 = ops.attention_qkv(a, pack, d ** -0.5)
    o1.backward(gy)
    q, k, v = (qkv[..., i * C:(i + 1) * C].contiguous().requires_grad_(True) for i in range(3))
    o2 = ops.attention(q, k, v, pack, d ** -0.5)
    o2.backward(gy)
    assert torch.equal(o1, o2)
    assert torch.equal(a.grad, torch.cat([q.grad, k.grad, v.grad], dim=2))


def test_attention_dropout_replay():
    """Training-mode attention dropout: the keep mask is a pure function of (seed, g, h, i, j); replay it on
    the host and compare with the oracle using the same mask (SURVEY §5 'Randomness')."""
    G, H, T, d, p = 2, 8, 37, 16, 0.1
    C = H * d
    rng = np.random.RandomState(11)
    q, k, v, gy = (torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32)) for _ in range(4))
    bias = make_bias(rng, G, H, T, [T, T - 5])
    seed = 0x1234567890ABCDEF
    keep = torch.from_numpy(ops.dropout_keep_mask(seed, G, H, T, p)).float()
    thr = int(p * 65536 + 0.5)
    frac = 1.0 - keep.mean().item()
    assert abs(frac - p) < 0.02
    inv_keep = 1.0 / (1.0 - thr / 65536.0)
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = ref_attention(qr, kr, vr, bias, H, d ** -0.5, keep=keep, inv_keep=inv_keep)
    ref.backward(gy)
    qd, kd, vd = (t.to(DEV).requires_grad_(True) for t in (q, k, v))
    pack = ops.pack_bias(bias.to(DEV), G, H, T)
    seed_dev = torch.tensor([5], dtype=torch.int64, device=DEV)
    out = ops.attention(qd, kd, vd, pack, d ** -0.5, p_drop=p, seed=seed - 5, seed_dev=seed_dev)
    out.backward(gy.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=5e-3, rtol=5e-3)
    for got, want in ((qd.grad, qr.grad), (kd.grad, kr.grad), (vd.grad, vr.grad)):
        w = want.numpy()
        np.testing.assert_allclose(got.cpu().numpy(), w, atol=1.5e-2 * max(1.0, np.abs(w).max()), rtol=1.5e-2)


def test_attention_linearity_in_v_large():
    """Size-independent property at the c5 shape (T=785, d=32): attention is linear in V, and rows of P sum
    to 1 (V = ones -> output = ones)."""
    G, H, T, d = 2, 8, 785, 32
    C = H * d
    g = torch.Generator(device="cpu").manual_seed(0)
    q = torch.randn(G, T, C, generator=g).to(DEV)
    k = torch.randn(G, T, C, generator=g).to(DEV)
    v1 = torch.randn(G, T, C, generator=g).to(DEV)
    v2 = torch.randn(G, T, C, generator=g).to(DEV)
    bias = torch.randn(G, H, T, T, generator=g)
    bias[1, :, :, 700:] = float("-inf")
    pack = ops.pack_bias(bias.to(DEV), G, H, T, dtype=torch.bfloat16)
    sc = d ** -0.5
    ones = ops.attention(q, k, torch.ones_like(v1), pack, sc)
    np.testing.assert_allclose(ones.cpu().numpy(), 1.0, atol=4e-3)
    o1, o2 = ops.attention(q, k, v1, pack, sc), ops.attention(q, k, v2, pack, sc)
    o12 = ops.attention(q, k, v1 + v2, pack, sc)
    np.testing.assert_allclose(o12.cpu().numpy(), (o1 + o2).cpu().numpy(), atol=3e-2)


# ----------------------------------------------------------------------------------------- bias assembly
def _g5_batch(z, prefix, fields):
    b = SimpleNamespace()
    for f in fields:
        a = z[f"{prefix}{f}"]
        if f in ("attn_bias", "time_normal"):
            t = torch.from_numpy(a.astype(np.float32))
        elif a.dtype == np.bool_:
            t = torch.from_numpy(a)
        else:
            t = torch.from_numpy(a.astype(np.int64))
        setattr(b, f, t)
    return b


def _seeded(shape, seed, scale=0.08):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32) * scale)


@pytest.mark.parametrize("variant,tag,narrow", [("stock", "stock", False), ("fq", "fsq", False), ("fq", "gow", True)])
def test_build_bias_fwd_bwd(golden_dir, variant, tag, narrow):
    z = np.load(os.path.join(golden_dir, "g5_bias.npz"))
    fields = ["attn_bias", "rel_pos", "edge_input"] + (["poi_pos"] if variant == "fq" else [])
    b = _g5_batch(z, f"{tag}/batch/", fields)
    H, D = 8, 20
    n_edge = 1537 if variant == "stock" else 128
    sd = {"rel_pos_encoder.weight": _seeded((512, H), 1), "edge_encoder.weight": _seeded((n_edge, H), 2),
          "edge_dis_encoder.weight": _seeded((128 * H * H, 1), 3), "graph_token_virtual_distance.weight": _seeded((1, H), 4)}
    if variant == "fq":
        sd["poi_pos_encoder.weight"] = _seeded((int(z[f"{tag}/num_bins"]), H), 5)
    for t in sd.values():
        t.requires_grad_(True)
    ref = mo.assemble_bias(sd, b, H, D, variant)
    gb = _seeded(tuple(ref.shape), 6, 1.0)
    (torch.where(torch.isfinite(ref), ref, torch.zeros_like(ref)) * gb).sum().backward()

    from mobgt_amd.model import hop_table_from, no_grad_row0
    dsd = {k: v.detach().clone().to(DEV).requires_grad_(True) for k, v in sd.items()}
    Dk = min(D, b.edge_input.shape[3])
    hop = hop_table_from(dsd["edge_encoder.weight"], dsd["edge_dis_encoder.weight"], H, Dk, fp16_roundtrip=(variant == "fq"))
    rel_pos, edge_input = b.rel_pos.to(DEV), b.edge_input.to(DEV)
    poi_pos = b.poi_pos.to(DEV) if variant == "fq" else None
    if narrow:
        rel_pos, edge_input, poi_pos = rel_pos.to(torch.int16), edge_input.to(torch.uint8), poi_pos.to(torch.int16)
    poi_tab = no_grad_row0(dsd["poi_pos_encoder.weight"]) if variant == "fq" else None
    pack = ops.build_bias(b.attn_bias.to(DEV), rel_pos, poi_pos, edge_input, no_grad_row0(dsd["rel_pos_encoder.weight"]),
                          poi_tab, hop, dsd["graph_token_virtual_distance.weight"], Dk)
    got = pack.dense().cpu()
    r = ref.detach()
    assert torch.equal(torch.isfinite(got), torch.isfinite(r))
    fin = torch.isfinite(r)
    np.testing.assert_allclose(got[fin].numpy(), r[fin].numpy(), rtol=1e-5, atol=1e-5 if variant == "stock" else 1e-4)
    assert torch.equal(pack.bias_t[..., :pack.T].float().cpu(), got.transpose(2, 3))
    # backward: feed dBias straight into the accumulator the attention layers would have filled
    T = pack.T
    pack.needs_grad = True
    pack.grad_buffer()[..., :T] = gb.to(DEV)
    pack.token.backward()
    for kname in sd:
        want = sd[kname].grad.clone()
        if kname in ("rel_pos_encoder.weight", "edge_encoder.weight", "poi_pos_encoder.weight"):
            want[0] = 0                                  # padding_idx=0 (model.py:63,66)
        gotg = dsd[kname].grad.cpu()
        # fq: the reference back-propagates through .half() casts (fp16 gradients, model_fqandtoyo.py:1178-1198)
        fp16_path = variant == "fq" and kname in ("edge_encoder.weight", "edge_dis_encoder.weight")
        np.testing.assert_allclose(gotg.numpy(), want.numpy(), rtol=2e-3, atol=2e-3 if fp16_path else 2e-5, err_msg=kname)


def test_build_bias_bwd_sums_bf16_layer_slices(golden_dir):
    """bf16 bias: each layer leaves its own bf16 dBias slice and mobgt_build_bias_bwd sums them in f32.  Same
    table gradients as the f32 accumulator holding the sum of those (bf16-rounded) slices."""
    from mobgt_amd.model import hop_table_from, no_grad_row0
    z = np.load(os.path.join(golden_dir, "g5_bias.npz"))
    b = _g5_batch(z, "fsq/batch/", ["attn_bias", "rel_pos", "edge_input", "poi_pos"])
    H, D, L = 8, 20, 3
    sd = {"rel_pos_encoder.weight": _seeded((512, H), 1), "edge_encoder.weight": _seeded((128, H), 2),
          "edge_dis_encoder.weight": _seeded((128 * H * H, 1), 3), "graph_token_virtual_distance.weight": _seeded((1, H), 4),
          "poi_pos_encoder.weight": _seeded((int(z["fsq/num_bins"]), H), 5)}
    G, N = b.rel_pos.shape[:2]
    T = N + 1
    slices = (torch.randn(L, G, H, T, T, generator=torch.Generator().manual_seed(11)) * 0.1).bfloat16()
    Dk = min(D, b.edge_input.shape[3])
    grads = {}
    for dtype in (torch.float32, torch.bfloat16):
        dsd = {k: v.detach().clone().to(DEV).requires_grad_(True) for k, v in sd.items()}
        hop = hop_table_from(dsd["edge_encoder.weight"], dsd["edge_dis_encoder.weight"], H, Dk, fp16_roundtrip=True)
        pack = ops.build_bias(b.attn_bias.to(DEV), b.rel_pos.to(DEV), b.poi_pos.to(DEV), b.edge_input.to(DEV),
                              no_grad_row0(dsd["rel_pos_encoder.weight"]), no_grad_row0(dsd["poi_pos_encoder.weight"]), hop,
                              dsd["graph_token_virtual_distance.weight"], Dk, dtype=dtype)
        pack.needs_grad = True
        if dtype == torch.float32:
            pack.grad_buffer()[..., :T] = slices.float().sum(0).to(DEV)
        else:
            pack.n_use = L
            pack.grad_buffer()[..., :T] = slices.to(DEV)
            pack.n_bwd = L
            assert pack.dbias.shape[0] == L and pack.dbias.dtype == torch.bfloat16
        pack.token.backward()
        grads[dtype] = {k: v.grad.cpu() for k, v in dsd.items() if v.grad is not None}
    assert set(grads[torch.float32]) == set(grads[torch.bfloat16]) and len(grads[torch.float32]) == 5
    for k in grads[torch.float32]:
        np.testing.assert_allclose(grads[torch.bfloat16][k].numpy(), grads[torch.float32][k].numpy(), rtol=1e-4, atol=1e-5,
                                   err_msg=k)


@pytest.mark.parametrize("case", ["lonely", "outlier", "nan"])
def test_build_bias_bwd_fixed_point_fallbacks(case):
    """The rel / poi table gradients are summed in 64-bit fixed point scaled from a strided SAMPLE of the gradient
    (csrc/bias.hip: 4 x 8 elements per thread, 2048 positions in the long-batch form).  What the sample cannot vouch for
    must still come out right: a gradient that is zero wherever the sample looks ('lonely': every add takes the global f32
    path), one unsampled element 1e9 times the rest ('outlier': it alone goes to the global table, the others keep their
    precision), and a NaN (poisons exactly the rows a float sum would)."""
    from mobgt_amd.model import hop_table_from, no_grad_row0
    rs = np.random.RandomState(4)
    G, N, H, D, n_bins = 5, 460, 8, 4, 300
    T = N + 1
    ld = (T + 63) // 64 * 64                            # (ops.PackedBias.ld since round 3)
    total = G * H * T * ld
    sampled = set()
    for k in range(2048):                               # the kernel's sample positions (8 waves x 64 lanes x 4)
        at = (k * total // 2048) & ~7
        sampled.update(range(at, at + 8))
    def free_pair(g, h, i, j):                          # first unsampled live element at or after (g, h, i, j)
        while ((g * H + h) * T + i) * ld + j in sampled:
            j += 1
        return g, h, i, j
    rel = torch.from_numpy(rs.randint(1, 400, size=(G, N, N))).to(DEV)
    poi = torch.from_numpy(rs.randint(1, n_bins, size=(G, N, N))).to(DEV)
    edge = torch.zeros(G, N, N, D, 1, dtype=torch.uint8, device=DEV)
    attn = torch.zeros(G, T, T, device=DEV)
    gen = torch.Generator().manual_seed(5)
    if case == "lonely":
        gb = torch.zeros(G, H, T, T)
        for at, v in ((free_pair(4, 3, 200, 17), 0.37), (free_pair(0, 5, 1, 2), -1.5)):
            gb[at] = v
        special = None
    else:
        gb = torch.randn(G, H, T, T, generator=gen) * 1e-3
        special = free_pair(1, 2, 30, 40)
        gb[special] = 1e6 if case == "outlier" else float("nan")
    sd = {"rel_pos_encoder.weight": _seeded((512, H), 1), "edge_encoder.weight": _seeded((128, H), 2),
          "edge_dis_encoder.weight": _seeded((128 * H * H, 1), 3), "graph_token_virtual_distance.weight": _seeded((1, H), 4),
          "poi_pos_encoder.weight": _seeded((n_bins, H), 5)}
    dsd = {k: v.detach().clone().to(DEV).requires_grad_(True) for k, v in sd.items()}
    hop = hop_table_from(dsd["edge_encoder.weight"], dsd["edge_dis_encoder.weight"], H, D, fp16_roundtrip=True)
    pack = ops.build_bias(attn, rel.to(torch.int16), poi.to(torch.int16), edge, no_grad_row0(dsd["rel_pos_encoder.weight"]),
                          no_grad_row0(dsd["poi_pos_encoder.weight"]), hop, dsd["graph_token_virtual_distance.weight"], D)
    pack.needs_grad = True
    buf = pack.grad_buffer()
    buf.zero_()
    buf[..., :T] = gb.to(DEV)
    pack.token.backward()
    g64 = gb.to(DEV).double()[:, :, 1:, 1:].permute(0, 2, 3, 1).reshape(-1, H)
    for name, idx in (("rel_pos_encoder.weight", rel), ("poi_pos_encoder.weight", poi)):
        want = torch.zeros(sd[name].shape, dtype=torch.float64, device=DEV)
        want.index_add_(0, idx.reshape(-1), g64)
        want = want.float().cpu()
        have = dsd[name].grad.cpu()
        assert torch.equal(torch.isnan(want), torch.isnan(have)), name
        hit = torch.zeros(want.shape[0], dtype=torch.bool)
        if special is not None:
            hit[int(idx[special[0], special[2] - 1, special[3] - 1])] = True       # the row the special element lands in
        if case == "nan":
            assert bool(torch.isnan(have[hit]).any()) and not bool(torch.isnan(have[~hit]).any())
        rows = ~hit
        # every other row: sums of ~2 000 terms of 1e-3 (or the two lonely values), f32-sum accuracy
        np.testing.assert_allclose(have[rows].numpy(), want[rows].numpy(), rtol=2e-5, atol=2e-7, err_msg=name)
        if case == "outlier":
            np.testing.assert_allclose(have[hit].numpy(), want[hit].numpy(), rtol=5e-6, err_msg=name)      # (f32 sums of ~2 000 terms; 1.1e-6 seen)


def test_build_bias_long_batch_form_matches_the_oracle():
    """G x T^2 >= 2^20 pairs selects the 8-wave workgroups of mobgt_build_bias_bwd (one row x 64 columns per wave, hop ids
    read as dwords, [head][row] LDS tables): forward and all five table gradients vs oracle.assemble_bias on synthetic
    trajectories-like indices (narrow device dtypes, bf16 layer slices)."""
    from mobgt_amd.model import hop_table_from, no_grad_row0
    rs = np.random.RandomState(3)
    G, N, H, D, L, n_bins, n_edge = 5, 460, 8, 20, 2, 700, 128
    T = N + 1
    b = SimpleNamespace()
    ii, jj = np.meshgrid(np.arange(N), np.arange(N), indexing="ij")
    spd = np.abs(ii - jj) + 1                                           # path-like graph: SPD = |i - j| (+1: 0 is padding)
    rel = np.broadcast_to(spd, (G, N, N)).copy()
    rel[rs.rand(G, N, N) < 0.2] = 510                                   # unreachable
    n_nodes = [N, N - 37, 300, N, 411]
    edge = rs.randint(1, 6, size=(G, N, N, D))
    edge[rs.rand(G, N, N, D) < 0.02] = rs.randint(16, 40)               # rare ids: the direct-atomic path
    hops = np.minimum(np.where(rel == 510, 0, rel - 1), D)
    edge[np.arange(D)[None, None, None, :] >= hops[..., None]] = 0      # L real hops, then zeros
    poi = rs.randint(1, n_bins, size=(G, N, N))
    attn = np.zeros((G, T, T), np.float32)
    for g, n in enumerate(n_nodes):
        rel[g, n:, :] = 0; rel[g, :, n:] = 0; poi[g, n:, :] = 0; poi[g, :, n:] = 0; edge[g, n:] = 0; edge[g, :, n:] = 0
        attn[g, :, n + 1:] = -np.inf
    b.attn_bias = torch.from_numpy(attn)
    b.rel_pos, b.poi_pos, b.edge_input = (torch.from_numpy(a.astype(np.int64)) for a in (rel, poi, edge[..., None]))
    sd = {"rel_pos_encoder.weight": _seeded((512, H), 1), "edge_encoder.weight": _seeded((n_edge, H), 2),
          "edge_dis_encoder.weight": _seeded((128 * H * H, 1), 3), "graph_token_virtual_distance.weight": _seeded((1, H), 4),
          "poi_pos_encoder.weight": _seeded((n_bins, H), 5)}
    for t in sd.values():
        t.requires_grad_(True)
    ref = mo.assemble_bias(sd, b, H, D, "fq")
    slices = (torch.randn(L, G, H, T, T, generator=torch.Generator().manual_seed(11)) * 0.1).bfloat16()
    gb = slices.float().sum(0)
    (torch.where(torch.isfinite(ref), ref, torch.zeros_like(ref)) * gb).sum().backward()

    dsd = {k: v.detach().clone().to(DEV).requires_grad_(True) for k, v in sd.items()}
    hop = hop_table_from(dsd["edge_encoder.weight"], dsd["edge_dis_encoder.weight"], H, D, fp16_roundtrip=True)
    pack = ops.build_bias(b.attn_bias.to(DEV), b.rel_pos.to(DEV).to(torch.int16), b.poi_pos.to(DEV).to(torch.int16),
                          b.edge_input.to(DEV).to(torch.uint8), no_grad_row0(dsd["rel_pos_encoder.weight"]),
                          no_grad_row0(dsd["poi_pos_encoder.weight"]), hop, dsd["graph_token_virtual_distance.weight"], D,
                          dtype=torch.bfloat16)
    got, r = pack.dense().float().cpu(), ref.detach()
    assert torch.equal(torch.isfinite(got), torch.isfinite(r))
    fin = torch.isfinite(r)
    np.testing.assert_allclose(got[fin].numpy(), r[fin].numpy(), rtol=8e-3, atol=2e-3)          # bf16 storage
    assert torch.equal(pack.bias_t[..., :pack.T].float().cpu(), got.transpose(2, 3))             # 4-round tiles of the long form
    assert bool(torch.isinf(pack.bias_t[..., pack.T:]).all())                                     # padding columns: -inf
    # the f32 output of the same long form (f32 tile in LDS, 3 workgroups per CU): the oracle's values to f32 accuracy, and the
    # bf16 pack above is exactly its rounding
    with torch.no_grad():
        pack32 = ops.build_bias(b.attn_bias.to(DEV), b.rel_pos.to(DEV).to(torch.int16), b.poi_pos.to(DEV).to(torch.int16),
                                b.edge_input.to(DEV).to(torch.uint8), dsd["rel_pos_encoder.weight"], dsd["poi_pos_encoder.weight"],
                                hop.detach(), dsd["graph_token_virtual_distance.weight"], D, dtype=torch.float32)
    got32 = pack32.dense().cpu()
    np.testing.assert_allclose(got32[fin].numpy(), r[fin].numpy(), rtol=1e-4, atol=1e-5)
    assert torch.equal(got32.bfloat16().float()[fin], got[fin])
    assert torch.equal(pack32.bias_t[..., :pack32.T].cpu(), got32.transpose(2, 3))
    pack.needs_grad = True
    pack.n_use = L
    pack.grad_buffer()[..., :T] = slices.to(DEV)
    pack.n_bwd = L
    pack.token.backward()
    for kname in sd:
        want = sd[kname].grad.clone()
        if kname != "graph_token_virtual_distance.weight" and kname != "edge_dis_encoder.weight":
            want[0] = 0
        gotg = dsd[kname].grad.cpu()
        scale = float(want.abs().max())
        # sums of ~10^5..10^6 f32 terms in a different order (+ the reference's fp16 gradients on the edge tables)
        np.testing.assert_allclose(gotg.numpy(), want.numpy(), rtol=3e-3, atol=2e-3 * max(scale, 1.0), err_msg=kname)


# ------------------------------------------------------------------------------------------------ spd
def _spd_case(counts_list, D=20):
    G = len(counts_list)
    N = max(c.shape[0] for c in counts_list)
    counts = np.zeros((G, N, N), np.int32)
    n_nodes = np.zeros(G, np.int32)
    for g, c in enumerate(counts_list):
        n = c.shape[0]
        counts[g, :n, :n] = c
        n_nodes[g] = n
    out = ops.spd_batched(torch.from_numpy(counts).to(DEV), torch.from_numpy(n_nodes).to(DEV), D)
    return {k: v.cpu().numpy() for k, v in out.items()}, N


def _check_spd(counts_list, D=20):
    out, N = _spd_case(counts_list, D)
    for g, c in enumerate(counts_list):
        n = c.shape[0]
        adj = c != 0
        M, path = ao.floyd_warshall(adj)
        assert np.array_equal(out["spd"][g, :n, :n], M), g
        assert np.array_equal(out["path"][g, :n, :n], path), g
        assert (out["spd"][g, n:, :] == -1).all() and (out["spd"][g, :, n:] == -1).all()
        rp = np.zeros((N, N), np.int64)
        rp[:n, :n] = M + 1
        assert np.array_equal(out["rel_pos"][g], rp)
        feat = np.zeros((n, n, 1), np.int64)
        feat[adj, 0] = c[adj] + 2
        md = int(M.max())
        ei = ao.gen_edge_input(md, path, feat).astype(np.int64) if md > 0 else np.zeros((n, n, 0, 1), np.int64)
        want = np.zeros((N, N, D, 1), np.int64)
        dd = min(D, ei.shape[2])
        want[:n, :n, :dd] = ei[:, :, :dd] + 1
        assert np.array_equal(out["edge_input"][g].astype(np.int64), want), g
        ind = np.zeros(N, np.int64)
        outd = np.zeros(N, np.int64)
        ind[:n] = adj.sum(1) + 1
        outd[:n] = adj.sum(0) + 1
        assert np.array_equal(out["in_degree"][g], ind) and np.array_equal(out["out_degree"][g], outd)


def test_spd_golden_graphs(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_algos.npz"))
    graphs = [z[f"{n}/counts"].astype(np.int64) for n in z["names"] if n != "cycle600"]
    _check_spd(graphs)
    # and directly against the reference's own outputs
    out, N = _spd_case(graphs)
    for g, name in enumerate([n for n in z["names"] if n != "cycle600"]):
        n = graphs[g].shape[0]
        assert np.array_equal(out["spd"][g, :n, :n], z[f"{name}/M"])
        assert np.array_equal(out["path"][g, :n, :n], z[f"{name}/path"])
        ref = z[f"{name}/edge_input20"].astype(np.int64) + 1
        dd = ref.shape[2]
        assert np.array_equal(out["edge_input"][g, :n, :n, :dd].astype(np.int64), ref)


def test_spd_random_and_large(golden_dir):
    rng = np.random.RandomState(5)
    graphs = [synth.random_digraph(rng, n, p) for n, p in ((1, 0.5), (2, 0.5), (9, 0.3), (31, 0.1), (64, 0.05), (100, 0.03))]
    graphs += [synth.make_trajectory(rng, 2000, n, 4)["edge_type"] for n in (3, 50, 120)]
    _check_spd(graphs)
    _check_spd(graphs, D=5)
    # global-memory path (N > 272) and the N > 510 sentinel case
    big = [synth.make_trajectory(rng, 2000, 300, 4)["edge_type"], synth.random_digraph(rng, 290, 0.01)]
    _check_spd(big)
    # multi-workgroup path: more graphs than one launch holds resident (chunk loop), ragged sizes down to n = 1
    # (workgroups without rows), a graph count that is not a multiple of 8 (plain block -> graph mapping)
    many = [synth.make_trajectory(rng, 2000, 280, 3)["edge_type"]]
    many += [synth.random_digraph(rng, int(n), 0.08) for n in rng.randint(1, 60, size=34)]
    _check_spd(many)
    z = np.load(os.path.join(golden_dir, "g1_algos.npz"))
    c = z["cycle600/counts"].astype(np.int64)
    out, _ = _spd_case([c])
    assert np.array_equal(out["spd"][0], z["cycle600/M"])
    assert np.array_equal(out["path"][0], z["cycle600/path"])


def test_split_floyd_warshall_gives_up_cleanly_and_survives_contention():
    """the 16 workgroups of a long graph wait for each other.  (a) With the wait bound forced to 0 every
    waiter gives up at once and the single-workgroup redo pass must deliver the same bit-exact result; (b) with the
    default bound, a second stream saturating the chip with large GEMMs must neither hang the launch nor change it."""
    from mobgt_amd import _lib
    rng = np.random.RandomState(12)
    graphs = [synth.make_trajectory(rng, 2000, 400, 3)["edge_type"], synth.random_digraph(rng, 310, 0.01),
              synth.make_trajectory(rng, 2000, 64, 3)["edge_type"]]
    lib = _lib.lib()
    try:
        _lib.check(lib.mobgt_spd_set_spin_limit(0), "mobgt_spd_set_spin_limit")
        _check_spd(graphs)
    finally:
        _lib.check(lib.mobgt_spd_set_spin_limit(-1), "mobgt_spd_set_spin_limit")
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    with torch.cuda.stream(side):
        for _ in range(40):
            a @ a
    _check_spd(graphs)                                   # runs while the side stream is busy
    torch.cuda.synchronize()
    _check_spd(graphs)


# ---------------------------------------------------------------------------------------------- embed
def test_embed_gather_sum_and_grad():
    rng = np.random.RandomState(2)
    C = 192
    t1 = torch.from_numpy(rng.standard_normal((50, C)).astype(np.float32))
    t2 = torch.from_numpy(rng.standard_normal((128, C)).astype(np.float32))
    t3 = torch.from_numpy(rng.standard_normal((128, C)).astype(np.float32))
    i1 = torch.from_numpy(rng.randint(-1, 50, size=(4, 9)))
    i2 = torch.from_numpy(rng.randint(0, 128, size=(4, 9)))
    i3 = torch.from_numpy(rng.randint(0, 128, size=(4, 9)))
    gy = torch.from_numpy(rng.standard_normal((4, 9, C)).astype(np.float32))
    a, b, c = (t.clone().requires_grad_(True) for t in (t1, t2, t3))
    ref = torch.where((i1 >= 0).unsqueeze(-1), a[i1.clamp(min=0)], torch.zeros(())) + b[i2] + c[i3]
    ref.backward(gy)
    ad, bd, cd = (t.to(DEV).requires_grad_(True) for t in (t1, t2, t3))
    out = ops.embed_gather_sum([ad, bd, cd], [i1.to(DEV), i2.to(DEV), i3.to(DEV)], padding_idx=[None, 0, 0])
    out.backward(gy.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ad.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-5, atol=1e-5)
    for got, want in ((bd.grad, b.grad), (cd.grad, c.grad)):
        w = want.clone()
        w[0] = 0
        np.testing.assert_allclose(got.cpu().numpy(), w.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C,R,idt", [(192, 4999, torch.int64), (256, 4096, torch.int32), (448, 5003, torch.int64), (40, 4100, torch.int16)])
def test_embed_scatter_of_long_batches_run_length_form(C, R, idt):
    """mobgt_embed_scatter_add from 4 096 rows (S-BIG: 12 560): a wave keeps the sum of a run of equal indices in registers
    (csrc/embed.hip scatter_add_runs_kernel; round 4: the wave's 16 gradient rows and indices are read once, all in flight).
    Long stretches of one index, runs that cross the 16-row groups, padding rows (no gradient), negative indices (contribute
    nothing), a ragged last group; both register widths (C <= 256 / <= 512).  Against index_add_ in float64."""
    rng = np.random.RandomState(C + R)
    sizes = (300, 64, 64)
    tabs = [torch.from_numpy(rng.standard_normal((n, C)).astype(np.float32)) for n in sizes]
    idx = []
    for n in sizes:
        i = rng.randint(0, n, size=R)
        runs = rng.randint(0, R - 40, size=60)
        for r0 in runs:
            i[r0:r0 + rng.randint(2, 40)] = rng.randint(0, n)          # stretches of one row, any alignment
        idx.append(i)
    idx[0][rng.randint(0, R, size=50)] = -1
    gy = torch.from_numpy(rng.standard_normal((R, C)).astype(np.float32))
    dev_t = [t.to(DEV).requires_grad_(True) for t in tabs]
    dev_i = [torch.from_numpy(i).to(idt).to(DEV) for i in idx]
    out = ops.embed_gather_sum(dev_t, dev_i, padding_idx=[None, 0, 0])
    out.backward(gy.to(DEV))
    for t, (n, i, pad) in enumerate(zip(sizes, idx, (None, 0, 0))):
        want = torch.zeros(n, C, dtype=torch.float64)
        keep = torch.from_numpy(i >= 0)
        want.index_add_(0, torch.from_numpy(i)[keep].long(), gy.double()[keep])
        if pad is not None:
            want[pad] = 0
        np.testing.assert_allclose(dev_t[t].grad.double().cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-5 * float(want.abs().max()), err_msg=str(t))


def test_partial_sum_multi_kernel_matches_sum_over_slices():
    """mobgt_partial_sum_multi (round 4): dst_i = sum over the s_i slices of src_i, n jobs of different sizes in one launch;
    against a float64 sum (and bit-equal to a sequential f32 sum in slice order)."""
    import ctypes
    from mobgt_amd import _lib
    gen = torch.Generator().manual_seed(3)
    jobs = [(1, 64), (3, 1028), (4, 256 * 1024), (7, 4100), (16, 768 * 256), (5, 4)]
    srcs = [torch.randn(s, n, generator=gen).to(DEV) for s, n in jobs]
    dsts = [torch.full((n,), float("nan"), device=DEV) for _, n in jobs]
    k = len(jobs)
    vp = ctypes.c_void_p
    _lib.check(_lib.lib().mobgt_partial_sum_multi(k, (vp * k)(*[t.data_ptr() for t in srcs]), (vp * k)(*[t.data_ptr() for t in dsts]),
                                                  (ctypes.c_int * k)(*[s for s, _ in jobs]), (ctypes.c_int64 * k)(*[n for _, n in jobs]),
                                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mobgt_partial_sum_multi")
    torch.cuda.synchronize()
    for (s_, n), src, dst in zip(jobs, srcs, dsts):
        seq = torch.zeros(n, device=DEV)
        for i in range(s_):
            seq = seq + src[i]
        assert torch.equal(dst, seq), (s_, n)
        np.testing.assert_allclose(dst.double().cpu().numpy(), src.double().sum(0).cpu().numpy(), rtol=0, atol=1e-5)
    # a slice length that is not a multiple of 4 is refused
    assert _lib.lib().mobgt_partial_sum_multi(1, (vp * 1)(srcs[0].data_ptr()), (vp * 1)(dsts[0].data_ptr()), (ctypes.c_int * 1)(1),
                                              (ctypes.c_int64 * 1)(6), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0


@pytest.mark.parametrize("all_hip", [False, True])
@pytest.mark.parametrize("G,K,V", [(16, 448, 7857), (5, 448, 1030), (1, 64, 2048), (16, 512, 1024), (16, 448, 40001)])
def test_skinny_linear_matches_torch(G, K, V, all_hip, monkeypatch):
    """mobgt_skinny_linear_fwd/bwd (the classifier head, M = G rows) against F.linear in fp64; the product path uses
    the dW/db kernel only, MOBGT_SKINNY_ALL=1 exercises the forward and dx kernels as well."""
    if all_hip:
        monkeypatch.setenv("MOBGT_SKINNY_ALL", "1")
    else:
        monkeypatch.delenv("MOBGT_SKINNY_ALL", raising=False)
    gen = torch.Generator().manual_seed(G + K + V)
    x = torch.randn(G, K, generator=gen)
    w = torch.randn(V, K, generator=gen) * 0.05
    b = torch.randn(V, generator=gen)
    gy = torch.randn(G, V, generator=gen)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    torch.nn.functional.linear(xr, wr, br).backward(gy.double())
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    assert ops.skinny_linear_ok(xd, wd)
    y = ops.skinny_linear(xd, wd, bd)
    y.backward(gy.to(DEV))
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(wd.grad.cpu().numpy(), wr.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), br.grad.numpy(), rtol=1e-4, atol=1e-4)


def test_gather_rows_t():
    gen = torch.Generator().manual_seed(4)
    a = torch.randn(300, 136, generator=gen).bfloat16().to(DEV)
    rows = torch.randint(0, 300, (72,), generator=gen).to(DEV)
    r, rt = ops.gather_rows_t(a, rows)
    assert torch.equal(r, a[rows]) and torch.equal(rt, a[rows].t().contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(192, 192), (1024, 192), (192, 1024), (576, 192), (32, 64)])
def test_pack_mfma_b_layout_is_bit_exact(N, K):
    """mobgt_pack_mfma_b: the 16 bytes W[16g + j][32s + 8q .. +7] land at ((g K/32 + s) 64 + j + 16q) * 16 bytes -- for the
    weight itself and for the transpose of a [K,N] weight -- bit for bit (the chain kernels' B operand order)."""
    import ctypes
    from mobgt_amd import _lib
    from mobgt_amd.ops import _stream
    torch.manual_seed(N + K)
    w = torch.randn(N, K, device=DEV).bfloat16()                       # [N,K]: packed as it is
    wt_src = w.t().contiguous()                                          # [K,N]: packed transposed -> the same operand
    outs = [torch.empty(N * K, dtype=torch.bfloat16, device=DEV) for _ in range(2)]
    vp, ci = ctypes.c_void_p, ctypes.c_int
    _lib.check(_lib.lib().mobgt_pack_mfma_b(2, (vp * 2)(w.data_ptr(), wt_src.data_ptr()), (vp * 2)(outs[0].data_ptr(), outs[1].data_ptr()),
                                            (ci * 2)(N, N), (ci * 2)(K, K), (ci * 2)(0, 1), _stream()), "mobgt_pack_mfma_b")
    torch.cuda.synchronize()
    S = K // 32
    g, s, q, j, e = np.meshgrid(np.arange(N // 16), np.arange(S), np.arange(4), np.arange(16), np.arange(8), indexing="ij")
    want = w.cpu().view(torch.int16).numpy()[16 * g + j, 32 * s + 8 * q + e].reshape(-1)     # order (g, s, q, j, e) = ((g S + s) 64 + 16 q + j) 8 + e
    for o in outs:
        np.testing.assert_array_equal(o.cpu().view(torch.int16).numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_embed_gather_multi_matches_torch_indexing(idt):
    """ops.embed_gather_multi: concatenated and summed gathers of one position list in one launch (zeros for negative
    indices, `accum` jobs added on top), and the scatter-add backward that skips padding rows -- against torch indexing."""
    from mobgt_amd import ops
    torch.manual_seed(3)
    R = 301
    tabs = [torch.randn(n, w, device=DEV, requires_grad=True) for n, w in ((50, 128), (49, 32), (30, 32), (10, 192), (128, 192), (2000, 192))]
    idx = [torch.randint(-1 if t in (0, 5) else 0, tab.shape[0], (R,), device=DEV).to(idt) for t, tab in enumerate(tabs)]
    pad = [None, 0, None, 0, 0, None]
    jobs = [(tabs[0], idx[0], 0, 0, False, pad[0]), (tabs[1], idx[1], 0, 128, False, pad[1]), (tabs[2], idx[2], 1, 160, False, pad[2]),
            (tabs[3], idx[3], 2, 0, False, pad[3]), (tabs[4], idx[4], 2, 0, True, pad[4]), (tabs[5], idx[5], 2, 0, True, pad[5])]
    pt, x4, add = ops.embed_gather_multi(jobs, [160, 192, 192])

    def take(t):
        i = idx[t].long()
        return torch.where((i >= 0)[:, None], tabs[t][i.clamp(min=0)], torch.zeros((), device=DEV))
    torch.testing.assert_close(pt, torch.cat((take(0), take(1)), 1), rtol=0, atol=0)
    torch.testing.assert_close(x4[:, 160:], take(2), rtol=0, atol=0)
    torch.testing.assert_close(add, take(3) + take(4) + take(5), rtol=1e-6, atol=1e-6)
    g_pt, g_x4, g_add = torch.randn_like(pt), torch.randn_like(x4), torch.randn_like(add)
    torch.autograd.backward([pt, x4, add], [g_pt, g_x4, g_add])
    srcs = [g_pt[:, :128], g_pt[:, 128:], g_x4[:, 160:], g_add, g_add, g_add]
    for t, tab in enumerate(tabs):
        i = idx[t].long()
        keep = (i >= 0) & ((i != pad[t]) if pad[t] is not None else torch.ones_like(i, dtype=torch.bool))
        want = torch.zeros_like(tab).index_add_(0, i[keep], srcs[t][keep])
        torch.testing.assert_close(tab.grad, want, rtol=1e-5, atol=1e-5)

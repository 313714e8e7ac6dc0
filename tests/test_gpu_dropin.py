"""The drop-in route of INTEGRATION.md on the MI355X: the reference's own model code keeps calling
`EncoderLayer(x, attn_bias, mask)` with a DENSE bias tensor it builds per batch (model.py:190).

* two batches in a row under `no_grad`, each with a freshly created (and then freed) dense bias of the same shape: the
  second batch must attend with ITS bias (the pack cache is keyed on tensor identity, not on the address the caching
  allocator recycles);
* the `mask` argument (model.py:446-448: masked scores := 0) against the reference's outputs (golden G9).

Tolerances as tests/test_gpu_layer.py: bf16 MFMA operands in the attention core -> 2e-2 relative to the output scale; the
masked branch runs its dense fp32 form -> 1e-4."""
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from inputs import MASK_CASES, mask_case, encoder_case, rand_bias     # noqa: E402
from test_gpu_layer import build_layer                                # noqa: E402
from test_oracle_model import encoder_param_list, seeded_state       # noqa: E402
from oracle import model_oracle as mo                                 # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("variant", ["stock", "fq"])
@pytest.mark.parametrize("fused", [True, False])
def test_two_dense_biases_back_to_back_under_no_grad(variant, fused):
    C, T, G, ffn = 128, 33, 2, 1024
    seed, x, _, _, n_real = encoder_case(variant, C, T, G)
    layers = [build_layer(variant, C, ffn, seed + 1 + i) for i in range(2)]      # two layers share the batch's bias
    for l in layers:
        l.fused = fused
    sds = [seeded_state([("L." + n, s) for n, s in encoder_param_list(variant, C, ffn)], seed + 1 + i) for i in range(2)]
    fn = mo.encoder_layer_stock if variant == "stock" else mo.encoder_layer_fq
    rng = np.random.RandomState(3)
    ptrs = []
    for batch in range(3):
        bias_np = rand_bias(rng, G, 8, T, n_real) * (1.0 + 2.0 * batch)             # clearly different values per batch
        with torch.no_grad():
            ref = torch.from_numpy(x)
            for sd in sds:
                ref = fn(sd, "L", ref, torch.from_numpy(bias_np), 8)
            xd = torch.from_numpy(x).to(DEV)
            bd = torch.from_numpy(bias_np).to(DEV) + 0.0          # a fresh tensor per batch, version 0 (model.py:190)
            ptrs.append(bd.data_ptr())
            y = xd
            for l in layers:
                y = l(y, bd, mask=None)
            torch.cuda.synchronize()
            got = y.cpu().numpy()
            del bd, y
            gc.collect()
        r = ref.numpy()
        np.testing.assert_allclose(got, r, atol=2e-2 * max(1.0, np.abs(r).max()), rtol=2e-2, err_msg=f"batch {batch}")
    # (whether the allocator handed a freed bias's address to a later batch depends on its state -- it does in a fresh process,
    #  which is how round 4's cache, keyed on (address, version, shape), went wrong; the test below forces the collision)
    print("bias addresses", [hex(p) for p in ptrs])


def test_pack_cache_is_keyed_on_the_tensor_object_not_on_its_address(monkeypatch):
    """The collision itself, without the allocator's cooperation: two DIFFERENT tensor objects with the same address, version,
    shape and dtype (the second is a fresh view of the first one's memory, refilled through another alias) must be packed
    twice; the same object is packed once for all layers of a forward."""
    from mobgt_amd import ops
    from mobgt_amd.model import MultiHeadAttention
    calls = []
    real = ops.pack_bias
    monkeypatch.setattr(ops, "pack_bias", lambda *a, **k: (calls.append(a[0].data_ptr()), real(*a, **k))[1])
    G, H, T = 2, 8, 5
    mha = MultiHeadAttention(128, 0.0, H).to(DEV).eval()
    x = torch.randn(G, T, 128, device=DEV)
    store = torch.randn(G * H * T * T, device=DEV)
    b1 = store.view(G, H, T, T)
    with torch.no_grad():
        y1 = mha(x, x, x, b1)
        y1b = mha(x, x, x, b1)                       # the same object again (a second layer): the cached pack
        assert len(calls) == 1 and torch.equal(y1, y1b)
        key = (b1.data_ptr(), b1._version, tuple(b1.shape))
        del b1
        gc.collect()
        store.detach().view(-1).mul_(-3.0)           # other values at the same address ...
        b2 = store.detach().view(G, H, T, T)         # ... behind a NEW tensor object
        if (b2.data_ptr(), b2._version, tuple(b2.shape)) != key:
            b2 = torch.empty(0, device=DEV).set_(store.untyped_storage(), 0, (G, H, T, T))
        assert b2.data_ptr() == key[0] and tuple(b2.shape) == key[2]
        y2 = mha(x, x, x, b2)
        assert len(calls) == 2, calls
        ref = mha(x, x, x, b2.clone())
        assert len(calls) == 3
        torch.testing.assert_close(y2, ref, rtol=0, atol=0)
        assert not torch.allclose(y2, y1)


def test_same_bias_tensor_trained_twice_repacks_after_its_gradient_was_taken():
    """A dense bias that requires grad, used for two forward / backward rounds: the second round's gradient is the second
    round's alone (the pack of round one is spent once its gradient has been handed out)."""
    C, T, G, ffn = 128, 5, 3, 1024
    seed, x, bias, gy, _ = encoder_case("fq", C, T, G)
    layer = build_layer("fq", C, ffn, seed + 1)
    bd = torch.from_numpy(bias).to(DEV).requires_grad_(True)
    grads = []
    for _ in range(2):
        xd = torch.from_numpy(x).to(DEV).requires_grad_(True)
        layer(xd, bd, mask=None).backward(torch.from_numpy(gy).to(DEV))
        grads.append(bd.grad.clone())
        bd.grad = None
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("variant", ["stock", "fq"])
def test_encoder_layer_with_mask_matches_reference_g9(golden_dir, variant):
    z = np.load(os.path.join(golden_dir, "g9_mask.npz"))
    for cname, C, T, G, ffn in MASK_CASES:
        name = f"{variant}/{cname}"
        seed, x, bias, gy, _, mask = mask_case(variant, C, T, G)
        layer = build_layer(variant, C, ffn, seed + 1)
        xd = torch.from_numpy(x).to(DEV).requires_grad_(True)
        bd = torch.from_numpy(bias).to(DEV).requires_grad_(True)
        y = layer(xd, bd, mask=torch.from_numpy(mask).to(DEV))
        y.backward(torch.from_numpy(gy).to(DEV))
        torch.cuda.synchronize()
        for got, key in ((y.detach(), "y"), (xd.grad, "dx"), (bd.grad, "dbias")):
            ref = z[f"{name}/{key}"]
            np.testing.assert_allclose(got.cpu().numpy(), ref, atol=1e-4 * max(1.0, np.abs(ref).max()), rtol=1e-4, err_msg=f"{name}/{key}")
        for pn, p in layer.named_parameters():
            if f"{name}/gstat/{pn}" in z:
                rn = float(z[f"{name}/gstat/{pn}"][1])
                assert np.isclose(p.grad.double().norm().item(), rn, rtol=1e-3, atol=1e-5), pn


def test_mask_with_a_packed_bias_and_without_a_bias():
    """MultiHeadAttention(q, k, v, attn_bias, mask) stand-alone: PackedBias input and attn_bias=None, against the oracle."""
    from mobgt_amd.model import MultiHeadAttention
    from mobgt_amd import ops
    C, T, G = 128, 9, 2
    rng = np.random.RandomState(5)
    mha = MultiHeadAttention(C, 0.1, 8).to(DEV).eval()
    sd = {"A." + k: v.detach().cpu() for k, v in mha.state_dict().items()}
    x = torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32))
    bias = torch.from_numpy(rand_bias(rng, G, 8, T, [T, T - 2]))
    mask = torch.from_numpy(rng.rand(G, T, T) < 0.3)
    with torch.no_grad():
        for b_cpu, b_dev in ((None, None), (bias, ops.pack_bias(bias.to(DEV), G, 8, T))):
            ref = mo.multi_head_attention(sd, "A", x, x, x, b_cpu, 8, mask=mask)
            got = mha(x.to(DEV), x.to(DEV), x.to(DEV), b_dev, mask=mask.to(DEV))
            np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("variant", ["stock", "fq"])
def test_writes_through_dot_data_reach_the_next_forward(variant):
    """the bf16 shadow weights / MFMA packs of a fused layer were re-derived only when a parameter's version
    counter moved -- `p.data.mul_()` / `p.data.copy_()` (the reference's own `init_params`, model.py:30-34; many optimizers) do
    not move it, and the next forward ran on stale bf16 weights.  Now the copy is unconditional unless a backward that saved the
    shadows is pending or the owner set `layer._weights_frozen`: a `.data` write between two forwards must show in the second
    one exactly as in a layer built with the new weights; with the opt-in flag set the old weights stay (documented)."""
    import copy
    C, T, G, ffn = 192 if variant == "fq" else 128, 33, 2, 1024
    seed, x, _, _, n_real = encoder_case(variant, C, T, G)
    layer = build_layer(variant, C, ffn, seed + 1)
    layer.act_dtype = torch.bfloat16
    layer.eval()
    rng = np.random.RandomState(9)
    bd = torch.from_numpy(rand_bias(rng, G, 8, T, n_real)).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    with torch.no_grad():
        y0 = layer(xd, bd).float().clone()
        for p in (layer.ffn.layer2.weight, layer.self_attention.linear_v.weight, layer.self_attention.output_layer.bias):
            v0 = p._version
            p.data.mul_(0.5)                            # version counter untouched
            assert p._version == v0
        y1 = layer(xd, bd).float().clone()
        fresh = copy.deepcopy(layer)
        for a in ("_shadows", "_packed", "_packed_t", "_shadow_ver", "_packed_ver"):
            fresh.__dict__.pop(a, None)
        fresh.self_attention.__dict__.pop("_wqkv", None)
        y_ref = fresh(xd, bd).float()
        assert float((y1 - y0).abs().max()) > 1e-2     # the write matters ...
        assert torch.equal(y1, y_ref)                   # ... and the forward after it is the forward of the new weights
        layer._weights_frozen = True                    # opt-in: the owner vouches that nothing writes the weights
        layer.ffn.layer2.weight.data.mul_(2.0)
        y2 = layer(xd, bd).float()
        assert torch.equal(y2, y1)
        layer._weights_frozen = False
        y3 = layer(xd, bd).float()
        assert float((y3 - y1).abs().max()) > 1e-2

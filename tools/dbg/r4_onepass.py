"""GPU box: the one-pass attention backward (csrc/attn.hip: attn_bwd_one_kernel) against the two passes: gradients at a few
shapes, then the all-cold timing of bench.time_attention for both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mobgt_amd import ops
import bench

def grads(G, H, T, d, p, one):
    ops._ATTN_ONE_PASS[0] = one
    C = H * d
    g = torch.Generator().manual_seed(T + d)
    q, k, v, do = (torch.randn(G, T, C, generator=g).cuda().bfloat16() for _ in range(4))
    bias = torch.randn(G, H, T, T, generator=g).cuda()
    bias[G - 1, :, :, T - 37:] = float("-inf")
    pack = ops.pack_bias(bias, G, H, T, dtype=torch.bfloat16)
    pack.needs_grad, pack.n_use = True, 1
    out, lse = ops._attn_fwd(q, k, v, pack, d ** -0.5, p, 7, None)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    pack.n_bwd = 0
    ops._attn_bwd(q, k, v, out, lse, do, dq, dk, dv, pack, d ** -0.5, p, 7, None)
    torch.cuda.synchronize()
    return dq.float(), dk.float(), dv.float(), pack.dbias[0, ..., :T].float()

for (G, H, T, d) in ((2, 8, 785, 32), (3, 8, 130, 24), (2, 8, 815, 24), (2, 8, 300, 16), (1, 8, 65, 32)):
    for p in (0.0, 0.1):
        a, b = grads(G, H, T, d, p, True), grads(G, H, T, d, p, False)
        msg = []
        for name, x, y in zip(("dq", "dk", "dv", "dbias"), a, b):
            rel = float((x - y).norm() / y.norm())
            mx = float((x - y).abs().max() / y.abs().max())
            msg.append("%s relL2 %.2e max %.2e" % (name, rel, mx))
            assert torch.isfinite(x).all()
        print("T %d d %d p %.1f: " % (T, d, p) + " | ".join(msg))
for one in (True, False, True, False):
    ops._ATTN_ONE_PASS[0] = one
    f, b, n = bench.time_attention(16, 8, 785, 32, torch.bfloat16, torch.bfloat16, reps=24, p_drop=0.1, backward=True)
    print("c5 all-cold: %s fwd %.1f us bwd %.1f us" % ("one-pass" if one else "two-pass", f * 1e6, b * 1e6))
ops._ATTN_ONE_PASS[0] = True
f, b, n = bench.time_attention(16, 8, 815, 24, torch.bfloat16, torch.bfloat16, reps=24, p_drop=0.1, backward=True)
print("T815 d24 all-cold one-pass: fwd %.1f bwd %.1f" % (f * 1e6, b * 1e6))
ops._ATTN_ONE_PASS[0] = False
f, b, n = bench.time_attention(16, 8, 815, 24, torch.bfloat16, torch.bfloat16, reps=24, p_drop=0.1, backward=True)
print("T815 d24 all-cold two-pass: fwd %.1f bwd %.1f" % (f * 1e6, b * 1e6))

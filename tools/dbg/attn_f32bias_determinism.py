"""Developer tool (GPU box): is the attention backward with an f32 / bf16 packed bias deterministic run to run?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mobgt_amd import ops
G, H, T, d = 4, 8, 53, 24
C = H * d
rng = np.random.RandomState(1)
for io_dt, bias_dt in ((torch.float32, torch.float32), (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16)):
    q, k, v, gy = (torch.from_numpy(rng.standard_normal((G, T, C)).astype(np.float32)).cuda().to(io_dt) for _ in range(4))
    bias = torch.from_numpy((rng.standard_normal((G, H, T, T)) * 0.5).astype(np.float32)).cuda()
    bias[1, :, :, 40:] = float("-inf")
    bias[3, :, :, 7:] = float("-inf")
    seed_dev = torch.tensor([11], dtype=torch.int64, device="cuda")
    first = None
    worst = 0.0
    nbad = 0
    for it in range(60):
        qd, kd, vd = (t.clone().requires_grad_(True) for t in (q, k, v))
        bd = bias.clone().requires_grad_(True)
        pack = ops.pack_bias(bd, G, H, T, dtype=bias_dt)
        out = ops.attention(qd, kd, vd, pack, d ** -0.5, p_drop=0.1, seed=5, seed_dev=seed_dev)
        out.backward(gy)
        if it % 7 == 3:
            junk = torch.randn(1 << 22, device="cuda").sum()      # (perturb timing / caches)
        torch.cuda.synchronize()
        cur = [t.detach().float().clone() for t in (out, qd.grad, kd.grad, vd.grad, bd.grad)]
        if first is None:
            first = cur
        else:
            for n, a, b in zip(("out", "dq", "dk", "dv", "dbias"), first, cur):
                if not torch.equal(a, b):
                    nbad += 1
                    dmax = float((a - b).abs().max())
                    worst = max(worst, dmax)
                    if nbad <= 5:
                        idx = torch.nonzero(a != b)
                        print("  run %d: %s differs at %d elements, max %.3e, first idx %s" % (it, n, idx.shape[0], dmax, idx[0].tolist()))
    print("io %s bias %s: %d mismatching tensors over 59 reruns, worst %.3e" % (io_dt, bias_dt, nbad, worst))

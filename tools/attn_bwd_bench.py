"""Developer tool: run the attention forward/backward at the c5 shape (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import ops

G, H, T, d = 16, 8, int(os.environ.get("T", 785)), int(os.environ.get("D", 32))
p_drop = float(os.environ.get("P", 0.1))
dt = torch.bfloat16
C = H * d
g = torch.Generator().manual_seed(0)
q, k, v, do = (torch.randn(G, T, C, generator=g).cuda().to(dt) for _ in range(4))
bias = torch.randn(G, H, T, T, generator=g).cuda()
pack = ops.pack_bias(bias, G, H, T, dtype=dt)
pack.needs_grad = os.environ.get('NODBIAS') != '1'
dq, dk, dv = (torch.empty_like(q) for _ in range(3))
for _ in range(int(os.environ.get("REPS", 10))):
    out, lse = ops._attn_fwd(q, k, v, pack, d ** -0.5, p_drop, 1, None)
    pack.n_bwd = 0
    ops._attn_bwd(q, k, v, out, lse, do, dq, dk, dv, pack, d ** -0.5, p_drop, 1, None)
torch.cuda.synchronize()
print("ok", float(dq.float().abs().mean()), float(dk.float().abs().mean()), float(dv.float().abs().mean()))

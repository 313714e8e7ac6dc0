"""Developer tool: launch the attention forward a few times at one shape (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import ops
G, H, T, d = (int(x) for x in sys.argv[1:5])
dt = torch.bfloat16
C = H * d
g = torch.Generator().manual_seed(0)
qkv = torch.randn(G, T, 3 * C, generator=g).cuda().to(dt)
bias = torch.randn(G, H, T, T, generator=g).cuda()
pack = ops.pack_bias(bias, G, H, T, dtype=dt)
q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
for _ in range(5):
    ops._attn_fwd(q, k, v, pack, d ** -0.5, 0.0, 1, None)
torch.cuda.synchronize()

"""Developer tool / bench.py's PMC child: run the attention forward + backward (training instantiation) at one or more
shapes, for `rocprofv3 --kernel-trace --stats` or `--pmc` runs.  Shapes: env SHAPES="G:T:d,G:T:d" (default: the c5 shape
from GG / T / D); REPS launches each, P = attention dropout, NODBIAS=1 drops the dBias output."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import ops

H = 8
default = "%s:%s:%s" % (os.environ.get("GG", 16), os.environ.get("T", 785), os.environ.get("D", 32))
shapes = [tuple(int(x) for x in s.split(":")) for s in os.environ.get("SHAPES", default).split(",")]
p_drop = float(os.environ.get("P", 0.1))
dt = torch.bfloat16
for G, T, d in shapes:
    C = H * d
    g = torch.Generator().manual_seed(0)
    q, k, v, do = (torch.randn(G, T, C, generator=g).cuda().to(dt) for _ in range(4))
    bias = torch.randn(G, H, T, T, generator=g).cuda()
    pack = ops.pack_bias(bias, G, H, T, dtype=dt)
    del bias
    pack.needs_grad = os.environ.get('NODBIAS') != '1'
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    for _ in range(int(os.environ.get("REPS", 10))):
        out, lse = ops._attn_fwd(q, k, v, pack, d ** -0.5, p_drop, 1, None)
        pack.n_bwd = 0
        ops._attn_bwd(q, k, v, out, lse, do, dq, dk, dv, pack, d ** -0.5, p_drop, 1, None)
    torch.cuda.synchronize()
    print("ok", (G, T, d), float(dq.float().abs().mean()), float(dk.float().abs().mean()), float(dv.float().abs().mean()))

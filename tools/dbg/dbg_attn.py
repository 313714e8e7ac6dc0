import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from mobgt_amd import ops
torch.manual_seed(0)
for bdt in (torch.float32, torch.bfloat16):
    for (G, H, T, d) in ((2, 8, 1, 16), (2, 8, 5, 16), (2, 8, 33, 16), (2, 8, 65, 24), (1, 8, 130, 32), (2, 8, 785, 32)):
        C = H * d
        q = torch.zeros(G, T, C, device="cuda")
        k = torch.zeros(G, T, C, device="cuda")
        v = torch.randn(G, T, C, device="cuda")
        bias = torch.randn(G, H, T, T, device="cuda")
        if bdt == torch.bfloat16:
            bias = bias.bfloat16().float()
        pack = ops.pack_bias(bias, G, H, T, dtype=bdt)
        out = ops.attention(q, k, v, pack, d ** -0.5)
        p = torch.softmax(bias, -1)
        vh = v.bfloat16().float().view(G, T, H, d).transpose(1, 2)
        ref = (p @ vh).transpose(1, 2).reshape(G, T, C)
        err = (out - ref).abs()
        print(bdt, (G, H, T, d), "max err %.4f" % float(err.max()), "rows bad", int((err.amax(-1) > 0.02).sum()), "of", G * T)
        if float(err.max()) > 0.05 and T <= 5:
            print(out[0, :, :4], ref[0, :, :4])

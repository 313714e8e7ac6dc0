"""Developer tool (GPU box, after tools/gcn_stamp.sh): timeline of workgroup SG_WG of the category GCN's two launches
(100 MHz wall clock).  python tools/gcn_stamp.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOBGT_HIP_LIB", os.path.join(ROOT, "mobgt_amd", "libmobgt_hip_sgstamp.so"))
import numpy as np
import torch
from mobgt_amd import _lib
from mobgt_amd.ops import _p, _stream
n, K0, H1, H2, H3 = 300, 300, 16, 64, 32
dev = "cuda"
A = torch.rand(n, n, device=dev) / n
AT = A.t().contiguous()
AX = torch.randn(n, K0, device=dev)
ws = [torch.randn(K0, H1, device=dev) * 0.1, torch.zeros(H1, device=dev), torch.randn(H1, H2, device=dev) * 0.1, torch.zeros(H2, device=dev),
      torch.randn(H2, H3, device=dev) * 0.1, torch.zeros(H3, device=dev)]
h1, t, h2, t2 = (torch.empty(n, w, device=dev) for w in (H1, H1, H2, H2))
out = torch.empty(n, H3, device=dev)
filler = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for it in range(4):
    filler.random_(0, 255)
    counter = torch.zeros(64, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mobgt_small_gcn_fwd(_p(AX), _p(A), *[_p(w) for w in ws], _p(h1), _p(t), _p(h2), _p(t2), _p(out), _p(counter),
                                              n, K0, H1, H2, H3, 0.2, 0.3, 1, None, 7, _stream()), "fwd")
    torch.cuda.synchronize()
s = counter[4:20].cpu().numpy().astype(np.int64)
names = {0: "start", 10: "first burst requested", 11: "weights stored", 12: "burst stored", 1: "barrier", 2: "layer 1 product", 3: "h1 stored", 4: "grid barrier 1",
         5: "t = A h1", 6: "h2 stored", 7: "grid barrier 2", 8: "t2 = A h2", 9: "out"}
order = [0, 10, 11, 12, 1, 2, 3, 4, 5, 6, 7, 8, 9]
print("forward, workgroup %s:" % os.environ.get("SG_WG", "0"))
prev = s[0]
for k in order:
    print("  %-24s %7.2f us  (+%.2f)" % (names[k], (s[k] - s[0]) * 0.01, (s[k] - prev) * 0.01))
    prev = s[k]
g = torch.randn(n, H3, device=dev)
grads = [torch.zeros_like(w) for w in ws]
scratch = torch.empty(n * (H1 + H2), device=dev)
for it in range(4):
    filler.random_(0, 255)
    counter = torch.zeros(64, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mobgt_small_gcn_bwd(_p(g), _p(AX), _p(AT), _p(ws[2]), _p(ws[4]), _p(h1), _p(t), _p(h2), _p(t2), *[_p(x) for x in grads],
                                              _p(scratch[n * H1:]), _p(scratch[:n * H1]), _p(counter), n, K0, H1, H2, H3, 0.2, 0.3, 1, None, 7,
                                              _stream()), "bwd")
    torch.cuda.synchronize()
s = counter[4:20].cpu().numpy().astype(np.int64)
print("backward:")
prev = s[0]
for k in range(9):
    print("  stamp %d %7.2f us  (+%.2f)" % (k, (s[k] - s[0]) * 0.01, (s[k] - prev) * 0.01))
    prev = s[k]

"""Developer tool: in-kernel timeline of small_gcn_fwd (build smallgcn.hip with -DSG_DEBUG -DSG_WG=<workgroup>)."""
import torch, sys
sys.path.insert(0, ".")
from mobgt_amd import _lib
from mobgt_amd.ops import _p, _stream
n, K0, H1, H2, H3 = 300, 300, 16, 64, 32
dev = torch.device("cuda")
a = torch.rand(n, n, device=dev); ax = torch.rand(n, K0, device=dev)
ws = [torch.rand(K0, H1, device=dev), torch.rand(H1, device=dev), torch.rand(H1, H2, device=dev), torch.rand(H2, device=dev),
      torch.rand(H2, H3, device=dev), torch.rand(H3, device=dev)]
keep = torch.empty(n * (2 * H1 + 2 * H2), device=dev)
h1, t, h2, t2 = keep.split([n * H1, n * H1, n * H2, n * H2])
out = torch.empty(n, H3, device=dev)
for it in range(5):
    counter = torch.zeros(64, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().mobgt_small_gcn_fwd(_p(ax), _p(a), *[_p(w) for w in ws], _p(h1), _p(t), _p(h2), _p(t2), _p(out),
                                              _p(counter), n, K0, H1, H2, H3, 0.2, 0.1, 1, None, 5, _stream()), "fwd")
    torch.cuda.synchronize()
    c = counter.cpu().numpy()[4:20].astype("int64")
    print([int((x - c[0]) & 0xffffffff) * 10 for x in c], "ns")

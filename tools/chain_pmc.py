"""Developer tool (GPU box): N stand-alone launches of mobgt_layer_chain_fwd (cluster form when the workspace allows it) at
R rows, each behind a 64 MB filler that evicts the L2s -- the target of `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace`
(tools/chain_pmc.sh).  python tools/chain_pmc.py [R] [launches]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mobgt_amd import _lib
from mobgt_amd.fused_layer import chain_workspace
from mobgt_amd.ops import _p, _stream
R = int(sys.argv[1]) if len(sys.argv) > 1 else 608
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
C, F = 192, 1024
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.05).bfloat16()


def pack(w):
    out = torch.empty_like(w)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    _lib.check(_lib.lib().mobgt_pack_mfma_b(1, (vp * 1)(w.data_ptr()), (vp * 1)(out.data_ptr()), (ci * 1)(w.shape[0]),
                                            (ci * 1)(w.shape[1]), None, _stream()), "mobgt_pack_mfma_b")
    return out
a, x = bf(R, C), torch.randn(R, C, device="cuda")
wo, w1, w2, wq = pack(bf(C, C)), pack(bf(F, C)), pack(bf(C, F)), pack(bf(3 * C, C))
bo, b1, b2, bq = bf(C), bf(F), bf(C), bf(3 * C)
ln = [torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")]
x1, x2, out = (torch.empty(R, C, device="cuda") for _ in range(3))
z, out_a, u, h, qkv = bf(R, C), bf(R, C), bf(R, F), bf(R, F), bf(R, 3 * C)
st = torch.empty(4, R, device="cuda")
ws = chain_workspace(a.device, C, R)
filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for it in range(N):
    if os.environ.get("NOFILL") != "1":
        filler.fill_(it & 255)
    _lib.check(_lib.lib().mobgt_layer_chain_fwd(_p(a), _p(x), _p(wo), _p(bo), _p(ln[0]), _p(ln[1]), _p(w1), _p(b1), _p(w2), _p(b2),
                                                _p(ln[2]), _p(ln[3]), _p(wq), _p(bq), _p(x1), _p(z), _p(u), _p(h), _p(x2), _p(out),
                                                _p(out_a), _p(qkv), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), R, C, F, 0.1, 1, None,
                                                9, 10, _p(ws), _stream()), "mobgt_layer_chain_fwd")
torch.cuda.synchronize()
print("done", R, N)

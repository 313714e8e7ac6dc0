"""Developer tool: in-kernel timeline of layer_chain_fwd (build chain.hip with -DCH_DEBUG)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
from mobgt_amd import _lib
from mobgt_amd.ops import _p, _stream
R, C, F = int(sys.argv[1]) if len(sys.argv) > 1 else 608, 192, 1024
dev = torch.device("cuda")
bf = lambda *s: (torch.randn(*s, device=dev) * 0.05).bfloat16()
a, x = bf(R, C), torch.randn(R, C, device=dev)
wo, bo, w1, b1, w2, b2, wq, bq = bf(C, C), bf(C), bf(F, C), bf(F), bf(C, F), bf(C), bf(3 * C, C), bf(3 * C)
ln = [torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev)]
x1, x2, out = (torch.empty(R, C, device=dev) for _ in range(3))
z, out_a, u, h, qkv = bf(R, C), bf(R, C), bf(R, F), bf(R, F), bf(R, 3 * C)
st = torch.empty(4, R, device=dev)
dbg = torch.zeros(16, dtype=torch.int32, device=dev)
lib = _lib.lib()


def pack(w):
    out = torch.empty_like(w)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    _lib.check(lib.mobgt_pack_mfma_b(1, (vp * 1)(w.data_ptr()), (vp * 1)(out.data_ptr()), (ci * 1)(w.shape[0]), (ci * 1)(w.shape[1]), None,
                                     _stream()), "pack")
    return out


wo, w1, w2, wq = pack(wo), pack(w1), pack(w2), pack(wq)
lib.mobgt_chain_debug_buffer.argtypes = [ctypes.c_void_p]
lib.mobgt_chain_debug_buffer(dbg.data_ptr())
junk = torch.empty(64 << 20, device=dev)
for it in range(6):
    if it % 2:
        junk.fill_(1.0)             # odd iterations: caches flushed by 256 MB of traffic
    torch.cuda.synchronize()
    _lib.check(lib.mobgt_layer_chain_fwd(_p(a), _p(x), _p(wo), _p(bo), _p(ln[0]), _p(ln[1]), _p(w1), _p(b1), _p(w2), _p(b2), _p(ln[2]),
                                         _p(ln[3]), _p(wq), _p(bq), _p(x1), _p(z), _p(u), _p(h), _p(x2), _p(out), _p(out_a), _p(qkv),
                                         _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), R, C, F, 0.1, 1, None, 9, 10, _stream()), "chain")
    torch.cuda.synchronize()
    c = dbg.cpu().numpy().astype("int64")
    print("flushed" if it % 2 else "warm   ", [int((v - c[0]) & 0xffffffff) * 10 for v in c[:10]], "ns")

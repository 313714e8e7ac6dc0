"""Developer tool (GPU box, after tools/chain_stamp.sh): timeline of the 64-row forward chain (layer_chain_fwd_big_kernel), every
workgroup's thread 0 (100 MHz wall clock: 10 ns per tick).  python tools/chain_big_stamp.py [R] [C]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOBGT_HIP_LIB", os.path.join(ROOT, "mobgt_amd", "libmobgt_hip_chstamp.so"))
import numpy as np
import torch
from mobgt_amd import _lib
from mobgt_amd.fused_layer import chain_workspace
from mobgt_amd.ops import _p, _stream
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12560
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
F = 1024
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
dbg = torch.zeros(1024 * 16, dtype=torch.int32, device="cuda")
raw = ctypes.CDLL(os.environ["MOBGT_HIP_LIB"])
raw.mobgt_chain_debug_buffer.argtypes = [ctypes.c_void_p]
assert raw.mobgt_chain_debug_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for it in range(5):
    filler.random_(0, 255)
    dbg.zero_()
    ev[0].record()
    _lib.check(_lib.lib().mobgt_layer_chain_fwd(_p(a), _p(x), _p(wo), _p(bo), _p(ln[0]), _p(ln[1]), _p(w1), _p(b1), _p(w2), _p(b2),
                                                _p(ln[2]), _p(ln[3]), _p(wq), _p(bq), _p(x1), _p(z), _p(u), _p(h), _p(x2), _p(out),
                                                _p(out_a), _p(qkv), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), R, C, F, 0.1, 1, None,
                                                9, 10, _p(ws), _stream()), "mobgt_layer_chain_fwd")
    ev[1].record()
    torch.cuda.synchronize()
print("launch %.1f us (events)" % (ev[0].elapsed_time(ev[1]) * 1e3))
d = dbg.view(-1, 16).cpu().numpy().astype(np.int64)
live = d[d[:, 13] != 0]
t0 = live[:, 0].min()
rel = (live[:, :14] - t0) * 0.01
names = ["start", "a/x in LDS", "out-proj + x1", "LN1", "chunk 0: FFN1", "chunk 0: gelu + u/h out", "-", "chunk 0 done (FFN2)", "chunk 1 done",
         "chunk 2 done", "x2", "LN2", "QKV", "end"]
print("R %d C %d: %d workgroups; last end %.2f us after the first start" % (R, C, len(live), rel[:, 13].max()))
print("%-26s %8s %8s %8s   %s" % ("stamp", "median", "min", "max", "median step"))
prev = None
for k, n in enumerate(names):
    if n == "-":
        continue
    col = rel[:, k]
    step = "" if prev is None else "%.2f" % float(np.median(col - prev))
    print("%-26s %8.2f %8.2f %8.2f   %s" % (n, np.median(col), col.min(), col.max(), step))
    prev = col

if live[:, 6].any():
    t = lambda k: float(np.median((live[:, k] - live[:, 1]) * 0.01))
    print("first product (out-proj), thread 0, after 'a/x in LDS': first chunks requested %.2f us, K loop done %.2f, epilogue done %.2f; phase end %.2f" % (
        t(6), t(14), t(15), t(2)))

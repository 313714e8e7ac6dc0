"""Developer tool (GPU box, after tools/chain_stamp.sh): timeline of the cluster form of mobgt_layer_chain_fwd, every
workgroup's thread 0 (100 MHz wall clock: 10 ns per tick).  python tools/chain_stamp.py [R]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOBGT_HIP_LIB", os.path.join(ROOT, "mobgt_amd", "libmobgt_hip_chstamp.so"))
import numpy as np
import torch
from mobgt_amd import _lib
from mobgt_amd.fused_layer import chain_workspace
from mobgt_amd.ops import _p, _stream
lib = _lib.lib()._lib if hasattr(_lib.lib(), "_lib") else _lib.lib()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 608
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
dbg = torch.zeros(1024 * 16, dtype=torch.int32, device="cuda")
raw = ctypes.CDLL(os.environ["MOBGT_HIP_LIB"])
raw.mobgt_chain_debug_buffer.argtypes = [ctypes.c_void_p]
assert raw.mobgt_chain_debug_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
filler = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for it in range(5):
    filler.random_(0, 255)                     # (evict the weights from L2 as the step's other kernels do)
    dbg.zero_()
    _lib.check(_lib.lib().mobgt_layer_chain_fwd(_p(a), _p(x), _p(wo), _p(bo), _p(ln[0]), _p(ln[1]), _p(w1), _p(b1), _p(w2), _p(b2),
                                                _p(ln[2]), _p(ln[3]), _p(wq), _p(bq), _p(x1), _p(z), _p(u), _p(h), _p(x2), _p(out),
                                                _p(out_a), _p(qkv), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), R, C, F, 0.1, 1, None,
                                                9, 10, _p(ws), _stream()), "mobgt_layer_chain_fwd")
    torch.cuda.synchronize()
d = dbg.view(-1, 16).cpu().numpy().astype(np.int64)
live = d[d[:, 10] != 0]
t0 = live[:, 0].min()
rel = (live[:, :11] - t0) * 0.01
names = ["start", "a/x in LDS", "Wo (full)", "LN1", "FFN1 slice", "u/h out + FFN2 partial", "put", "get + x2", "LN2", "QKV slice", "end"]
print("R %d: %d live workgroups; last end %.2f us after the first start" % (R, len(live), rel[:, 10].max()))
print("%-22s %8s %8s %8s   %s" % ("stamp", "median", "min", "max", "median step"))
prev = None
for k, n in enumerate(names):
    col = rel[:, k]
    step = "" if prev is None else "%.2f" % float(np.median(col - prev))
    print("%-22s %8.2f %8.2f %8.2f   %s" % (n, np.median(col), col.min(), col.max(), step))
    prev = col

print("poll rounds of thread 0: median %d, max %d; thread 0 done with get %.2f us after STAMP(6) (median)" % (
    np.median(live[:, 11]), live[:, 11].max(), float(np.median((live[:, 12] - live[:, 6]) * 0.01))))
if live[:, 13].any():
    print("put acknowledged %.2f us after STAMP(6) (median, thread 0)" % float(np.median((live[:, 13] - live[:, 6]) * 0.01)))
if live[:, 15].any():
    print("thread 0: Wo product done %.2f us after STAMP(1); W1 slice chunks arrive %.2f us after their request" % (
        float(np.median((live[:, 14] - live[:, 1]) * 0.01)), float(np.median((live[:, 15] - live[:, 14]) * 0.01))))

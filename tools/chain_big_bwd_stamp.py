"""Developer tool (GPU box, after tools/chain_stamp.sh): timeline of the 64-row backward chain (layer_chain_bwd_big_kernel), every
workgroup's thread 0 (100 MHz wall clock: 10 ns per tick).  python tools/chain_big_bwd_stamp.py [R] [C]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOBGT_HIP_LIB", os.path.join(ROOT, "mobgt_amd", "libmobgt_hip_chstamp.so"))
import numpy as np
import torch
from mobgt_amd import _lib
from mobgt_amd.ops import _p, _stream
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12560
C = int(sys.argv[2]) if len(sys.argv) > 2 else 256
F = 1024
bf = lambda *s: (torch.randn(*s, device="cuda") * 0.05).bfloat16()
f32 = lambda *s: torch.randn(*s, device="cuda")


def pack(w):
    out = torch.empty_like(w)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    _lib.check(_lib.lib().mobgt_pack_mfma_b(1, (vp * 1)(w.data_ptr()), (vp * 1)(out.data_ptr()), (ci * 1)(w.shape[0]),
                                            (ci * 1)(w.shape[1]), None, _stream()), "mobgt_pack_mfma_b")
    return out
dout, x2, x1 = f32(R, C), f32(R, C), f32(R, C)
u = bf(R, F)
st = torch.rand(4, R, device="cuda") + 0.5
w2t, w1t, wot = pack(bf(F, C)), pack(bf(C, F)), pack(bf(C, C))
n1w, nxw = torch.ones(C, device="cuda"), torch.ones(C, device="cuda")
df, dy, da, du = bf(R, C), bf(R, C), bf(R, C), bf(R, F)
dx1 = torch.empty(R, C, device="cuda")
sums = torch.zeros(6, C, device="cuda")
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
    _lib.check(_lib.lib().mobgt_layer_chain_bwd(_p(dout), _p(x2), _p(x1), _p(u), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _p(n1w), _p(nxw),
                                                _p(w2t), _p(w1t), _p(wot), _p(df), _p(du), _p(dy), _p(da), _p(dx1), _p(sums[0]), _p(sums[1]),
                                                _p(sums[2]), _p(sums[3]), _p(sums[4]), _p(sums[5]), R, C, F, 0.1, 1, None, 9, 10, None, None, 0,
                                                None, None, None, None, None, None, None, None, None, None, _stream()), "mobgt_layer_chain_bwd")
    ev[1].record()
    torch.cuda.synchronize()
print("launch %.1f us (events)" % (ev[0].elapsed_time(ev[1]) * 1e3))
d = dbg.view(-1, 16).cpu().numpy().astype(np.int64)
live = d[d[:, 11] != 0]
t0 = live[:, 0].min()
rel = (live[:, :12] - t0) * 0.01
names = ["start", "norm2' (dx2, df)", "column sums", "u chunk 0 staged", "chunk 0: du = (df W2) gelu'", "chunk 0 done (du out, dz)", "chunk 1 done",
         "chunk 2 done", "norm1' (dx1, dy)", "column sums", "da = dy Wo", "end"]
print("R %d C %d: %d workgroups; last end %.2f us after the first start" % (R, C, len(live), rel[:, 11].max()))
print("%-30s %8s %8s %8s   %s" % ("stamp", "median", "min", "max", "median step"))
prev = None
for k, n in enumerate(names):
    col = rel[:, k]
    step = "" if prev is None else "%.2f" % float(np.median(col - prev))
    print("%-30s %8.2f %8.2f %8.2f   %s" % (n, np.median(col), col.min(), col.max(), step))
    prev = col

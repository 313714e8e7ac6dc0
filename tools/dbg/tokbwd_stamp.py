"""Developer tool (GPU box): timeline of mobgt_token_bwd_chain's node blocks (stamp build: hipcc -DTB_STAMP into
mobgt_amd/libmobgt_hip_tbstamp.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MOBGT_HIP_LIB", os.path.join(ROOT, "mobgt_amd", "libmobgt_hip_tbstamp.so"))
import numpy as np, torch
from mobgt_amd import _lib
from mobgt_amd.ops import _p, _stream
G, N, C, W2 = 16, 37, 192, 160
R = G * N
dev = "cuda"
dout = torch.randn(G, N + 1, C, device=dev); real = torch.ones(R, device=dev)
y4 = torch.randn(R, C, device=dev); x4 = torch.randn(R, C, device=dev)
w4 = torch.randn(C, C, device=dev) * 0.1; w2 = torch.randn(W2, W2, device=dev) * 0.1
d_nf, d_add, dx4 = (torch.empty(R, C, device=dev) for _ in range(3))
d_pt = torch.empty(R, W2, device=dev); d_tok = torch.zeros(C, device=dev)
dbg = torch.zeros(256 * 8, dtype=torch.int32, device=dev)
raw = ctypes.CDLL(os.environ["MOBGT_HIP_LIB"]); raw.mobgt_tokbwd_debug_buffer.argtypes = [ctypes.c_void_p]
assert raw.mobgt_tokbwd_debug_buffer(ctypes.c_void_p(dbg.data_ptr())) == 0
filler = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
for it in range(4):
    filler.fill_(it)
    _lib.check(_lib.lib().mobgt_token_bwd_chain(_p(dout), _p(real), _p(y4), _p(x4), C, _p(w4), _p(w2), _p(d_nf), _p(d_add), _p(dx4), C, _p(d_pt),
                                                _p(d_tok), G, N, C, W2, 0.2, 0.2, 0.1, 0.1, 5, None, 1, 2, 3, _stream()), "x")
    torch.cuda.synchronize()
d = dbg.view(-1, 8).cpu().numpy().astype(np.int64)
live = d[d[:, 5] != 0]
t0 = live[:, 0].min()
names = ["start", "rows done", "dx4 product", "g2 ready", "d_pt product", "end"]
for k, n in enumerate(names):
    col = (live[:, k] - t0) * 0.01
    print("%-14s median %6.2f  max %6.2f" % (n, np.median(col), col.max()))

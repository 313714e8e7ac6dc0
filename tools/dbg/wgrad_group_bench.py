"""Developer tool (GPU box): the S-FSQ step's grouped leaf weight gradients (mobgt_linear_wgrad_multi), each problem alone and
together, graph-timed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mobgt_amd import _lib
from mobgt_amd.ops import _stream
probs = [(16, 320, 320, 0, 0, 1), (592, 192, 192, 1, 0, 1), (592, 160, 160, 1, 0, 1), (608, 64, 128, 0, 0, 0), (7856, 16, 64, 0, 1, 1), (7856, 304, 16, 0, 1, 1)]
dev = "cuda"
data = []
for R, M, N, mg, mx, hasdb in probs:
    g, x = torch.randn(R, M, device=dev), torch.randn(R, N, device=dev)
    data.append(dict(g=g, x=x, gm=torch.randn(R, M, device=dev) if mg else None, xm=torch.randn(R, N, device=dev) if mx else None,
                     dw=torch.zeros(M, N, device=dev), db=torch.zeros(M if not mx else N, device=dev) if hasdb else None, R=R, M=M, N=N, dbx=mx))
vp, i64, ci, cf = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
ptr = lambda t: t.data_ptr() if t is not None else None


def call(items):
    n = len(items)
    _lib.check(_lib.lib().mobgt_linear_wgrad_multi(
        n, (vp * n)(*[ptr(d["g"]) for d in items]), (i64 * n)(*[d["g"].stride(0) for d in items]),
        (vp * n)(*[ptr(d["x"]) for d in items]), (i64 * n)(*[d["x"].stride(0) for d in items]),
        (vp * n)(*[ptr(d["gm"]) for d in items]), (vp * n)(*[ptr(d["xm"]) for d in items]), (cf * (3 * n))(*([1.0, 0.2, 0.2] * n)),
        (vp * n)(*[ptr(d["dw"]) for d in items]), (i64 * n)(*[d["dw"].stride(0) for d in items]),
        (vp * n)(*[ptr(d["db"]) for d in items]), (ci * n)(*[int(d["dbx"]) for d in items]),
        (i64 * n)(*[d["R"] for d in items]), (ci * n)(*[d["M"] for d in items]), (ci * n)(*[d["N"] for d in items]), (ci * n)(*[1] * n),
        _stream()), "multi")


def timeit(f, n=30):
    filler = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                filler.fill_(i)
                f()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=s):
            for i in range(n):
                filler.fill_(i)
    torch.cuda.synchronize()
    out = []
    for gr in (g, g2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return out[0] - out[1]
for d in data:
    print("R %5d M %4d N %4d: %.1f us alone" % (d["R"], d["M"], d["N"], timeit(lambda: call([d]))))
print("all six: %.1f us" % timeit(lambda: call(data)))
print("the three short ones: %.1f us;  the three long ones: %.1f us" % (timeit(lambda: call(data[:3])), timeit(lambda: call(data[3:]))))
print("the two 7856-row ones: %.1f us;  304 x 16 alone again: %.1f us" % (timeit(lambda: call(data[4:])), timeit(lambda: call(data[5:]))))

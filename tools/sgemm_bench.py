"""Developer tool: mobgt_small_gemm_f32 vs the library for the GCN / fuse / head shapes (graph-replayed, per-call us)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import ops
dev = torch.device("cuda")


def timeit(fn, n=20, reps=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * n) * 1e6


for M, N, K, nk in [(7856, 16, 303, 0), (7856, 64, 16, 0), (7856, 128, 64, 0), (300, 16, 300, 0), (300, 64, 16, 0), (300, 64, 300, 0),
                    (300, 32, 64, 0), (608, 160, 160, 1), (608, 192, 192, 1), (608, 160, 160, 0), (608, 192, 192, 0), (16, 320, 320, 1),
                    (16, 320, 320, 0), (7856, 64, 128, 1), (7856, 16, 64, 1), (300, 64, 32, 1), (300, 16, 64, 1), (300, 64, 300, 1)]:
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev) if nk else torch.randn(K, N, device=dev)
    lib = (lambda: a @ b.t()) if nk else (lambda: a @ b)
    mine = lambda: ops.small_gemm(a, b, None, bool(nk))
    err = float((mine() - lib()).abs().max() / lib().abs().max())
    print(f"M={M:5d} N={N:4d} K={K:4d} nk={nk}  library {timeit(lib):6.2f} us   small_gemm {timeit(mine):6.2f} us   rel err {err:.1e}")

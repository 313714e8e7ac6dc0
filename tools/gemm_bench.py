"""Developer tool: mobgt_layer_gemm vs the library for the encoder layer's GEMM shapes (graph-replayed, per-call us)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mobgt_amd import ops
dev = torch.device("cuda")
R, C, F = int(os.environ.get("R", 608)), int(os.environ.get("C", 192)), 1024


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


for name, M, N, K, kn in [("qkv", R, 3 * C, C, False), ("out", R, C, C, False), ("ffn1", R, F, C, False), ("ffn2", R, C, F, False),
                          ("dh", R, F, C, True), ("dz", R, C, F, True), ("da", R, C, C, True), ("dx", R, C, 3 * C, True)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(K, N, device=dev).bfloat16() if kn else torch.randn(N, K, device=dev).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    lib = (lambda: a @ w) if kn else (lambda: torch.addmm(b, a, w.t()))
    mine = lambda: ops.layer_gemm(a, w, None if kn else b, kn)
    print(f"{name:5s} M={M} N={N} K={K} kn={int(kn)}  library {timeit(lib):6.2f} us   layer_gemm {timeit(mine):6.2f} us")

"""Developer tool: time mobgt_linear_wgrad against torch's GEMM paths at the encoder's weight-gradient shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgt_amd import ops
from mobgt_amd.fused_layer import _mm_tn_f32

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for R, M, N in [(2432, 192, 192), (2432, 576, 192), (2432, 1024, 192), (2432, 192, 1024), (800, 1024, 192), (800, 192, 1024), (12560, 256, 256), (12560, 768, 256), (12560, 256, 1024)]:
    g = torch.randn(R, M, device="cuda").bfloat16(); x = torch.randn(R, N, device="cuda").bfloat16()
    a = timeit(lambda: ops.linear_wgrad(g, x, with_bias=True))
    b = timeit(lambda: _mm_tn_f32(g, x))
    c = timeit(lambda: torch.mm(g.t(), x, out_dtype=torch.float32))
    print(f"R={R} M={M} N={N}: wgrad {a:.1f} us (incl 2 zero fills)  splitk-bmm {b:.1f} us  plain mm {c:.1f} us")

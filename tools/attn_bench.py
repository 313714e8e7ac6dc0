"""Developer tool: time the attention kernels at given shapes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from mobgt_amd import ops

def run(G, H, T, d, dt=torch.bfloat16, reps=30):
    t = bench.time_attention_kernel(G, H, T, d, dt, dt, reps=reps)
    b = bench.attn_algorithmic_bytes(G, T, H * d, H, 2, 2)
    print(f"fwd G{G} H{H} T{T} d{d}: {t*1e6:8.1f} us  {b/t/1e9:8.1f} GB/s  {b/t/1e9/8000*100:5.1f}% of 8 TB/s")

def run_bwd(G, H, T, d, dt=torch.bfloat16, reps=10):
    C = H * d
    g = torch.Generator().manual_seed(0)
    q, k, v, do = (torch.randn(G, T, C, generator=g).cuda().to(dt) for _ in range(4))
    bias = torch.randn(G, H, T, T, generator=g).cuda()
    pack = ops.pack_bias(bias, G, H, T, dtype=dt)
    pack.needs_grad = True
    out, lse = ops._attn_fwd(q, k, v, pack, d ** -0.5, 0.0, 1, None)
    dq, dk, dv = (torch.empty_like(q) for _ in range(3))
    for _ in range(3):
        pack.n_bwd = 0
        ops._attn_bwd(q, k, v, out, lse, do, dq, dk, dv, pack, d ** -0.5, 0.0, 1, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        pack.n_bwd = 0
        ops._attn_bwd(q, k, v, out, lse, do, dq, dk, dv, pack, d ** -0.5, 0.0, 1, None)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 1e3 / reps
    b = G * (20 * T * C * 2 + H * T * T * (2 + 8 + 2))
    print(f"bwd G{G} H{H} T{T} d{d}: {t*1e6:8.1f} us  {b/t/1e9:8.1f} GB/s (dq+dbias RMW + dkv passes)")

if __name__ == "__main__":
    for shp in [(16, 8, 785, 32), (16, 8, 815, 24), (16, 8, 257, 24), (16, 8, 129, 16), (16, 8, 51, 24), (16, 8, 12, 24)]:
        run(*shp)
    for shp in [(16, 8, 785, 32), (16, 8, 51, 24)]:
        run_bwd(*shp)

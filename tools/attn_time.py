"""Developer tool: time the attention forward / backward launches at the c5 shape (and any G H T d given as arguments)
exactly as bench.py's roofline_stress leg does (hipGraph of back-to-back launches, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

G, H, T, d = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (16, 8, 785, 32)
dt = torch.bfloat16
for p in (0.1, 0.0):
    tf, tb = bench.time_attention(G, H, T, d, dt, dt, reps=20, p_drop=p, backward=True)
    bf = bench.attn_fwd_bytes(G, T, H * d, H, 2, 2)
    bb = bench.attn_bwd_bytes(G, T, H * d, H, 2, 2, 2)
    print(f"p={p}: fwd {tf*1e6:.1f} us ({bf/tf/1e9:.0f} GB/s, {bf/tf/8e12:.3f})  bwd {tb*1e6:.1f} us ({bb/tb/1e9:.0f} GB/s, {bb/tb/8e12:.3f})")

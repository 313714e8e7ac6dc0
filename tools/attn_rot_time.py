"""Developer tool (GPU box): in-step-like timing of the attention kernels at the c5 shape -- launches rotate over distinct
input sets larger than the Infinity Cache (bench.time_attention) -- and at an S-FSQ shape.  Prints microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

P = float(os.environ.get("P", 0.1))
for rep in range(int(os.environ.get("ROUNDS", 3))):
    f, b, n = bench.time_attention(16, 8, 785, 32, torch.bfloat16, torch.bfloat16, reps=24, p_drop=P, backward=True)
    print("c5 p=%.2f rotating(%d sets): fwd %.1f us  bwd %.1f us   frac fwd %.3f" % (P, n, f * 1e6, b * 1e6, 183.8784e6 / f / 8e12))
f, b, n = bench.time_attention(16, 8, 42, 24, torch.bfloat16, torch.bfloat16, reps=50, p_drop=0.1, backward=True)
print("fsq T42 d24: fwd %.2f us  bwd %.2f us" % (f * 1e6, b * 1e6))
f, b, n = bench.time_attention(16, 8, 130, 24, torch.bfloat16, torch.bfloat16, reps=50, p_drop=0.1, backward=True)
print("fsq T130 d24: fwd %.2f us  bwd %.2f us" % (f * 1e6, b * 1e6))

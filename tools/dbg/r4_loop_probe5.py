"""Debug probe: does what bench.py does BEFORE its fresh-batch loop (the c5 attention stress measurement: > 1 GB of rotating
inputs through the caching allocator, graph captures) slow the loop down?"""
import os, sys, time, gc
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from mobgt_amd import workloads
from mobgt_amd.train import EpochLoop
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
pool = workloads.make_pool("fsq", 64, 16, uni, seed0=5000)
data = [t for trajs in pool for t in trajs]
pc = time.perf_counter
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if mode in ("stress", "stress_empty"):
    t5f, t5b, n5 = bench.time_attention(16, 8, 785, 32, torch.bfloat16, torch.bfloat16, reps=24, p_drop=0.1, backward=True)
    print("stress done", t5f, t5b, "reserved MB", torch.cuda.memory_reserved() >> 20, "allocated MB", torch.cuda.memory_allocated() >> 20)
if mode == "stress_empty":
    torch.cuda.empty_cache()
loop = EpochLoop(model, coll, data, batch_size=16, seed=1)
out = []
for ep in range(0, 14):
    torch.cuda.synchronize(); t0 = pc()
    n = loop.run_epoch(ep)["steps"]
    torch.cuda.synchronize()
    out.append((pc() - t0) / n * 1e3)
print(mode, " ".join("%.3f" % v for v in out[1:]), "reserved MB", torch.cuda.memory_reserved() >> 20)

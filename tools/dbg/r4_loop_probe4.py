"""Debug probe: EpochLoop (fresh batch every step), time per epoch over many epochs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from mobgt_amd import workloads
from mobgt_amd.train import EpochLoop
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
pool = workloads.make_pool("fsq", 64, 16, uni, seed0=5000)
data = [t for trajs in pool for t in trajs]
loop = EpochLoop(model, coll, data, batch_size=16, seed=1)
pc = time.perf_counter
out = []
for ep in range(0, 30):
    torch.cuda.synchronize(); t0 = pc()
    n = loop.run_epoch(ep)["steps"]
    torch.cuda.synchronize()
    out.append(((pc() - t0) / n * 1e3, len(loop.slots)))
print(" ".join("%.3f/%d" % v for v in out))
# the same epoch again and again
out = []
for rep in range(10):
    torch.cuda.synchronize(); t0 = pc()
    n = loop.run_epoch(7)["steps"]
    torch.cuda.synchronize()
    out.append((pc() - t0) / n * 1e3)
print("epoch 7 repeated:", " ".join("%.3f" % v for v in out))

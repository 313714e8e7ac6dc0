"""Debug probe: per-chunk step time of the pre-collated S-FSQ step graph over many steps (is there a slow phase?), then the
same after an idle pause of the GPU."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from mobgt_amd import workloads
from mobgt_amd.train import TrainStep
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
pool = workloads.make_pool("fsq", 8, 16, uni)
batches = [coll(t) for t in pool]
ts = TrainStep(model, batches, use_graph=True, seed=1)
ts.prepare()
torch.cuda.synchronize()
def chunks(tag, n_chunks=12, per=100, i0=0):
    out = []
    for c in range(n_chunks):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(per):
            ts.step(i0 + c * per + i)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / per * 1e3)
    print(tag, " ".join("%.4f" % v for v in out), flush=True)
chunks("from cold:")
time.sleep(5.0)
chunks("after 5 s idle:", 8, 100, 5000)
time.sleep(0.5)
chunks("after 0.5 s idle:", 4, 100, 9000)

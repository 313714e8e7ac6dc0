"""Developer tool (GPU box): which gradients does FlatGrads.gather still copy (not written into a sink)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mobgt_amd import workloads
from mobgt_amd.train import TrainStep
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
batches = [coll(t) for t in workloads.make_pool("fsq", 2, 16, uni)]
orig = torch._foreach_copy_
names = {}
def spy(dst, src):
    tot = 0
    for d, s in zip(dst, src):
        tot += d.numel()
    print("foreach_copy: %d tensors, %d elements" % (len(dst), tot))
    for d in sorted(dst, key=lambda t: -t.numel())[:40]:
        print("   ", tuple(d.shape), names.get(d.data_ptr(), "?"))
    return orig(dst, src)
torch._foreach_copy_ = spy
ts = TrainStep(model, batches, use_graph=False, seed=1)
ts.prepare()
for p, v in zip(ts.flat.params, ts.flat.views):
    for n, q in model.named_parameters():
        if q is p:
            names[v.data_ptr()] = n
ts.step(0)
torch.cuda.synchronize()
print("flat params", sum(p.numel() for p in ts.flat.params), "n", len(ts.flat.params))
big = sorted(((p.numel(), n) for n, p in model.named_parameters()), reverse=True)[:12]
print(big)

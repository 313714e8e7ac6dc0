"""Developer tool (GPU box): which parameter gradients of a variant's train step still go through the trainer's gather copy
(train.FlatGrads.gather) instead of being written into their sinks.  python tools/dbg/gather_todo.py [fq|stock]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mobgt_amd import workloads, train
variant = sys.argv[1] if len(sys.argv) > 1 else "stock"
uni, model, coll = workloads.build("fsq", "cuda", seed=1, variant=variant)
batches = [coll(b) for b in workloads.make_pool("fsq", 2, 16, uni)]
ts = train.TrainStep(model, batches, use_graph=False, seed=5)
names = {id(p): n for n, p in model.named_parameters()}
orig = train.FlatGrads.gather


def spy(self, start=0, stop=None, grads=None):
    stop_ = len(self.params) if stop is None else stop
    gs = grads if grads is not None else [p.grad for p in self.params[start:stop_]]
    for p, v, g in zip(self.params[start:stop_], self.views[start:stop_], gs):
        if g is not None and g.data_ptr() != v.data_ptr():
            print("copied:", names.get(id(p)), tuple(p.shape))
    return orig(self, start, stop, grads)
train.FlatGrads.gather = spy
ts.step(0)
torch.cuda.synchronize()

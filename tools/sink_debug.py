"""Developer tool: which parameters' gradients still need the post-backward gather copy (no in-place sink)?"""
import sys
import torch
sys.path.insert(0, ".")
from mobgt_amd import workloads
from mobgt_amd import train as T

dev = torch.device("cuda")
uni, model, coll = workloads.build("fsq", dev)
pool = workloads.make_pool("fsq", 2, 16, uni)
batches = [coll(t) for t in pool]
names = {id(p): n for n, p in model.named_parameters()}
orig = T.FlatGrads.gather


def gather(self, start=0, stop=None, grads=None):
    stop_ = len(self.params) if stop is None else stop
    ps, vs = self.params[start:stop_], self.views[start:stop_]
    gs = grads if grads is not None else [p.grad for p in ps]
    todo = [(names.get(id(p), "?"), tuple(p.shape)) for p, g, v in zip(ps, gs, vs) if g is not None and g.data_ptr() != v.data_ptr()]
    print("gather", start, stop_, "copies", len(todo), "elements", sum(int(torch.tensor(s).prod()) for _, s in todo))
    for n, s in todo:
        print("   ", n, s)
    return orig(self, start, stop, grads)


T.FlatGrads.gather = gather
ts = T.TrainStep(model, batches, use_graph=False, seed=1)
ts.prepare()
print("---- step")
ts.step(0)
torch.cuda.synchronize()

"""Developer tool (GPU box): what is in the step's grouped weight-gradient launches, and which single weight-gradient launches
remain?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mobgt_amd import workloads, ops, _lib
from mobgt_amd.train import TrainStep
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
batches = [coll(t) for t in workloads.make_pool("fsq", 2, 16, uni)]
ts = TrainStep(model, batches, use_graph=False, seed=1)
ts.prepare()
orig = ops.flush_deferred_wgrads
def spy():
    items = ops._WGRAD_DEFER["items"]
    print("flush: %d problems" % len(items))
    for it in items:
        g, x = it[0], it[1]
        print("   R %5d  M %4d  N %4d  %s  mask_g %s mask_x %s db %s" % (g.shape[0], g.shape[1], x.shape[1], str(g.dtype)[6:], it[2] is not None, it[3] is not None, it[6] is not None))
    return orig()
ops.flush_deferred_wgrads = spy
import mobgt_amd.train as tr
real = _lib.lib()
class Spy:
    def __getattr__(self, name):
        f = getattr(real, name)
        if name.startswith("mobgt_linear_wgrad") and name != "mobgt_linear_wgrad_multi":
            def g(*a):
                ints = [int(v) if isinstance(v, int) else None for v in a]
                print("single launch %s  ints %s" % (name, [v for v in ints if v is not None and v < 10 ** 7]))
                return f(*a)
            return g
        return f
_lib._lib = Spy() if hasattr(_lib, "_lib") else None
ts.step(0)
torch.cuda.synchronize()

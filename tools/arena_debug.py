"""Developer tool: which zero-initialised accumulators of a training step do not fit the zero arena (each of those is a
fill launch)?  Run on the GPU box: python tools/arena_debug.py"""
import sys, traceback
import torch
sys.path.insert(0, ".")
from mobgt_amd import ops, workloads
from mobgt_amd.train import TrainStep

fails, total = [], [0]
orig = ops.ZeroArena.take


def take(self, n):
    t = orig(self, n)
    total[0] = max(total[0], self.off)
    if t is None:
        fails.append((n, self.off, [f"{f.name}:{f.lineno}" for f in traceback.extract_stack(limit=7)[:-1]]))
    return t


ops.ZeroArena.take = take
dev = torch.device("cuda")
uni, model, coll = workloads.build("fsq", dev)
pool = workloads.make_pool("fsq", 2, 16, uni)
batches = [coll(t) for t in pool]
ts = TrainStep(model, batches, use_graph=False, seed=1)
ts.prepare()
fails.clear()
ts.step(0)
torch.cuda.synchronize()
print("arena size", ts.arena.buf.numel(), "high water", total[0])
for n, off, where in fails:
    print("MISS", n, "at offset", off, " <- ".join(reversed(where)))

"""Developer tool: where do the remaining fill / zero / add launches of a training step come from?  (eager step under torch.profiler)"""
import sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, ".")
from mobgt_amd import workloads
from mobgt_amd.train import TrainStep

dev = torch.device("cuda")
uni, model, coll = workloads.build("fsq", dev)
pool = workloads.make_pool("fsq", 2, 16, uni)
batches = [coll(t) for t in pool]
ts = TrainStep(model, batches, use_graph=False, seed=1)
ts.prepare()
ts.step(0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    ts.step(1)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::add", "aten::add_", "aten::sum", "aten::copy_", "aten::cat"):
        st = [s for s in (ev.stack or []) if "mobgt_amd" in s or "autograd" in s][:4]
        print(ev.name, ev.input_shapes, " | ".join(st))

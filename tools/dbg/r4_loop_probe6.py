"""Debug probe: bench.py's order -- main TrainStep on 8 pre-collated batches, [the c5 attention stress measurement], then the
fresh-batch loop -- with the loop's host time by line group."""
import os, sys, time, types
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from mobgt_amd import workloads
from mobgt_amd.train import EpochLoop, TrainStep
from mobgt_amd.data import bucket_nodes
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
batches = [coll(t) for t in workloads.make_pool("fsq", 8, 16, uni)]
ts = TrainStep(model, batches, use_graph=True, seed=1)
ts.prepare()
for i in range(120):
    ts.step(i)
torch.cuda.synchronize()
if mode == "stress":
    bench.time_attention(16, 8, 785, 32, torch.bfloat16, torch.bfloat16, reps=24, p_drop=0.1, backward=True)
print(mode, "reserved MB", torch.cuda.memory_reserved() >> 20, "allocated MB", torch.cuda.memory_allocated() >> 20, flush=True)
pool = workloads.make_pool("fsq", 64, 16, uni, seed0=5000)
data = [t for trajs in pool for t in trajs]
loop = EpochLoop(model, coll, data, batch_size=16, seed=1)
loop.run_epoch(0); loop.run_epoch(1)
torch.cuda.synchronize()
acc = {}
pc = time.perf_counter
def add(k, t0):
    t1 = pc(); acc[k] = acc.get(k, 0.0) + t1 - t0; return t1
def stage(self, ids):
    t = pc()
    trajs = [self.dataset[i] for i in ids]
    G = len(trajs)
    N = bucket_nodes(max(len(t_["node_name"]) for t_ in trajs), self.buckets)
    slot = self._slot(G, N)
    st = slot["stages"][slot["turn"]]
    slot["turn"] ^= 1
    t = add("s.fetch+slot", t)
    if st["free"] is not None:
        st["free"].synchronize()
    t = add("s.free.sync", t)
    self.collator.pack_host(trajs, idx0=ids[:G] if len(ids) == G else 0, n_pad=N, out=st["np"])
    t = add("s.pack", t)
    self._check_host(st["np"])
    t = add("s.check", t)
    st["used"] = True
    with torch.cuda.stream(self.copy_stream):
        st["dev"][:st["pin"].numel()].copy_(st["pin"], non_blocking=True)
        t = add("s.h2d", t)
        st["work"] = self.collator.finish_into(st["dev_views"], st["work"])
        t = add("s.finish_into", t)
        st["ready"].record(self.copy_stream)
    t = add("s.record+exit", t)
    return slot, st
def launch(self, slot, st):
    t = pc()
    cur = torch.cuda.current_stream()
    cur.wait_event(st["ready"])
    n = slot["copy_bytes"]
    slot["buf"][:n].copy_(st["dev"][:n], non_blocking=True)
    if st["free"] is None:
        st["free"] = torch.cuda.Event()
    st["free"].record(cur)
    t = add("l.wait+d2d+record", t)
    if slot["index"] is None or self.ts is None:
        self._ensure_trainer(slot)
    r = self.ts.step(slot["index"])
    t = add("l.ts.step", t)
    return r
loop._stage = types.MethodType(stage, loop)
loop._launch = types.MethodType(launch, loop)
out = []
n = 0
for ep in range(2, 8):
    torch.cuda.synchronize(); t0 = pc()
    k = loop.run_epoch(ep)["steps"]; n += k
    torch.cuda.synchronize()
    out.append((pc() - t0) / k * 1e3)
print(mode, "ms/step per epoch:", " ".join("%.3f" % v for v in out))
print(mode, "host us/step:", {k: round(v / n * 1e6, 1) for k, v in acc.items()})

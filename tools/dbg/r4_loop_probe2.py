"""Debug probe: host time of EpochLoop._stage / _launch by line group (perf_counter around copies of the two methods' bodies)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from mobgt_amd import workloads
from mobgt_amd.train import EpochLoop
from mobgt_amd.data import bucket_nodes
uni, model, coll = workloads.build("fsq", "cuda", seed=1)
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
    trajs = [t_ for t_ in trajs if t_ is not None and len(t_["node_name"]) <= self.collator.max_node]
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
        t = add("s.stream_ctx", t)
        st["dev"][:st["pin"].numel()].copy_(st["pin"], non_blocking=True)
        t = add("s.h2d", t)
        if slot["side"]:
            st["work"] = self.collator.finish_into(st["dev_views"], st["work"])
        t = add("s.finish_into", t)
        st["ready"].record(self.copy_stream)
        t = add("s.record", t)
    t = add("s.stream_exit", t)
    return slot, st

def launch(self, slot, st):
    t = pc()
    cur = torch.cuda.current_stream()
    cur.wait_event(st["ready"])
    t = add("l.wait_event", t)
    n = slot["copy_bytes"]
    slot["buf"][:n].copy_(st["dev"][:n], non_blocking=True)
    t = add("l.d2d", t)
    if st["free"] is None:
        st["free"] = torch.cuda.Event()
    st["free"].record(cur)
    t = add("l.record", t)
    if slot["index"] is None or self.ts is None:
        self._ensure_trainer(slot)
    r = self.ts.step(slot["index"])
    t = add("l.ts.step", t)
    return r

import types
loop._stage = types.MethodType(stage, loop)
loop._launch = types.MethodType(launch, loop)
for ep in range(2, 4):
    loop.run_epoch(ep)
torch.cuda.synchronize()
acc.clear()
t0 = pc(); n = 0
for ep in range(20, 26):
    n += loop.run_epoch(ep)["steps"]
torch.cuda.synchronize()
print("ms/step %.4f over %d steps" % ((pc() - t0) / n * 1e3, n))
print("host us/step:", {k: round(v / n * 1e6, 1) for k, v in acc.items()})
